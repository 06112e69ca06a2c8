// core.hip -- error state, device selection, library info.
#include "common.h"

#include <atomic>
#include <mutex>
#include <cstdlib>

namespace cymf {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

static int device_count_quiet() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int staging_memtype() {
    static const int mt = [] {
        const char *e = getenv("CYMF_STAGING_MEMTYPE");
        const int v = e ? atoi(e) : 2;
        return v < 0 || v > 2 ? 2 : v;
    }();
    return mt;
}

int default_memtype() {
    static const int mt = [] {
        const char *e = getenv("CYMF_DEFAULT_MEMTYPE");
        const int v = e ? atoi(e) : 2;
        return v < 0 || v > 2 ? 2 : v;
    }();
    return mt;
}

// Set by an atexit hook registered when this library is loaded.  libamdhip64 is a dependency of this library, so it is
// loaded (and registers its own exit handlers) first; exit() runs handlers in reverse order of registration, hence
// ours runs BEFORE the HIP runtime tears itself down: from then on nothing here calls into HIP any more.
static std::atomic<bool> g_exiting{false};
static void mark_exiting() { g_exiting.store(true); }
__attribute__((constructor)) static void register_exit_hook() { atexit(mark_exiting); }

bool process_exiting() { return g_exiting.load(); }

bool runtime_alive(int device) {
    if (g_exiting.load()) return false;
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}

int use_device(int device) {
    int n = device_count_quiet();
    if (n <= 0)
        return fail(CYMF_ERR_NO_DEVICE, "no HIP device visible: libcymf_hip has no CPU fallback (gfx950 required)");
    if (device < 0 || device >= n) return fail(CYMF_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    CYMF_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CYMF_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    CYMF_HIP(hipSetDevice(device));
    // registered again after the first successful HIP call: should the runtime install its exit handler lazily (during
    // that call), ours still comes later in registration order, i.e. earlier in exit()
    static std::once_flag once;
    std::call_once(once, [] { atexit(mark_exiting); });
    return 0;
}

}  // namespace cymf

using namespace cymf;

extern "C" const char *cymf_last_error(void) { return g_err.c_str(); }
extern "C" int cymf_version(void) { return 100; }
extern "C" int cymf_device_count(void) { return device_count_quiet(); }

extern "C" int cymf_device_name(int device, char *buf, int buflen) {
    if (!buf || buflen <= 0) return fail(CYMF_ERR_INVALID, "cymf_device_name: bad buffer");
    int n = device_count_quiet();
    if (n <= 0) return fail(CYMF_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(CYMF_ERR_INVALID, "device %d out of range", device);
    hipDeviceProp_t prop;
    CYMF_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

extern "C" int cymf_device_sync(int device) {
    CYMF_TRY(use_device(device));
    CYMF_HIP(hipDeviceSynchronize());
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
}  // namespace

extern "C" int cymf_device_stream_copy_gbps(int device, int64_t bytes, int iters, double *gbps_out) {
    if (!gbps_out || bytes < (1 << 20) || iters < 1) return fail(CYMF_ERR_INVALID, "cymf_device_stream_copy_gbps: bad arguments");
    CYMF_TRY(use_device(device));
    const int64_t n = bytes / 16;
    DevBuf<float4> a, b;
    CYMF_TRY(a.alloc((size_t)n));
    CYMF_TRY(b.alloc((size_t)n));
    CYMF_HIP(hipMemset(a.p, 0, (size_t)n * 16));
    hipEvent_t e0, e1;
    CYMF_HIP(hipEventCreate(&e0));
    CYMF_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(256), 0, nullptr, a.p, b.p, n);   // warm-up
    CYMF_HIP(hipEventRecord(e0, nullptr));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(256), 0, nullptr, a.p, b.p, n);
    CYMF_HIP(hipEventRecord(e1, nullptr));
    CYMF_HIP(hipEventSynchronize(e1));
    float ms = 0;
    CYMF_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *gbps_out = 2.0 * (double)n * 16.0 * iters / (ms * 1e-3) / 1e9;
    return 0;
}

// ---------------------------------------------------------------- seam probe (diagnostic, include/cymf_amd.h)
namespace {
__global__ __launch_bounds__(256) void probe_read_kernel(const uint32_t *__restrict__ b, int n, uint32_t want,
                                                         unsigned long long *__restrict__ bad) {
    unsigned int miss = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) miss += b[i] != want ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) miss += __shfl_xor(miss, off, 64);
    if ((threadIdx.x & 63) == 0 && miss) atomicAdd(bad, (unsigned long long)miss);
}
__global__ __launch_bounds__(256) void probe_write_kernel(uint32_t *__restrict__ b, int n, uint32_t v) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) b[i] = v;
}
}  // namespace

extern "C" int cymf_device_seam_probe(int device, int memtype, int rounds, int64_t *stale_out, int64_t *words_out) {
    if (!stale_out || !words_out || rounds < 1 || memtype < 0 || memtype > 2)
        return fail(CYMF_ERR_INVALID, "cymf_device_seam_probe: bad arguments");
    CYMF_TRY(use_device(device));
    hipDeviceProp_t prop;
    CYMF_HIP(hipGetDeviceProperties(&prop, device));
    const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    constexpr int N = 4096;   // 16 KB: L1- and L2-resident between the kernels
    DevBuf<uint32_t> buf;
    buf.fine = memtype;
    CYMF_TRY(buf.alloc(N));
    DevBuf<unsigned long long> bad;   // [0] warm-up sink, [1..3] seams
    bad.fine = 2;
    CYMF_TRY(bad.alloc(4));
    uint32_t *host = nullptr;
    CYMF_HIP(hipHostMalloc((void **)&host, N * sizeof(uint32_t)));
    hipStream_t s0 = nullptr, s1 = nullptr;
    hipEvent_t ev = nullptr, ev_back = nullptr;
    CYMF_HIP(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CYMF_HIP(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CYMF_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CYMF_HIP(hipEventCreateWithFlags(&ev_back, hipEventDisableTiming));
    CYMF_HIP(hipMemsetAsync(bad.p, 0, 4 * sizeof(unsigned long long), s0));
    CYMF_HIP(hipMemsetAsync(buf.p, 0, N * sizeof(uint32_t), s0));
    int64_t host_bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint32_t old_v = (uint32_t)r, new_v = (uint32_t)r + 1;
        // every CU caches the old value
        hipLaunchKernelGGL(probe_read_kernel, dim3(n_cu), dim3(256), 0, s0, buf.p, N, old_v, bad.p + 0);
        // one workgroup (one XCD) rewrites the buffer
        hipLaunchKernelGGL(probe_write_kernel, dim3(1), dim3(256), 0, s0, buf.p, N, new_v);
        // seam 0: next kernel of the stream, all CUs; seam 1: next kernel, one workgroup
        hipLaunchKernelGGL(probe_read_kernel, dim3(n_cu), dim3(256), 0, s0, buf.p, N, new_v, bad.p + 1);
        hipLaunchKernelGGL(probe_read_kernel, dim3(1), dim3(256), 0, s0, buf.p, N, new_v, bad.p + 2);
        // seam 2: second stream behind an event
        CYMF_HIP(hipEventRecord(ev, s0));
        CYMF_HIP(hipStreamWaitEvent(s1, ev, 0));
        hipLaunchKernelGGL(probe_read_kernel, dim3(n_cu), dim3(256), 0, s1, buf.p, N, new_v, bad.p + 3);
        CYMF_HIP(hipGetLastError());
        CYMF_HIP(hipEventRecord(ev_back, s1));
        CYMF_HIP(hipStreamWaitEvent(s0, ev_back, 0));
        // seam 3: kernel-written buffer read by the copy engine
        CYMF_HIP(hipMemcpyAsync(host, buf.p, N * sizeof(uint32_t), hipMemcpyDeviceToHost, s0));
        CYMF_HIP(hipStreamSynchronize(s0));
        for (int i = 0; i < N; ++i) host_bad += host[i] != new_v ? 1 : 0;
    }
    unsigned long long h_bad[4] = {0, 0, 0, 0};
    CYMF_HIP(hipMemcpyAsync(h_bad, bad.p, sizeof h_bad, hipMemcpyDeviceToHost, s0));
    CYMF_HIP(hipStreamSynchronize(s0));
    CYMF_HIP(hipStreamSynchronize(s1));
    stale_out[0] = (int64_t)h_bad[1] / n_cu;   // per reader, like the single-workgroup seam
    stale_out[1] = (int64_t)h_bad[2];
    stale_out[2] = (int64_t)h_bad[3] / n_cu;
    stale_out[3] = host_bad;
    // round up: any stale reader must show
    if (h_bad[1] && !stale_out[0]) stale_out[0] = 1;
    if (h_bad[3] && !stale_out[2]) stale_out[2] = 1;
    *words_out = (int64_t)rounds * N;
    (void)hipEventDestroy(ev);
    (void)hipEventDestroy(ev_back);
    (void)hipStreamDestroy(s0);
    (void)hipStreamDestroy(s1);
    (void)hipHostFree(host);
    return 0;
}
