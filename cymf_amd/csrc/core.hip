// core.hip -- error state, device selection, library info.
#include "common.h"

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
