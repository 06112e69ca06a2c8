// common.h -- error plumbing, device buffers and wave-level helpers shared by the kernels.
// gfx950 (CDNA4) only: 64-lane wavefronts, DPP row operations, no portability layer.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cymf_amd.h"

namespace cymf {

// ------------------------------------------------------------------ errors
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define CYMF_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return ::cymf::fail(CYMF_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                   \
                                hipGetErrorString(e__), __FILE__, __LINE__);                    \
    } while (0)

#define CYMF_TRY(expr)                                                                          \
    do {                                                                                        \
        int rc__ = (expr);                                                                      \
        if (rc__ != 0) return rc__;                                                             \
    } while (0)

int use_device(int device);   // validates + hipSetDevice; CYMF_ERR_NO_DEVICE if none
// false once the process is inside exit() (our atexit hook has run: the HIP runtime's own teardown follows it) or when
// hipSetDevice no longer succeeds (hipErrorDeinitialized and friends).  The destroy entry points and DevBuf::release
// then return without touching HIP: handles that outlive the runtime are leaked quietly instead of crashing in exit()
// (gpurun_out/exact.log of round 1: SIGSEGV under exit() after rocprofv3's tool finalization, a trainer freed late).
bool runtime_alive(int device);
bool process_exiting();
int staging_memtype();        // CYMF_STAGING_MEMTYPE, default 2 (uncached): buffers a kernel fills and a copy engine reads, or vice versa
int default_memtype();        // CYMF_DEFAULT_MEMTYPE (0 coarse, 1 fine-grained, 2 uncached = default), read once

// ------------------------------------------------------------------ device memory
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && !process_exiting()) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    // Memory type.  -1 = the library default (default_memtype(): uncached), 0 = coarse-grained (hipMalloc),
    // 1 = hipDeviceMallocFinegrained, 2 = hipDeviceMallocUncached.
    //   * Tables that wavefronts of ONE kernel update concurrently from all 8 XCDs (HOGWILD, the exact mode's row
    //     hand-offs) must be uncached: the per-XCD L2s are not coherent with each other inside a kernel.
    //   * Everything else is uncached too, because cached buffers were observed STALE on some MI355X boxes at the
    //     seams between kernels and copies: negatives resolved from the previous epoch's draws, a download
    //     staging buffer arriving on the host with rows of another matrix, host-read draws that did not match the
    //     stream (DESIGN.md 2 lists the incidents; intermittent, per box, dependent on the allocation history of
    //     the process, gone with uncached buffers).  The kernels here stream or gather with little L2 reuse, so
    //     this costs 0-3 % (RelMF 10 %); CYMF_DEFAULT_MEMTYPE=1 / 0 restores cached buffers for experiments.
    int fine = -1;
    int alloc(size_t count) {
        if (count == n && p) return 0;
        release();
        if (count == 0) return 0;
        const int mt = fine < 0 ? default_memtype() : fine;
        hipError_t e = mt ? hipExtMallocWithFlags((void **)&p, count * sizeof(T), mt == 2 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained)
                          : hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(CYMF_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        }
        n = count;
        return 0;
    }
    // keep the allocation while it is large enough (per-epoch payloads whose size varies a little)
    int reserve(size_t count) { return (p && n >= count) ? 0 : alloc(count); }
    // H2D copies.  upload()/upload_into() return only when the source has been consumed (copy + stream sync):
    // callers hand in temporaries (std::vector staging in set_data) that die at the end of their block, and whether
    // hipMemcpyAsync has finished reading a PAGEABLE source when it returns is not something HIP promises.
    // The *_async forms skip the sync: the caller keeps `src` alive until it has synchronised `s` itself.
    int upload_into_async(const T *src, size_t count, size_t capacity, hipStream_t s) {
        CYMF_TRY(reserve(capacity > count ? capacity : count));
        if (count) CYMF_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
        return 0;
    }
    int upload_async(const T *src, size_t count, hipStream_t s) {
        CYMF_TRY(alloc(count));
        if (count) CYMF_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
        return 0;
    }
    int upload_into(const T *src, size_t count, size_t capacity, hipStream_t s = nullptr) {
        CYMF_TRY(upload_into_async(src, count, capacity, s));
        if (count) CYMF_HIP(hipStreamSynchronize(s));
        return 0;
    }
    int upload(const T *src, size_t count, hipStream_t s = nullptr) {
        CYMF_TRY(upload_async(src, count, s));
        if (count) CYMF_HIP(hipStreamSynchronize(s));
        return 0;
    }
    int zero(hipStream_t s = nullptr) {
        if (n) CYMF_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
        return 0;
    }
};

// Host loops over the triplets of a large problem (set_data: bucketing 10^8 interactions) on a few threads: f(t, begin, end) for the
// t-th of T contiguous chunks of [0, n), T = host_threads(n) <= 16 (CYMF_HOST_THREADS), the caller's thread included.
inline int host_threads(int64_t n) {
    static const int avail = [] {
        const char *e = getenv("CYMF_HOST_THREADS");
        const int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return v < 1 ? 1 : (v > 16 ? 16 : v);
    }();
    return n < ((int64_t)1 << 20) ? 1 : avail;
}
template <typename F>
void parallel_chunks(int64_t n, int T, F f) {
    if (T <= 1 || n <= 0) { f(0, (int64_t)0, n); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)T - 1);
    for (int t = 1; t < T; ++t) pool.emplace_back([&, t] { f(t, n * t / T, n * (t + 1) / T); });
    f(0, (int64_t)0, n / T);
    for (auto &th : pool) th.join();
}

// Pinned host staging (hipHostMalloc).  A hipMemcpyAsync FROM PAGEABLE memory is not asynchronous -- the runtime stages it and waits, and
// that wait was measured at 10-21 ms per epoch of 11 MB of schedule arrays on the test boxes (quantised in ~10.5 ms steps: a blocked wait on a
// timer tick, not the copy), where the same bytes from pinned memory are enqueued in microseconds and move at PCIe speed.
template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    void release() {
        if (p && !process_exiting()) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
    int reserve(size_t count) {
        if (p && n >= count) return 0;
        release();
        if (count == 0) return 0;
        hipError_t e = hipHostMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(CYMF_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        }
        n = count;
        return 0;
    }
};

// A side stream: lowest priority.  Not for the scheduling order but for the HARDWARE QUEUES: the runtime deals a process's
// streams onto a handful of hardware queues, and two streams that land on one queue run strictly one after the other -- the
// "hidden" side work then sits between the main stream's kernels (measured: rocprofv3 kernel trace of the eight-rank BPR
// schedule).  Streams of different priority never share a queue.  high = true: highest priority instead
// (the exchange of a sharded job).
inline hipError_t create_side_stream(hipStream_t *s, bool high = false) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, high ? greatest : least);
}

// ------------------------------------------------------------------ wave64 helpers (device)
#if defined(__HIPCC__)

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// DPP control words (GCN3/CDNA encoding)
constexpr int DPP_QUAD_PERM_1032 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_PERM_2301 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes, result in every lane (wave-uniform value, usable as SGPR).
// 4 DPP steps inside each row of 16 (every lane then holds its row's sum), then row_bcast:15 adds row 0 into row 1 and row 2
// into row 3, row_bcast:31 adds (row 0 + row 1) into rows 2 and 3: lane 63 holds (r3 + r2) + (r1 + r0) -- the same two pair
// sums and the same final addition as reading the four row sums with v_readlane and adding them, bit for bit, in three
// instructions instead of eight.  Inline assembly: written with update_dpp the compiler emits a v_mov_b32_dpp and a
// separate add per step; the s_nop are the two wait states a DPP read needs after the VALU write of its operand.
// `volatile`: a cross-lane instruction must execute with the EXEC mask of this point.  Several callers consume the sum only
// under `if (lane == 0)`; hipcc has sunk a non-volatile cross-lane asm into such a masked branch before
// (profiles/r02_permlane_hazard.md), where row_bcast would read lanes that are switched off.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<DPP_QUAD_PERM_1032>(v);
    v += dpp_f32<DPP_QUAD_PERM_2301>(v);
    v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f32<DPP_ROW_MIRROR>(v);
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ double wave_sum(double v) {
    // fp64 path is a test/parity path: plain butterfly through ds_bpermute is fine
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ int bcast_lane(int v, int src_lane) {   // src_lane wave-uniform
    return __builtin_amdgcn_readlane(v, src_lane);
}

#endif  // __HIPCC__

}  // namespace cymf
