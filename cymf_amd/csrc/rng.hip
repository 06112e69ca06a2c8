// rng.hip -- MT19937 + Lemire-32 on the device, bit-exact with the reference's
// UniformGenerator (cymf/math.pyx:12-18; libstdc++-11 bits/uniform_int_dist.h:241-270).
//
// One stream is inherently serial (x[k+624] = f(x[k], x[k+1], x[k+397])), but the recurrence
// has lag 227 = 624-397, so a 624-word block regenerates in three data-parallel sub-steps of
// 227/227/170 words.  One 256-thread workgroup owns a stream: state double-buffered in LDS,
// one barrier per sub-step, then tempering + Lemire multiply + ordered compaction of the rare
// rejected words (probability (2^32 mod range)/2^32 per word; a block with no rejection takes
// the fast path with no prefix scan).
#include "rng.h"

namespace cymf {

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int RNG_THREADS = 256;

__global__ void rng_seed_kernel(RngState *st, uint32_t seed) {
    // std::mt19937(seed): init_genrand recurrence, serial by definition (624 steps)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t x = seed;
        st->mt[0] = x;
        for (int i = 1; i < MT_N; ++i) {
            x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
            st->mt[i] = x;
        }
        st->idx = MT_N;
        st->pad = 0;
        st->raw_consumed = 0;
        st->draws = 0;
    }
}

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__global__ __launch_bounds__(RNG_THREADS) void rng_generate_kernel(RngState *st, uint32_t range, uint32_t thr,
                                                                  int64_t n_skip, int64_t n_out,
                                                                  uint32_t *__restrict__ out) {
    __shared__ uint32_t buf[2][MT_N];
    __shared__ int s_wcnt[RNG_THREADS / 64];
    __shared__ uint32_t s_end;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int cur = 0;
    for (int k = tid; k < MT_N; k += RNG_THREADS) buf[0][k] = st->mt[k];
    uint32_t idx = st->idx;
    uint64_t raw_consumed = st->raw_consumed;
    int64_t remaining = n_skip + n_out;   // draws still to produce (uniform across the workgroup)
    int64_t produced = 0;                 // draws produced by this launch
    __syncthreads();

    while (remaining > 0) {
        if (idx == MT_N) {
            const uint32_t *c = buf[cur];
            uint32_t *nx = buf[cur ^ 1];
            if (tid < 227) nx[tid] = c[tid + MT_M] ^ mt_mix(c[tid], c[tid + 1]);
            __syncthreads();
            if (tid < 227) { int k = tid + 227; nx[k] = nx[k - 227] ^ mt_mix(c[k], c[k + 1]); }
            __syncthreads();
            if (tid < 170) { int k = tid + 454; nx[k] = nx[k - 227] ^ mt_mix(c[k], k == MT_N - 1 ? nx[0] : c[k + 1]); }
            __syncthreads();
            cur ^= 1;
            idx = 0;
        }
        const uint32_t *c = buf[cur];
        // up to three words per thread: w = idx + tid + 256 r
        uint32_t val[3];
        bool ok[3], have[3];
        bool any_rej = false;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            uint32_t w = idx + tid + RNG_THREADS * r;
            have[r] = w < MT_N;
            uint64_t p = (uint64_t)mt_temper(have[r] ? c[w] : 0u) * (uint64_t)range;
            val[r] = (uint32_t)(p >> 32);
            ok[r] = have[r] && ((uint32_t)p >= thr);     // Lemire: redraw while low < threshold
            any_rej |= have[r] && !ok[r];
        }
        const int avail = MT_N - (int)idx;
        if (!__syncthreads_or(any_rej)) {
            const int take = remaining < (int64_t)avail ? (int)remaining : avail;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int rank = tid + RNG_THREADS * r;
                if (rank < take) {
                    int64_t d = produced + rank;
                    if (d >= n_skip) out[d - n_skip] = val[r];
                }
            }
            idx += take;
            raw_consumed += take;
            produced += take;
            remaining -= take;
        } else {
            // ordered compaction of the accepted words of this block
            int64_t base = 0;
            if (tid == 0) s_end = MT_N;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                unsigned long long m = __ballot(ok[r]);
                int lane_pre = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == 0) s_wcnt[wave] = __popcll(m);
                __syncthreads();
                int wave_off = 0, total = 0;
#pragma unroll
                for (int q = 0; q < RNG_THREADS / 64; ++q) {
                    int cnt = s_wcnt[q];
                    if (q < wave) wave_off += cnt;
                    total += cnt;
                }
                int64_t rank = base + wave_off + lane_pre;
                if (ok[r] && rank < remaining) {
                    int64_t d = produced + rank;
                    if (d >= n_skip) out[d - n_skip] = val[r];
                    if (rank == remaining - 1) s_end = idx + tid + RNG_THREADS * r + 1;   // last word consumed
                }
                base += total;
                __syncthreads();
            }
            if (base <= remaining) {
                raw_consumed += avail;
                idx = MT_N;
                produced += base;
                remaining -= base;
            } else {
                uint32_t e = s_end;
                raw_consumed += e - idx;
                idx = e;
                produced += remaining;
                remaining = 0;
            }
        }
        __syncthreads();
    }
    for (int k = tid; k < MT_N; k += RNG_THREADS) st->mt[k] = buf[cur][k];
    if (tid == 0) {
        st->idx = idx;
        st->raw_consumed = raw_consumed;
        st->draws += (uint64_t)produced;
    }
}

__global__ void widen_u32_i64_kernel(const uint32_t *__restrict__ in, int64_t *__restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (int64_t)in[i];
}

}  // namespace

int DeviceRng::init(uint32_t seed, uint64_t range, hipStream_t s) {
    if (range < 1 || range > 0xffffffffull)
        return fail(CYMF_ERR_UNSUPPORTED,
                    "index stream range %llu outside [1, 2^32-1] (the reference densifies X for RelMF, "
                    "so U*I >= 2^32 cells is not a realistic input)", (unsigned long long)range);
    CYMF_TRY(st_.alloc(1));
    range_ = (uint32_t)range;
    thr_ = (uint32_t)(0u - range_) % range_;   // (2^32 - range) mod range, bits/uniform_int_dist.h:260
    hipLaunchKernelGGL(rng_seed_kernel, dim3(1), dim3(64), 0, s, st_.p, seed);
    CYMF_HIP(hipGetLastError());
    return 0;
}

int DeviceRng::generate(int64_t n_skip, int64_t n, uint32_t *d_out, hipStream_t s) {
    if (!st_.p) return fail(CYMF_ERR_INVALID, "DeviceRng::generate before init");
    if (n_skip < 0 || n < 0) return fail(CYMF_ERR_INVALID, "negative draw count");
    if (n_skip + n == 0) return 0;
    hipLaunchKernelGGL(rng_generate_kernel, dim3(1), dim3(RNG_THREADS), 0, s, st_.p, range_, thr_, n_skip, n, d_out);
    CYMF_HIP(hipGetLastError());
    return 0;
}

}  // namespace cymf

using namespace cymf;

extern "C" int cymf_rng_fill_uniform(int device, uint32_t seed, uint64_t range, int64_t n, int64_t skip,
                                     int64_t *out) {
    if (n < 0 || skip < 0 || (n > 0 && !out)) return fail(CYMF_ERR_INVALID, "cymf_rng_fill_uniform: bad arguments");
    CYMF_TRY(use_device(device));
    DeviceRng rng;
    CYMF_TRY(rng.init(seed, range, nullptr));
    if (n == 0) return 0;
    DevBuf<uint32_t> d32;
    DevBuf<int64_t> d64;
    CYMF_TRY(d32.alloc((size_t)n));
    CYMF_TRY(d64.alloc((size_t)n));
    CYMF_TRY(rng.generate(skip, n, d32.p, nullptr));
    int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(widen_u32_i64_kernel, dim3(blocks), dim3(256), 0, nullptr, d32.p, d64.p, n);
    CYMF_HIP(hipGetLastError());
    CYMF_HIP(hipMemcpy(out, d64.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    return 0;
}
