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

#include <algorithm>

#include "mt_jump_poly.h"

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

// ---------------------------------------------------------------- ranges of 2^32 and more (64-bit draws)
// RelMF draws cells from UniformGenerator(0, U*I) on `long` (cymf/relmf.pyx:128), and libstdc++ leaves the Lemire
// branch once the range exceeds the generator's 32 bits (bits/uniform_int_dist.h:325-347):
//   range == 2^32 : one raw word is the draw;
//   range  > 2^32 : do { tmp = 2^32 * D(0, urange >> 32)(g); ret = tmp + g(); } while (ret > urange || ret < tmp),
//                   urange = range - 1, the inner draw D(0, hi_max) being the Lemire branch on hi_max + 1 < 2^32.
// A draw consumes a data-dependent number of raw words, so the block of 624 words is regenerated by the workgroup as
// above and then walked by ONE lane as a two-state machine (need high part / need low word).  A launch ends on a
// completed draw, so no state beyond the generator's own survives it.  Slow by design (~10^8 words/s): these ranges
// mean a dense X of more than 4 G cells, which the reference could not train in hours either.
__global__ __launch_bounds__(RNG_THREADS) void rng_generate_wide_kernel(RngState *st, uint64_t urange, uint32_t range_hi,
                                                                       uint32_t thr_hi, int64_t n_skip, int64_t n_out,
                                                                       uint64_t *__restrict__ out) {
    __shared__ uint32_t buf[2][MT_N];
    __shared__ uint32_t s_idx;
    __shared__ long long s_remaining, s_produced;
    __shared__ unsigned long long s_raw;
    const int tid = threadIdx.x;
    int cur = 0;
    for (int k = tid; k < MT_N; k += RNG_THREADS) buf[0][k] = st->mt[k];
    if (tid == 0) { s_idx = st->idx; s_raw = st->raw_consumed; s_remaining = n_skip + n_out; s_produced = 0; }
    __syncthreads();
    const bool direct = urange == 0xffffffffull;   // range == 2^32
    bool have_hi = false;                            // lane 0's machine state
    uint64_t tmp = 0;
    while (s_remaining > 0) {                        // uniform: s_remaining is read after a barrier
        if (s_idx == MT_N) {
            const uint32_t *c = buf[cur];
            uint32_t *nx = buf[cur ^ 1];
            if (tid < 227) nx[tid] = c[tid + MT_M] ^ mt_mix(c[tid], c[tid + 1]);
            __syncthreads();
            if (tid < 227) { int k = tid + 227; nx[k] = nx[k - 227] ^ mt_mix(c[k], c[k + 1]); }
            __syncthreads();
            if (tid < 170) { int k = tid + 454; nx[k] = nx[k - 227] ^ mt_mix(c[k], k == MT_N - 1 ? nx[0] : c[k + 1]); }
            __syncthreads();
            cur ^= 1;
            if (tid == 0) s_idx = 0;
            __syncthreads();
        }
        if (tid == 0) {
            const uint32_t *c = buf[cur];
            uint32_t idx = s_idx;
            long long remaining = s_remaining, produced = s_produced;
            unsigned long long raw = s_raw;
            while (idx < MT_N && remaining > 0) {
                const uint32_t w = mt_temper(c[idx++]);
                ++raw;
                uint64_t ret;
                if (direct) {
                    ret = w;
                } else if (!have_hi) {
                    const uint64_t p = (uint64_t)w * (uint64_t)range_hi;
                    if ((uint32_t)p < thr_hi) continue;          // Lemire redraw of the high part
                    tmp = (p >> 32) << 32;
                    have_hi = true;
                    continue;
                } else {
                    have_hi = false;
                    ret = tmp + w;
                    if (ret > urange || ret < tmp) continue;     // the pair is rejected as a whole
                }
                if (produced >= n_skip) out[produced - n_skip] = ret;
                ++produced;
                --remaining;
            }
            s_idx = idx; s_remaining = remaining; s_produced = produced; s_raw = raw;
        }
        __syncthreads();
    }
    for (int k = tid; k < MT_N; k += RNG_THREADS) st->mt[k] = buf[cur][k];
    if (tid == 0) {
        st->idx = s_idx;
        st->raw_consumed = s_raw;
        st->draws += (uint64_t)s_produced;
    }
}

// ---------------------------------------------------------------- parallel mode
// Jump ahead by J = MT_JUMP_WORDS raw words: state(n+J) = XOR over the set bits k of g of
// state(n+k), evaluated on the recurrence's own word sequence x (x_0..x_623 = state):
// out[m] = XOR_k g_k x[k+m].  The 624 + 19937 words live in LDS (82 KB).
constexpr int JUMP_THREADS = 1024;
constexpr int JUMP_DEG = 19937;
constexpr int JUMP_LDS_WORDS = MT_N + JUMP_DEG + 3;

__global__ __launch_bounds__(JUMP_THREADS) void mt_jump_kernel(const uint32_t *__restrict__ in_state,
                                                              uint32_t *__restrict__ out_state,
                                                              const uint32_t *__restrict__ poly) {
    extern __shared__ uint32_t x[];
    const int tid = threadIdx.x;
    in_state += (size_t)blockIdx.x * MT_N;        // workgroup q: state q of the input run -> state q of the output run
    out_state += (size_t)blockIdx.x * MT_N;
    for (int k = tid; k < MT_N; k += JUMP_THREADS) x[k] = in_state[k];
    __syncthreads();
    for (int b = 0; b < JUMP_DEG; b += 227) {     // lag 227: one data-parallel slab per barrier
        const int k = b + tid;
        if (tid < 227 && k < JUMP_DEG) x[k + MT_N] = x[k + MT_M] ^ mt_mix(x[k], x[k + 1]);
        __syncthreads();
    }
    if (tid < MT_N) {
        uint32_t acc = 0;
        for (int w = 0; w < MT_N; ++w) {
            uint32_t bits = poly[w];              // wave-uniform
            const int base = w * 32 + tid;
            while (bits) {
                const int k = __builtin_ctz(bits);
                bits &= bits - 1;
                acc ^= x[base + k];
            }
        }
        out_state[tid] = acc;
    }
}

constexpr int REJ_CAP_MIN = 2048;   // recorded rejection offsets per chunk (at least; sized from the range's rejection rate)

// Workgroup q generates chunk (c0 + q) of the raw stream from its start state and writes the
// accepted words (Lemire) compacted to tmp[q*J ...]; words before skip0 belong to an earlier call
// (chunk 0 only).  Rejected raw offsets are recorded so that the host can settle the stream position.
__global__ __launch_bounds__(RNG_THREADS) void rng_chunk_kernel(const uint32_t *__restrict__ states, uint32_t skip0,
                                                               uint32_t range, uint32_t thr,
                                                               uint32_t *__restrict__ tmp, uint32_t *__restrict__ counts,
                                                               uint32_t *__restrict__ rej, uint32_t *__restrict__ rej_cnt,
                                                               uint32_t rej_cap, uint32_t limit) {
    __shared__ uint32_t buf[2][MT_N];
    __shared__ int s_wcnt[RNG_THREADS / 64];
    __shared__ uint32_t s_nrej;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const uint32_t first = q == 0 ? skip0 : 0u;
    uint32_t *__restrict__ out = tmp + (size_t)q * MT_JUMP_WORDS;
    uint32_t *__restrict__ myrej = rej + (size_t)q * rej_cap;
    for (int k = tid; k < MT_N; k += RNG_THREADS) buf[0][k] = states[(size_t)q * MT_N + k];
    if (tid == 0) s_nrej = 0;
    int cur = 0;
    uint32_t base = 0;   // accepted so far (uniform)
    __syncthreads();
    for (int blk = 0; blk < MT_JUMP_BLOCKS; ++blk) {
        if (base >= limit) break;                  // the call asked for no more than `limit` draws in all (uniform)
        const uint32_t *c = buf[cur];
        uint32_t *nx = buf[cur ^ 1];
        if (tid < 227) nx[tid] = c[tid + MT_M] ^ mt_mix(c[tid], c[tid + 1]);
        __syncthreads();
        if (tid < 227) { int k = tid + 227; nx[k] = nx[k - 227] ^ mt_mix(c[k], c[k + 1]); }
        __syncthreads();
        if (tid < 170) { int k = tid + 454; nx[k] = nx[k - 227] ^ mt_mix(c[k], k == MT_N - 1 ? nx[0] : c[k + 1]); }
        __syncthreads();
        cur ^= 1;
        const uint32_t off0 = (uint32_t)blk * MT_N;
        if (off0 + MT_N <= first) continue;        // whole block consumed by an earlier call (uniform)
        const uint32_t *g = buf[cur];
        uint32_t val[3];
        bool ok[3];
        bool special = false;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint32_t w = tid + RNG_THREADS * r;
            const bool have = w < MT_N;
            const uint64_t p = (uint64_t)mt_temper(have ? g[w] : 0u) * (uint64_t)range;
            val[r] = (uint32_t)(p >> 32);
            const bool live = have && off0 + w >= first;
            const bool acc = (uint32_t)p >= thr;
            ok[r] = live && acc;
            special |= have && !ok[r];
            if (live && !acc) {
                const uint32_t slot = atomicAdd(&s_nrej, 1u);
                if (slot < rej_cap) myrej[slot] = off0 + w;
            }
        }
        if (!__syncthreads_or(special)) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const uint32_t w = tid + RNG_THREADS * r;
                if (w < MT_N) out[base + w] = val[r];
            }
            base += MT_N;
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const unsigned long long m = __ballot(ok[r]);
                const int lane_pre = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == 0) s_wcnt[wave] = __popcll(m);
                __syncthreads();
                int wave_off = 0, total = 0;
#pragma unroll
                for (int v = 0; v < RNG_THREADS / 64; ++v) {
                    const int cnt = s_wcnt[v];
                    if (v < wave) wave_off += cnt;
                    total += cnt;
                }
                if (ok[r]) out[base + wave_off + lane_pre] = val[r];
                base += total;
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        counts[q] = base;
        rej_cnt[q] = s_nrej;
    }
}

// out[d - n_skip] = d-th accepted word over the chunks, for d in [n_skip, n_total)
__global__ __launch_bounds__(256) void rng_gather_kernel(const uint32_t *__restrict__ tmp, const uint32_t *__restrict__ counts,
                                                        int n_chunks, int64_t n_skip, int64_t n_total,
                                                        uint32_t *__restrict__ out) {
    __shared__ int64_t s_prefix;
    const int q = blockIdx.y;
    if (threadIdx.x == 0) {
        int64_t p = 0;
        for (int v = 0; v < q; ++v) p += counts[v];
        s_prefix = p;
    }
    __syncthreads();
    const int64_t prefix = s_prefix;
    const int64_t cnt = counts[q];
    const uint32_t *__restrict__ src = tmp + (size_t)q * MT_JUMP_WORDS;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < cnt; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = prefix + r;
        if (d >= n_skip && d < n_total) out[d - n_skip] = src[r];
    }
}

__global__ void widen_u32_i64_kernel(const uint32_t *__restrict__ in, int64_t *__restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (int64_t)in[i];
}

}  // namespace

DeviceRng::~DeviceRng() {
    if (!process_exiting())
        for (uint32_t *p : retired_) (void)hipFree(p);
    if (pend_.done) (void)hipEventDestroy(pend_.done);
    if (pend_.h_counts) (void)hipHostFree(pend_.h_counts);
    if (pend_.h_rej) (void)hipHostFree(pend_.h_rej);
    if (pend_.h_rej_cnt) (void)hipHostFree(pend_.h_rej_cnt);
}

int DeviceRng::init(uint32_t seed, uint64_t range, hipStream_t s, bool parallel) {
    if (range < 1 || range > ((uint64_t)1 << 62))
        return fail(CYMF_ERR_INVALID, "index stream range %llu outside [1, 2^62]", (unsigned long long)range);
    CYMF_TRY(st_.alloc(1));
    if (range > 0xffffffffull) {   // 64-bit draws: the one-lane walker (rng_generate_wide_kernel)
        wide_ = true;
        urange_ = range - 1;
        range_ = (uint32_t)((urange_ >> 32) + 1);            // range of the high part D(0, urange >> 32); 1 when range == 2^32
        thr_ = (uint32_t)(0u - range_) % range_;
        parallel_ = false;
        hipLaunchKernelGGL(rng_seed_kernel, dim3(1), dim3(64), 0, s, st_.p, seed);
        CYMF_HIP(hipGetLastError());
        return 0;
    }
    range_ = (uint32_t)range;
    thr_ = (uint32_t)(0u - range_) % range_;   // (2^32 - range) mod range, bits/uniform_int_dist.h:260
    hipLaunchKernelGGL(rng_seed_kernel, dim3(1), dim3(64), 0, s, st_.p, seed);
    CYMF_HIP(hipGetLastError());
    // the per-chunk rejection list must hold the expected rejections with a wide margin
    // (a range like U*I = 1.6e8 of RelMF rejects 3 % of the words: 130 k per chunk)
    const double rej_per_chunk = (double)thr_ / 4294967296.0 * (double)MT_JUMP_WORDS;
    rej_cap_ = REJ_CAP_MIN;
    while ((double)rej_cap_ < rej_per_chunk * 1.5 + 1024.0) rej_cap_ *= 2;
    parallel_ = parallel && rej_cap_ <= (1 << 19);   // beyond ~8 % rejections the one-workgroup walker serves (2 MB of list per chunk)
    if (parallel_) {
        CYMF_TRY(poly_.upload(&MT_JUMP_POLYS[0][0], (size_t)MT_JUMP_LEVELS * MT_N, s));
        states_cap_ = 64;
        CYMF_TRY(states_.alloc((size_t)states_cap_ * MT_N));
        // chunk 0 starts at the seeded state (RngState begins with mt[624])
        CYMF_HIP(hipMemcpyAsync(states_.p, st_.p, MT_N * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        states_known_ = 1;
        raw_pos_ = 0;
        CYMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mt_jump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(JUMP_LDS_WORDS * sizeof(uint32_t))));
        if (!pend_.done) CYMF_HIP(hipEventCreateWithFlags(&pend_.done, hipEventDisableTiming));
    }
    return 0;
}

int DeviceRng::ensure_states(int64_t last_chunk, hipStream_t s) {
    // A launch of the widest level costs the same ~0.45 ms for one state as for 256 (one workgroup per CU): once that level is
    // available the run is always extended to the next multiple of 256 states, so that an epoch of n chunks costs n / 256
    // launches instead of one or two partial ones per call (RelMF 20000 x 8000: 635 chunks per epoch, 5 launches -> 2.5).
    if (states_known_ >= 256) last_chunk = ((last_chunk + 256) / 256) * 256 - 1;
    if (last_chunk + 1 > states_cap_) {
        int64_t cap = states_cap_;
        while (cap < last_chunk + 1) cap *= 2;
        DevBuf<uint32_t> bigger;
        CYMF_TRY(bigger.alloc((size_t)cap * MT_N));
        CYMF_HIP(hipMemcpyAsync(bigger.p, states_.p, (size_t)states_known_ * MT_N * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        CYMF_HIP(hipStreamSynchronize(s));
        std::swap(bigger.p, states_.p);
        std::swap(bigger.n, states_.n);
        retired_.push_back(bigger.p);   // (not freed here: see rng.h)
        bigger.p = nullptr; bigger.n = 0;
        states_cap_ = cap;
    }
    // Level l jumps by D = 4^l chunks: a run of up to D states follows from the run D chunks before it in ONE launch, one
    // workgroup per state (~0.5 ms).  The largest D <= states_known_ is taken each time: 3 launches per level reach 256 states,
    // 256 more per launch after that (one workgroup per CU: the 82 KB of LDS allow no second one).  (A chain of single jumps, one launch per chunk, was the first version: 11 ms per 100 M
    // draws -- hidden under the step kernels on one GPU, but every rank of a sharded job generates the whole stream while its
    // own steps shrink with the world size.)
    while (states_known_ <= last_chunk) {
        int lvl = 0;
        int64_t D = 1;
        while (lvl + 1 < MT_JUMP_LEVELS && D * 4 <= states_known_) { D *= 4; ++lvl; }
        const int n = (int)std::min<int64_t>(D, last_chunk + 1 - states_known_);
        hipLaunchKernelGGL(mt_jump_kernel, dim3(n), dim3(JUMP_THREADS), JUMP_LDS_WORDS * sizeof(uint32_t), s,
                           states_.p + (size_t)(states_known_ - D) * MT_N, states_.p + (size_t)states_known_ * MT_N,
                           poly_.p + (size_t)lvl * MT_N);
        CYMF_HIP(hipGetLastError());
        states_known_ += n;
    }
    return 0;
}

int DeviceRng::finalize() {
    if (!pend_.active) return 0;
    CYMF_HIP(hipEventSynchronize(pend_.done));
    pend_.active = false;
    int64_t prefix = 0;
    for (int64_t q = 0; q < pend_.n_chunks; ++q) {
        const int64_t cnt = pend_.h_counts[q];
        if (pend_.h_rej_cnt[q] > (uint32_t)rej_cap_)
            return fail(CYMF_ERR_UNSUPPORTED, "index stream: %u rejections in one chunk exceed the parallel generator's list",
                        pend_.h_rej_cnt[q]);
        if (prefix + cnt >= pend_.n_total) {
            // the last requested draw is the accepted word of rank r in chunk q: its raw offset is
            // r (+ skip0 in chunk 0) plus the rejected words at or before it
            const int64_t r = pend_.n_total - 1 - prefix;
            uint64_t pos = (uint64_t)r + (q == 0 ? pend_.skip0 : 0);
            uint32_t *lst = pend_.h_rej + (size_t)q * rej_cap_;
            const uint32_t nr = pend_.h_rej_cnt[q];
            std::sort(lst, lst + nr);
            for (uint32_t i = 0; i < nr && (uint64_t)lst[i] <= pos; ++i) ++pos;
            raw_pos_ = (uint64_t)(pend_.c0 + q) * (uint64_t)MT_JUMP_WORDS + pos + 1;
            return 0;
        }
        prefix += cnt;
    }
    return fail(CYMF_ERR_HIP, "index stream: chunked generator produced %lld of %lld draws", (long long)prefix,
                (long long)pend_.n_total);
}

int DeviceRng::generate_parallel(int64_t n_total, int64_t n_skip, uint32_t *d_out, hipStream_t s) {
    CYMF_TRY(finalize());
    const int64_t J = MT_JUMP_WORDS;
    const int64_t c0 = (int64_t)(raw_pos_ / (uint64_t)J);
    const uint64_t skip0 = raw_pos_ % (uint64_t)J;
    const double p_rej = (double)thr_ / 4294967296.0;
    const int64_t need_raw = n_total + (int64_t)((double)n_total * p_rej * 1.5) + 4096;
    const int64_t n_chunks = ((int64_t)skip0 + need_raw + J - 1) / J;
    CYMF_TRY(ensure_states(c0 + n_chunks - 1, s));
    CYMF_TRY(tmp_.reserve((size_t)n_chunks * (size_t)J));   // kept while large enough: the chunk count moves by one between epochs, and a hipFree is a device-wide sync
    if (n_chunks > pend_.cap_chunks) {
        if (pend_.h_counts) (void)hipHostFree(pend_.h_counts);
        if (pend_.h_rej) (void)hipHostFree(pend_.h_rej);
        if (pend_.h_rej_cnt) (void)hipHostFree(pend_.h_rej_cnt);
        pend_.cap_chunks = n_chunks * 2;
        CYMF_HIP(hipHostMalloc((void **)&pend_.h_counts, (size_t)pend_.cap_chunks * sizeof(uint32_t)));
        CYMF_HIP(hipHostMalloc((void **)&pend_.h_rej_cnt, (size_t)pend_.cap_chunks * sizeof(uint32_t)));
        CYMF_HIP(hipHostMalloc((void **)&pend_.h_rej, (size_t)pend_.cap_chunks * rej_cap_ * sizeof(uint32_t)));
        CYMF_TRY(counts_.alloc((size_t)pend_.cap_chunks));
        CYMF_TRY(rej_cnt_.alloc((size_t)pend_.cap_chunks));
        CYMF_TRY(rej_.alloc((size_t)pend_.cap_chunks * rej_cap_));
    }
    hipLaunchKernelGGL(rng_chunk_kernel, dim3((unsigned)n_chunks), dim3(RNG_THREADS), 0, s, states_.p + (size_t)c0 * MT_N,
                       (uint32_t)skip0, range_, thr_, tmp_.p, counts_.p, rej_.p, rej_cnt_.p, (uint32_t)rej_cap_,
                       (uint32_t)std::min<int64_t>(n_total, 0xffffffffll));
    CYMF_HIP(hipGetLastError());
    if (d_out && n_total > n_skip) {
        hipLaunchKernelGGL(rng_gather_kernel, dim3(64, (unsigned)n_chunks), dim3(256), 0, s, tmp_.p, counts_.p, (int)n_chunks,
                           n_skip, n_total, d_out);
        CYMF_HIP(hipGetLastError());
    }
    CYMF_HIP(hipMemcpyAsync(pend_.h_counts, counts_.p, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CYMF_HIP(hipMemcpyAsync(pend_.h_rej_cnt, rej_cnt_.p, (size_t)n_chunks * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CYMF_HIP(hipMemcpyAsync(pend_.h_rej, rej_.p, (size_t)n_chunks * rej_cap_ * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CYMF_HIP(hipEventRecord(pend_.done, s));
    pend_.active = true;
    pend_.c0 = c0; pend_.n_chunks = n_chunks; pend_.n_total = n_total; pend_.skip0 = skip0;
    return 0;
}

int DeviceRng::generate64(int64_t n_skip, int64_t n, uint64_t *d_out, hipStream_t s) {
    if (!st_.p || !wide_) return fail(CYMF_ERR_INVALID, "DeviceRng::generate64 needs a stream initialised with range >= 2^32");
    if (n_skip < 0 || n < 0) return fail(CYMF_ERR_INVALID, "negative draw count");
    if (n_skip + n == 0) return 0;
    hipLaunchKernelGGL(rng_generate_wide_kernel, dim3(1), dim3(RNG_THREADS), 0, s, st_.p, urange_, range_, thr_, n_skip, n, d_out);
    CYMF_HIP(hipGetLastError());
    return 0;
}

int DeviceRng::generate(int64_t n_skip, int64_t n, uint32_t *d_out, hipStream_t s) {
    if (!st_.p) return fail(CYMF_ERR_INVALID, "DeviceRng::generate before init");
    if (wide_) return fail(CYMF_ERR_INVALID, "DeviceRng::generate: this stream draws 64-bit values (generate64)");
    if (n_skip < 0 || n < 0) return fail(CYMF_ERR_INVALID, "negative draw count");
    if (n_skip + n == 0) return 0;
    if (parallel_) return generate_parallel(n_skip + n, n_skip, d_out, s);
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
    CYMF_TRY(rng.init(seed, range, nullptr, /*parallel=*/skip + n >= (int64_t)4 << 20));
    if (n == 0) return 0;
    if (rng.wide()) {   // range >= 2^32: 64-bit draws straight into the output type
        DevBuf<uint64_t> w;
        CYMF_TRY(w.alloc((size_t)n));
        CYMF_TRY(rng.generate64(skip, n, w.p, nullptr));
        CYMF_HIP(hipMemcpy(out, w.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
        return 0;
    }
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
