// relmf_tiles.hip -- RelMF lock-free mode without atomics on HBM: a stratified tile schedule.
// Replaces the hot loop of RelMF._fit_relmf, cymf/relmf.pyx:142-148 (model: cymf/model.pyx:99-142), in the
// reference's num_threads > 1 regime (cymf/relmf.pyx:143 prange: unordered, lock-free).
//
// An epoch is U*I cell draws of ONE mt19937 stream (cymf/relmf.pyx:128,144-146).  The previous kernel bucketed them
// by user and added every item-row delta with float atomics on HBM: bound by the atomic rate (1.3 TB/s), not by
// memory.  Here the cells are cut into B x B tiles (user block b, item block b'); sub-step s runs the B tiles
// (b, (b+s) mod B), which share no row: one workgroup per tile keeps the tile's item rows in LDS and hands each of its
// wavefronts whole users (row and optimizer state in registers while the user's draws of the tile are applied).
// Inside the workgroup the wavefronts do share the item rows: an update is added with a compare-and-swap on the LDS
// word -- gfx950's ds_add_f32 serialises per lane (192 cycles per wave instruction, tools/micro/lds_atomics.hip),
// integer LDS atomics run at the full LDS rate -- so no update is lost and none is ever applied to HBM concurrently.
// The rows reach HBM once per tile, not once per draw.
//
//   bucketing (side stream, one epoch ahead): two counting-sort passes, by user block and, inside each, by item
//     block; every pass sorts a 16 K-cell segment in LDS first and writes each bucket's run contiguously.
//   tile kernel: sort the tile's ~ub*ib draws by user in LDS, then every wavefront walks its users.
#include "relmf_tiles.h"

#include <algorithm>
#include <cstdlib>

namespace cymf {
namespace {

constexpr int BK_THREADS = 512;                      // (with 16 cells per thread: 96 VGPRs -- two such wavefronts per SIMD fit beside a tile workgroup's four)
constexpr int BK_CPT = 16;                       // cells per thread and segment
constexpr int BK_SEG = BK_THREADS * BK_CPT;      // 8192 cells = 32 KB of LDS

__device__ __forceinline__ uint32_t cell_digit(uint32_t c, uint32_t I, uint32_t ub, uint32_t ib, int pass) {
    const uint32_t u = c / I;
    return pass == 1 ? u / ub : (c - u * I) / ib;
}

// exclusive prefix of v over the threads of the workgroup (s_wtot: 32 words of LDS); *total = sum over all threads
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wtot, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    __syncthreads();                       // s_wtot may still be read from an earlier call
    if (lane == 63) s_wtot[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, all = 0;
    for (int q = 0; q < nw; ++q) {
        const uint32_t t = s_wtot[q];
        if (q < wave) wbase += t;
        all += t;
    }
    if (total) *total = all;
    return wbase + inc - v;
}

// ---------------------------------------------------------------- bucketing: count / scan / scatter
// PASS 1: the whole stream, digit = user block.  PASS 2: bucket y of pass 1 (blockIdx.y), digit = item block.
template <int PASS>
__global__ __launch_bounds__(BK_THREADS) void tile_count_kernel(const uint32_t *__restrict__ in, uint32_t n_all,
                                                               const uint32_t *__restrict__ off1, uint32_t I, uint32_t ub,
                                                               uint32_t ib, int B, uint32_t *__restrict__ gcnt) {
    extern __shared__ uint32_t s_cnt[];   // [B]
    const int tid = threadIdx.x, y = blockIdx.y;
    const uint32_t lo = PASS == 1 ? 0u : off1[y], hi = PASS == 1 ? n_all : off1[y + 1];
    const int64_t len = (int64_t)hi - lo;
    for (int64_t seg = blockIdx.x; seg * BK_SEG < len; seg += gridDim.x) {
        for (int k = tid; k < B; k += BK_THREADS) s_cnt[k] = 0u;
        __syncthreads();
        const int64_t base = (int64_t)lo + seg * BK_SEG;
#pragma unroll 4
        for (int q = 0; q < BK_CPT; ++q) {
            const int64_t t = base + (int64_t)q * BK_THREADS + tid;
            if (t < hi) atomicAdd(&s_cnt[cell_digit(in[t], I, ub, ib, PASS)], 1u);
        }
        __syncthreads();
        for (int k = tid; k < B; k += BK_THREADS) {
            const uint32_t v = s_cnt[k];
            if (v) atomicAdd(gcnt + (size_t)y * B + k, v);
        }
        __syncthreads();
    }
}

// row y of cnt[rows][n]: off[y*n + k] = base[y] + exclusive prefix; cur = off; the last row also closes off[rows*n]
__global__ __launch_bounds__(64) void tile_scan_kernel(const uint32_t *__restrict__ cnt, int n, const uint32_t *__restrict__ base_arr,
                                                      uint32_t *__restrict__ off, uint32_t *__restrict__ cur) {
    const int y = blockIdx.x, lane = threadIdx.x;
    const uint32_t *c = cnt + (size_t)y * n;
    const int per = (n + 63) / 64, k0 = lane * per, k1 = k0 + per < n ? k0 + per : n;
    uint32_t sum = 0;
    for (int k = k0; k < k1; ++k) sum += c[k];
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    uint32_t run = (base_arr ? base_arr[y] : 0u) + inc - sum;
    for (int k = k0; k < k1; ++k) {
        off[(size_t)y * n + k] = run;
        cur[(size_t)y * n + k] = run;
        run += c[k];
    }
    if (y == (int)gridDim.x - 1 && lane == 63) off[(size_t)gridDim.x * n] = (base_arr ? base_arr[y] : 0u) + inc;
}

template <int PASS>
__global__ __launch_bounds__(BK_THREADS) void tile_scatter_kernel(const uint32_t *__restrict__ in, uint32_t n_all,
                                                                 const uint32_t *__restrict__ off1, uint32_t I, uint32_t ub,
                                                                 uint32_t ib, int B, uint32_t *__restrict__ gcur,
                                                                 uint32_t *__restrict__ out) {
    extern __shared__ uint32_t sm[];
    uint32_t *s_cnt = sm;                 // [B]  counts, then ranks are taken from it
    uint32_t *s_loff = s_cnt + B;         // [B]  bucket start inside the segment
    uint32_t *s_gbase = s_loff + B;       // [B]  bucket start in the output, this segment's share
    uint32_t *s_wtot = s_gbase + B;       // [32]
    uint32_t *s_sorted = s_wtot + 32;     // [BK_SEG]
    const int tid = threadIdx.x, y = blockIdx.y;
    const uint32_t lo = PASS == 1 ? 0u : off1[y], hi = PASS == 1 ? n_all : off1[y + 1];
    const int64_t len = (int64_t)hi - lo;
    gcur += (size_t)y * B;
    const int per = (B + BK_THREADS - 1) / BK_THREADS;   // buckets per thread in the scan (B <= 4096: <= 4)
    for (int64_t seg = blockIdx.x; seg * BK_SEG < len; seg += gridDim.x) {
        for (int k = tid; k < B; k += BK_THREADS) s_cnt[k] = 0u;
        __syncthreads();
        const int64_t base = (int64_t)lo + seg * BK_SEG;
        const int seg_len = (int)((int64_t)hi - base < BK_SEG ? (int64_t)hi - base : BK_SEG);
        uint32_t cell[BK_CPT], rank[BK_CPT];
#pragma unroll
        for (int q = 0; q < BK_CPT; ++q) {
            const int t = q * BK_THREADS + tid;
            if (t < seg_len) {
                cell[q] = in[base + t];
                rank[q] = atomicAdd(&s_cnt[cell_digit(cell[q], I, ub, ib, PASS)], 1u);
            }
        }
        __syncthreads();
        {   // exclusive scan of the counts; reserve this segment's share of every bucket in the output
            uint32_t sum = 0;
            for (int j = 0; j < per; ++j) { const int k = tid * per + j; if (k < B) sum += s_cnt[k]; }
            uint32_t run = block_excl_scan(sum, s_wtot, nullptr);
            for (int j = 0; j < per; ++j) {
                const int k = tid * per + j;
                if (k < B) {
                    const uint32_t v = s_cnt[k];
                    s_loff[k] = run;
                    run += v;
                    s_gbase[k] = v ? atomicAdd(gcur + k, v) : 0u;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < BK_CPT; ++q) {
            const int t = q * BK_THREADS + tid;
            if (t < seg_len) s_sorted[s_loff[cell_digit(cell[q], I, ub, ib, PASS)] + rank[q]] = cell[q];
        }
        __syncthreads();
        for (int idx = tid; idx < seg_len; idx += BK_THREADS) {   // consecutive threads write consecutive words of a bucket's run
            const uint32_t c = s_sorted[idx];
            const uint32_t dg = cell_digit(c, I, ub, ib, PASS);
            out[(size_t)s_gbase[dg] + ((uint32_t)idx - s_loff[dg])] = c;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- the tile kernel
struct RelTileDev {
    RelTileParams p;
    int32_t U, I, K, B, ub, ib;
};

// *addr += delta, losslessly, where the caller read `expected` there before: one ds_cmpst_rtn_b32 when nobody else
// touched the word meanwhile, otherwise retried on the value found (the delta was formed from a slightly older row:
// that is the lock-free mode's bounded staleness; it is never dropped)
__device__ __forceinline__ void lds_add_cas(float *addr, float expected, float delta) {
    unsigned int *a = reinterpret_cast<unsigned int *>(addr);
    unsigned int old = __float_as_uint(expected);
    while (true) {
        const unsigned int want = __float_as_uint(__uint_as_float(old) + delta);
        const unsigned int seen = atomicCAS(a, old, want);
        if (seen == old) break;
        old = seen;
    }
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int LPD>
__device__ __forceinline__ uint32_t row_or(uint32_t v) {   // OR over the LPD (16 or 8) lanes of a worker, in every lane of it
    v |= dpp_u32<DPP_QUAD_PERM_1032>(v);
    v |= dpp_u32<DPP_QUAD_PERM_2301>(v);
    v |= dpp_u32<DPP_ROW_HALF_MIRROR>(v);
    if constexpr (LPD == 16) v |= dpp_u32<DPP_ROW_MIRROR>(v);
    return v;
}
template <int LPD>
__device__ __forceinline__ float row_sum(float v) {        // sum over the LPD lanes of a worker, in every lane of it
    v += dpp_f32<DPP_QUAD_PERM_1032>(v);
    v += dpp_f32<DPP_QUAD_PERM_2301>(v);
    v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);
    if constexpr (LPD == 16) v += dpp_f32<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ unsigned long long pack2(float a, float b) {
    return ((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a);
}

// One workgroup = one tile (user block b, item block (b + shift) mod B).
//   1. the tile's item rows (+ optimizer state) and its cells' q = x / max(p, M) go to LDS;
//   2. the tile's draws are COUNTED per cell (packed byte counters, integer LDS atomics): a user's draws of the tile are
//      then "item il, cnt[il] times", which a worker walks as bit masks (items with count >= c, c = 1, 2, ...) -- no
//      sorted list, no dependent LDS read to learn the next item;
//   3. row workers: a wavefront is four ROWS of 16 lanes (the DPP row), each row a worker that owns one user at a time.
//      The user's factor row and optimizer state sit in the row's registers (element k = 64 j + 4 l + t of lane l: one
//      ds_read_b128 per 64 elements; the hardware's b128 lane groups interleave the rows without bank conflicts), the
//      dot product is four DPP steps inside the row, and everything per draw (q, y, gradient scale, loss) is per-lane
//      arithmetic: ~20 instructions per draw where a whole wavefront per draw spent ~70 (64 lanes with one element each;
//      the rest was reduction and scalar bookkeeping).  Workers take the block's users round-robin, one round at a time.
//      Software pipeline: the next user's rows come from HBM while the current one is processed; the next draw's item
//      row and q are read from LDS before the current draw is computed (two register sets used alternately, so that no
//      copy forces the wait early); a draw's compare-and-swaps are checked after the next draw's dot product.
//      Lanes past K compute on zeros inside the rows' padding (no masking in the loop).
//   4. the item rows go back to HBM once.
template <int R, int OPT, int MAXT, int LPD>
__global__ __launch_bounds__(MAXT) void relmf_tile_kernel(RelTileDev d, const uint32_t *__restrict__ sorted,
                                                        const uint32_t *__restrict__ toff, int shift,
                                                        double *__restrict__ loss_acc, int *__restrict__ err,
                                                        long long *__restrict__ stamps) {
    constexpr int NS = opt_num_states(OPT);
    constexpr int RS = R * 64;                       // LDS row stride (floats)
    constexpr int CH = 64 / LPD;                     // consecutive row elements per lane and 64-element block (4: one b128, 8: two)
    constexpr int EPL = CH * R;                      // row elements per lane
    constexpr int WPW = 64 / LPD;                    // workers per wavefront
    constexpr int WPL = CH / 4;                      // words of byte counters per lane (CH items per lane)
    constexpr float SFILL = OPT == CYMF_OPT_ADAGRAD ? 1.0f : 0.0f;   // lanes past K of state rows (rows.h: Row::load)
    // FX (SGD, AdaGrad): the tile's item rows live in LDS as FIXED-POINT integers and every update is an integer LDS atomic add --
    // commutative, lossless, at the full LDS rate (2 cycles per CU instruction, tools/micro/lds_atomics.hip) and without a return
    // value to wait for.  The float version added a row's delta with a compare-and-swap on the words it had READ a step earlier
    // (the software pipeline): with 40 workers on 32 item rows somebody else had touched the row in between for 62 % of the
    // swaps (counted: 2.47 retries per lane and draw with AdaGrad), each retry a dependent LDS round trip.
    //   rows: int32 with ONE power-of-two scale per item row, chosen when the tile is loaded so that the row's largest entry
    //     (floored at 2^-6) lies in [2^27, 2^28): resolution max|h| * 2^-28 -- below float32's own for every entry of the row --
    //     with a factor of eight of head-room inside the tile visit (an entry that ends beyond 2^30 fails the epoch through
    //     err = 3 rather than risk a wrap); block floating point instead of one fixed scale because the initial factors are
    //     ~1e-3 / K and updates of 1e-9 must still add up (the lr -> 0 test of tests/test_gpu_models.py);
    //   AdaGrad's accumulator, a sum that only grows and may grow a thousandfold inside one visit: int64 in units of 2^-32
    //     (ds_add_u64: measurably slower than the 32-bit adds -- rows in int64 too cost 22.3 instead of 19 ms per epoch).
    // Adam's moments are not additive: its rows stay float with the compare-and-swap path below.
    constexpr bool FX = OPT != CYMF_OPT_ADAM;
    constexpr float FX_S = 4294967296.0f, FX_FRAC = 2.3283064365386963e-10f;   // the int64 accumulators' unit
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, n_waves = nthreads >> 6;
    const int b = blockIdx.x, bp = (b + shift) % d.B;
    const int u0 = b * d.ub, i0 = bp * d.ib;
    const int nu = d.U - u0 < d.ub ? d.U - u0 : d.ub, ni = d.I - i0 < d.ib ? d.I - i0 : d.ib;
    if (nu <= 0 || ni <= 0) return;
    const uint32_t t0 = toff[(size_t)b * d.B + bp], t1 = toff[(size_t)b * d.B + bp + 1];
    if (t0 == t1) return;
    // developer timing (CYMF_RELMF_TILE_STAMPS=1): shader-clock stamps of workgroup 0's phases, wave 0 and the last wave
    auto stamp = [&](int slot) {
        if (stamps && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == n_waves - 1))
            stamps[(wave == 0 ? 0 : 8) + slot] = (long long)__builtin_readcyclecounter();
    };
    stamp(0);
    const int K = d.K;
    float *sh = reinterpret_cast<float *>(smem);                 // [ib][RS] item rows (FX: int32 block floating point)
    int *shi = reinterpret_cast<int *>(smem);
    float *ss0 = sh + (size_t)d.ib * RS;                          // [ib][RS] optimizer state 0 (NS >= 1; AdaGrad: int64 fixed point)
    long long *sa64 = reinterpret_cast<long long *>(ss0);
    constexpr int S0W = (OPT == CYMF_OPT_ADAGRAD) ? 2 : (NS >= 1 ? 1 : 0);   // 4-byte words per element of state 0
    float *ss1 = ss0 + (size_t)S0W * d.ib * RS;                   // [ib][RS] optimizer state 1 (NS == 2)
    float *s_q = ss1 + (NS == 2 ? (size_t)d.ib * RS : 0);         // [ub][ib]
    uint32_t *s_cc = reinterpret_cast<uint32_t *>(s_q + (size_t)d.ub * d.ib);   // [ub][16] words = [ub][16 lanes][4 byte counters]
    float2 *s_scale = reinterpret_cast<float2 *>(s_cc + (size_t)d.ub * 16);     // [ib] FX: (scale, 1 / scale) of every item row
    uint32_t *s_rowmax = reinterpret_cast<uint32_t *>(s_scale + d.ib);          // [ib] FX: bits of the row's largest |entry|
    const uint32_t I = (uint32_t)d.I;

    if constexpr (FX) {   // the scale of every row: largest |entry| (an LDS integer max over the float bits), floored at 2^-6
        for (int e = tid; e < d.ib; e += nthreads) s_rowmax[e] = 0u;
        __syncthreads();
        for (int e = tid; e < ni * RS; e += nthreads) {
            const int row = e / RS, k = e - row * RS;
            if (k < K) {
                const float hv = d.p.H[(int64_t)(i0 + row) * K + k];
                if (!(fabsf(hv) < 1.0e9f)) atomicExch(err, 3);   // NaN / Inf: fail, never convert
                atomicMax(&s_rowmax[row], __float_as_uint(fabsf(hv)));
            }
        }
        __syncthreads();
        for (int e = tid; e < d.ib; e += nthreads) {
            int ex = 0;
            (void)frexpf(fmaxf(__uint_as_float(s_rowmax[e]), 0.015625f), &ex);   // max = f * 2^ex, f in [0.5, 1)
            s_scale[e] = make_float2(ldexpf(1.0f, 28 - ex), ldexpf(1.0f, ex - 28));
        }
        __syncthreads();
    }
    for (int e = tid; e < ni * RS; e += nthreads) {               // the tile's item rows (+ state) -> LDS
        const int row = e / RS, k = e - row * RS;
        const bool in = k < K;
        const int64_t g = (int64_t)(i0 + row) * K + k;
        if constexpr (FX) {
            shi[e] = in ? __float2int_rn(d.p.H[g] * s_scale[row].x) : 0;
            if constexpr (OPT == CYMF_OPT_ADAGRAD) sa64[e] = (long long)((double)(in ? d.p.H0[g] : SFILL) * 4294967296.0);
        } else {
            sh[e] = in ? d.p.H[g] : 0.0f;
            if constexpr (NS >= 1) ss0[e] = in ? d.p.H0[g] : SFILL;
            if constexpr (NS == 2) ss1[e] = in ? d.p.H1[g] : 0.0f;
        }
    }
    // q_ui = x_ui / max(p_i, M) of the tile's cells (cymf/model.pyx:117), staged once: a draw then costs one LDS read
    for (int e = tid; e < nu * d.ib; e += nthreads) {
        const int ul = e / d.ib, il = e - ul * d.ib;
        s_q[e] = il < ni ? d.p.X[((int64_t)u0 + ul) * d.I + i0 + il] / fmaxf(d.p.prop[i0 + il], d.p.clip) : 0.0f;
    }
    for (int e = tid; e < nu * 16; e += nthreads) s_cc[e] = 0u;
    __syncthreads();
    // draws per cell: item il of user ul is byte (il >> 4) of word [ul][il & 15] -- lane l of a worker reads ONE word
    // and holds the counts of items l, 16 + l, 32 + l, 48 + l (ib <= 64).  A byte cannot overflow in practice: a cell is
    // drawn Poisson(1) times per epoch.  If one did (old value 255), the add has already carried into the neighbouring
    // item's counter: it is taken back (adds and subtractions commute, nothing reads the words before the barrier), the
    // 256th draw of that cell is NOT applied, and the epoch is failed through err = 2 -- no silent loss of a draw.
    for (uint32_t t = t0 + tid; t < t1; t += nthreads) {
        const uint32_t c = sorted[t];
        const uint32_t u = c / I, i = c - u * I;
        const uint32_t ul = u - (uint32_t)u0, il = i - (uint32_t)i0;
        if (ul < (uint32_t)nu && il < (uint32_t)ni) {
            const uint32_t slot = il / LPD;              // item il: lane il % LPD of a worker, its slot-th counter
            const uint32_t old = atomicAdd(&s_cc[ul * 16 + (il % LPD) * WPL + (slot >> 2)], 1u << (8 * (slot & 3)));
            if (((old >> (8 * (slot & 3))) & 255u) == 255u) {   // counter overflow: never with uniform draws
                atomicSub(&s_cc[ul * 16 + (il % LPD) * WPL + (slot >> 2)], 1u << (8 * (slot & 3)));
                atomicExch(err, 2);
            }
        } else {
            atomicExch(err, 1);   // a cell outside its tile: broken bucketing must not become a wild LDS access
        }
    }
    __syncthreads();
    stamp(1);

    const int row = lane / LPD, l16 = lane % LPD;    // (l16: the lane's index inside its worker, 0 .. LPD-1)
    const int n_workers = n_waves * WPW, worker = wave * WPW + row;
    const int lane_off = CH * l16;
    float loss_u = 0.0f;
    unsigned int dbg_retries = 0, dbg_draws = 0;   // (developer counters, reported with CYMF_RELMF_TILE_STAMPS=1)
    using f2 = __attribute__((ext_vector_type(2))) float;
    f2 pl_acc2[EPL / 2];                                         // l2 term of all draws, one packed accumulator per register pair
#pragma unroll
    for (int pr = 0; pr < EPL / 2; ++pr) pl_acc2[pr] = f2{0.0f, 0.0f};
    struct UserRegs { float w[EPL], w0[EPL], w1[EPL]; int ul; uint32_t cw[WPL]; };
    auto fetch_user = [&](int ul, UserRegs &g) {
        const bool ok = ul < nu;
        g.ul = ok ? ul : 0;
#pragma unroll
        for (int q = 0; q < WPL; ++q) g.cw[q] = ok ? s_cc[ul * 16 + l16 * WPL + q] : 0u;
        const int64_t base = ((int64_t)u0 + g.ul) * K;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int k = 64 * (e / CH) + CH * l16 + (e % CH);
            const bool in = ok && k < K;
            g.w[e] = in ? d.p.W[base + k] : 0.0f;
            g.w0[e] = (NS >= 1 && in) ? d.p.W0[base + k] : SFILL;
            g.w1[e] = (NS == 2 && in) ? d.p.W1[base + k] : 0.0f;
        }
    };
    struct ItemRegs { float h[EPL], t0[EPL], t1[EPL], q; int il; float2 sc; };
    UserRegs nxt;
    fetch_user(worker, nxt);
    stamp(2);
    for (int round = 0; round * n_workers < nu; ++round) {
        UserRegs cu = nxt;
        fetch_user(worker + (round + 1) * n_workers, nxt);
        uint32_t any_cw = cu.cw[0];
#pragma unroll
        for (int q = 1; q < WPL; ++q) any_cw |= cu.cw[q];
        const bool had_draws = row_or<LPD>(any_cw) != 0u;
        // items of this user with count >= c, as a 64-bit mask held by every lane of the row.  The mask is walked from a
        // user-specific rotation: workers that all started at item 0 would meet on the same item rows at the same time
        // (measured with Adam, whose steps do not shrink with the gradient: norm of H +19 % against the sequential order).
        const int rot = (cu.ul * 37 + round * 11) & 63;
        uint32_t pass = 0;
        unsigned long long mask = 0ull;
        auto next_mask = [&]() {
            ++pass;
            // counter `slot` of lane l is item slot * LPD + l = bit slot * LPD + l of the mask
            uint32_t lo_bits = 0u, hi_bits = 0u;
#pragma unroll
            for (int slot = 0; slot < CH; ++slot) {
                const uint32_t bit = ((cu.cw[slot >> 2] >> (8 * (slot & 3))) & 255u) >= pass;
                if (slot * LPD < 32) lo_bits |= bit << (slot * LPD + l16);
                else hi_bits |= bit << (slot * LPD - 32 + l16);
            }
            const uint32_t lo = row_or<LPD>(lo_bits), hi = row_or<LPD>(hi_bits);
            const unsigned long long m = ((unsigned long long)hi << 32) | lo;    // bit il
            mask = rot ? (m >> rot) | (m << (64 - rot)) : m;                      // bit (il - rot) mod 64
        };
        auto pop_item = [&]() -> int {   // next set bit of the rotated mask, cleared; -1: this pass is exhausted
            const int bit = mask ? (__builtin_ctzll(mask) + rot) & 63 : -1;
            mask &= mask - 1ull;
            return bit;
        };
        auto next_item = [&]() -> int {
            int il = pop_item();
            if (il < 0) { next_mask(); il = pop_item(); }   // at most one empty pass: masks are nested, an empty one ends the user
            return il;
        };
        auto load_item = [&](int il, ItemRegs &g) {
            g.il = il;
            const int a = il < 0 ? 0 : il;
            g.q = s_q[cu.ul * d.ib + a];
            if constexpr (FX) g.sc = s_scale[a];
#pragma unroll
            for (int jv = 0; jv < R * WPL; ++jv) {               // jv: float4 number jv % WPL of block jv / WPL
                const int off = a * RS + 64 * (jv / WPL) + lane_off + 4 * (jv % WPL), e = 4 * jv;
                if constexpr (FX) {
                    const int4 v = *reinterpret_cast<const int4 *>(shi + off);
                    g.h[e] = (float)v.x * g.sc.y; g.h[e + 1] = (float)v.y * g.sc.y; g.h[e + 2] = (float)v.z * g.sc.y; g.h[e + 3] = (float)v.w * g.sc.y;
                    if constexpr (OPT == CYMF_OPT_ADAGRAD) {   // int64 with 32 fractional bits: (float) integer part + (float) fraction * 2^-32
                        const int4 a = *reinterpret_cast<const int4 *>(sa64 + off), b = *reinterpret_cast<const int4 *>(sa64 + off + 2);
                        g.t0[e] = fmaf((float)(unsigned int)a.x, FX_FRAC, (float)a.y);
                        g.t0[e + 1] = fmaf((float)(unsigned int)a.z, FX_FRAC, (float)a.w);
                        g.t0[e + 2] = fmaf((float)(unsigned int)b.x, FX_FRAC, (float)b.y);
                        g.t0[e + 3] = fmaf((float)(unsigned int)b.z, FX_FRAC, (float)b.w);
                    }
                } else {
                const float4 v = *reinterpret_cast<const float4 *>(sh + off);
                g.h[e] = v.x; g.h[e + 1] = v.y; g.h[e + 2] = v.z; g.h[e + 3] = v.w;
                if constexpr (NS >= 1) { const float4 s4 = *reinterpret_cast<const float4 *>(ss0 + off); g.t0[e] = s4.x; g.t0[e + 1] = s4.y; g.t0[e + 2] = s4.z; g.t0[e + 3] = s4.w; }
                }
                if constexpr (NS == 2) { const float4 s4 = *reinterpret_cast<const float4 *>(ss1 + off); g.t1[e] = s4.x; g.t1[e + 1] = s4.y; g.t1[e + 2] = s4.z; g.t1[e + 3] = s4.w; }
            }
        };
        // the previous draw's pending compare-and-swaps: pairs of row words (and of AdaGrad's accumulator words)
        bool pend = false;
        int p_il = 0;
        unsigned long long p_old[EPL / 2], p_seen[EPL / 2], a_old[EPL / 2], a_seen[EPL / 2];
        float p_dlt[EPL], a_dlt[EPL];
        auto settle = [&]() {
            // Adam: a swap that failed is NOT retried.  Its step does not shrink with the gradient, so two workers that formed
            // their steps from the same moments and both added them would move the row twice for one update of the moments
            // (measured on 1500 x 1400, K = 32, three epochs: norm of H +21 % against the sequential order with retries,
            // +8 % without; loss +0.6 % / +0.2 %).  The row keeps the other worker's step.
            if (pend && OPT != CYMF_OPT_ADAM) {
#pragma unroll
                for (int pr = 0; pr < EPL / 2; ++pr) {
                    const int e0 = 2 * pr, loc = p_il * RS + 64 * (e0 / CH) + lane_off + (e0 % CH);
                    unsigned long long *a = reinterpret_cast<unsigned long long *>(sh + loc);
                    unsigned long long old = p_old[pr], seen = p_seen[pr];
                    while (seen != old) {      // somebody else updated the words meanwhile: add the deltas to what is there now
                        ++dbg_retries;
                        old = seen;
                        seen = atomicCAS(a, old, pack2(__uint_as_float((unsigned int)old) + p_dlt[e0], __uint_as_float((unsigned int)(old >> 32)) + p_dlt[e0 + 1]));
                    }
                    if constexpr (OPT == CYMF_OPT_ADAGRAD) {
                        unsigned long long *b2 = reinterpret_cast<unsigned long long *>(ss0 + loc);
                        old = a_old[pr]; seen = a_seen[pr];
                        while (seen != old) {
                            old = seen;
                            seen = atomicCAS(b2, old, pack2(__uint_as_float((unsigned int)old) + a_dlt[e0], __uint_as_float((unsigned int)(old >> 32)) + a_dlt[e0 + 1]));
                        }
                    }
                }
            }
            pend = false;
        };
        // one draw: `cur` was loaded a step ago; `nx` is loaded now for the step after
        auto step = [&](ItemRegs &cur, ItemRegs &nx) {
            const bool act = cur.il >= 0;
            const int il_n = act ? next_item() : -1;
            load_item(il_n, nx);
            float py = 0.0f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) py += cu.w[e] * cur.h[e];
            float y = row_sum<LPD>(py);
            if (pend) asm volatile("" : "+v"(p_seen[0]), "+v"(y));   // keeps the wait for the previous draw's swaps behind this dot product
            settle();
            if (act) {
                const int il = cur.il;
#pragma unroll
                for (int pr = 0; pr < EPL / 2; ++pr) {           // l2 term of all draws, reduced once at the end: pairs as they lie in
                    const f2 wv2 = {cu.w[2 * pr], cu.w[2 * pr + 1]}, hv2 = {cur.h[2 * pr], cur.h[2 * pr + 1]};   // the registers, or the packing moves cost more than the FMAs
                    pl_acc2[pr] = __builtin_elementwise_fma(wv2, wv2, pl_acc2[pr]);
                    pl_acc2[pr] = __builtin_elementwise_fma(hv2, hv2, pl_acc2[pr]);
                }
                // q (1 - y)^2 + (1 - q) y^2 = q + y (y - 2 q)   (cymf/model.pyx:117; the LPD lanes of the worker hold the same draw:
                // the sum is divided by LPD at the end);   q (1 - y) + (1 - q)(0 - y) = q - y   (cymf/model.pyx:131-139, no factor 2)
                const float q = cur.q;
                loss_u += fmaf(y, fmaf(-2.0f, q, y), q);
                const float c = q - y;
                float hnew[EPL], s0n[EPL], s1n[EPL];
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const float wv = cu.w[e], hv = cur.h[e];
                    const float gw = -(c * hv) + d.p.wd * wv;
                    const float gh = -(c * wv) + d.p.wd * hv;
                    opt_update<float, OPT, true>(d.p.opt, cu.w[e], cu.w0[e], cu.w1[e], gw);
                    hnew[e] = hv; s0n[e] = NS >= 1 ? cur.t0[e] : 0.0f; s1n[e] = NS == 2 ? cur.t1[e] : 0.0f;
                    opt_update<float, OPT, true>(d.p.opt, hnew[e], s0n[e], s1n[e], gh);
                }
                if constexpr (FX) {   // integer atomic adds of the row's (and AdaGrad's accumulator's) increments: no return, no retry
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const int loc = il * RS + 64 * (e / CH) + lane_off + (e % CH);
                        atomicAdd(shi + loc, __float2int_rn((hnew[e] - cur.h[e]) * cur.sc.x));
                        if constexpr (OPT == CYMF_OPT_ADAGRAD)
                            atomicAdd(reinterpret_cast<unsigned long long *>(sa64 + loc), (unsigned long long)__float2ll_rn((s0n[e] - cur.t0[e]) * FX_S));
                    }
                    ++dbg_draws;
                } else {
                // Adam's moments are not additive and are stored as this worker's consistent pair (as in the other lock-free
                // kernels); its row takes the step through a compare-and-swap that is not retried (see settle)
#pragma unroll
                for (int pr = 0; pr < EPL / 2; ++pr) {
                    const int e0 = 2 * pr, loc = il * RS + 64 * (e0 / CH) + lane_off + (e0 % CH);
                    p_old[pr] = pack2(cur.h[e0], cur.h[e0 + 1]);
                    p_dlt[e0] = hnew[e0] - cur.h[e0]; p_dlt[e0 + 1] = hnew[e0 + 1] - cur.h[e0 + 1];
                    p_seen[pr] = atomicCAS(reinterpret_cast<unsigned long long *>(sh + loc), p_old[pr], pack2(hnew[e0], hnew[e0 + 1]));
                    if constexpr (OPT == CYMF_OPT_ADAGRAD) {
                        a_old[pr] = pack2(cur.t0[e0], cur.t0[e0 + 1]);
                        a_dlt[e0] = s0n[e0] - cur.t0[e0]; a_dlt[e0 + 1] = s0n[e0 + 1] - cur.t0[e0 + 1];
                        a_seen[pr] = atomicCAS(reinterpret_cast<unsigned long long *>(ss0 + loc), a_old[pr], pack2(s0n[e0], s0n[e0 + 1]));
                    }
                    if constexpr (OPT == CYMF_OPT_ADAM) {
                        *reinterpret_cast<float2 *>(ss0 + loc) = make_float2(s0n[e0], s0n[e0 + 1]);
                        *reinterpret_cast<float2 *>(ss1 + loc) = make_float2(s1n[e0], s1n[e0 + 1]);
                    }
                }
                p_il = il;
                pend = true;
                ++dbg_draws;
                }
                // the same item twice in a row (last item of one pass, first of the next): the early read of its row
                // predates the update just issued
                if (il_n == il) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        nx.h[e] = hnew[e];
                        if constexpr (NS >= 1) nx.t0[e] = s0n[e];
                        if constexpr (NS == 2) nx.t1[e] = s1n[e];
                    }
                }
            }
        };
        ItemRegs ia, ib2;
        load_item(next_item(), ia);
        while (__any(ia.il >= 0)) {
            step(ia, ib2);
            if (!__any(ib2.il >= 0)) break;
            step(ib2, ia);
        }
        settle();
        if (had_draws) {
            const int64_t base = ((int64_t)u0 + cu.ul) * K;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int k = 64 * (e / CH) + CH * l16 + (e % CH);
                if (k < K) {
                    d.p.W[base + k] = cu.w[e];
                    if constexpr (NS >= 1) d.p.W0[base + k] = cu.w0[e];
                    if constexpr (NS == 2) d.p.W1[base + k] = cu.w1[e];
                }
            }
        }
    }
    stamp(3);
    __syncthreads();
    stamp(4);
    for (int e = tid; e < ni * RS; e += nthreads) {               // the item rows go back to HBM once per tile
        const int row2 = e / RS, k = e - row2 * RS;
        if (k < K) {
            const int64_t g = (int64_t)(i0 + row2) * K + k;
            if constexpr (FX) {
                const int hv_i = shi[e];
                if (hv_i > (1 << 30) || hv_i < -(1 << 30)) atomicExch(err, 3);   // grew more than fourfold inside one tile visit: too close to a wrap
                d.p.H[g] = (float)hv_i * s_scale[row2].y;
                if constexpr (OPT == CYMF_OPT_ADAGRAD) d.p.H0[g] = (float)((double)sa64[e] * (1.0 / 4294967296.0));
            } else {
                d.p.H[g] = sh[e];
                if constexpr (NS >= 1) d.p.H0[g] = ss0[e];
                if constexpr (NS == 2) d.p.H1[g] = ss1[e];
            }
        }
    }
    float pl_acc = 0.0f;
#pragma unroll
    for (int pr = 0; pr < EPL / 2; ++pr) pl_acc += pl_acc2[pr][0] + pl_acc2[pr][1];
    const float l2 = wave_sum(pl_acc), lu = wave_sum(loss_u) * (1.0f / (float)LPD);
    if (lane == 0) atomicAdd(loss_acc, (double)(lu + d.p.wd * l2));
    stamp(5);
    if (stamps) {   // compare-and-swap retries per lane and pair of words, and draws per lane, summed over the launch
        atomicAdd(reinterpret_cast<unsigned long long *>(stamps) + 6, (unsigned long long)dbg_retries);
        atomicAdd(reinterpret_cast<unsigned long long *>(stamps) + 7, (unsigned long long)dbg_draws);
    }
}

size_t tile_lds_bytes(const RelTilePlan &p) {
    const int NS = p.opt == CYMF_OPT_SGD ? 0 : (p.opt == CYMF_OPT_ADAGRAD ? 1 : 2);
    const int words = 1 + (p.opt == CYMF_OPT_ADAGRAD ? 2 : NS);   // AdaGrad's accumulator rows are int64 fixed point (relmf_tile_kernel: FX)
    return (size_t)words * p.ib * p.R * 64 * sizeof(float) + (size_t)p.ub * p.ib * sizeof(float) + (size_t)p.ub * 16 * sizeof(uint32_t) +
           (size_t)p.ib * (sizeof(float2) + sizeof(uint32_t));   // + per-row scales and maxima (FX)
}

template <typename F>
int allow_lds(F kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        CYMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

template <int R>
int launch_tile(const RelTilePlan &p, const RelTileDev &d, const uint32_t *sorted, const uint32_t *toff, int shift, double *loss,
                int *err, long long *stamps, hipStream_t s) {
#define TILE_L_(O_, MAXT_, LPD_)                                                                                        \
    do {                                                                                                                \
        CYMF_TRY(allow_lds(relmf_tile_kernel<R, O_, MAXT_, LPD_>, p.lds_bytes));                                        \
        hipLaunchKernelGGL((relmf_tile_kernel<R, O_, MAXT_, LPD_>), dim3(p.B), dim3(p.threads), p.lds_bytes, s, d, sorted, toff, shift, loss, err, stamps); \
    } while (0)
#define TILE_(O_)                                                                                                       \
    do {                                                                                                                \
        if constexpr (R == 1) {                                                                                         \
            if (p.lpd == 8) {   /* eight lanes per draw, eight elements per lane */                                     \
                if (p.threads > 768) TILE_L_(O_, 1024, 8);                                                              \
                else if (p.threads > 512) TILE_L_(O_, 768, 8);                                                          \
                else TILE_L_(O_, 512, 8);                                                                               \
                break;                                                                                                  \
            }                                                                                                           \
        }                                                                                                               \
        if (p.threads > 512) TILE_L_(O_, 1024, 16);   /* up to 16 wavefronts: the 128-VGPR build */                     \
        else TILE_L_(O_, 512, 16);                                                                                      \
    } while (0)
    if (p.opt == CYMF_OPT_SGD) TILE_(CYMF_OPT_SGD);
    else if (p.opt == CYMF_OPT_ADAGRAD) TILE_(CYMF_OPT_ADAGRAD);
    else TILE_(CYMF_OPT_ADAM);
#undef TILE_
#undef TILE_L_
    return 0;
}

}  // namespace

bool relmf_tile_plan(int32_t U, int32_t I, int32_t K, int opt, RelTilePlan *plan) {
    if (K > 256 || U < 1 || I < 1) return false;
    RelTilePlan p;
    p.U = U; p.I = I; p.K = K; p.opt = opt;
    p.N = (int64_t)U * I;
    if (p.N >= ((int64_t)1 << 32)) return false;
    p.R = (K + 63) / 64;
    const int NS = opt == CYMF_OPT_SGD ? 0 : (opt == CYMF_OPT_ADAGRAD ? 1 : 2);
    // item rows (+ state) of a tile: at most 48 KB of LDS and one lane per item
    const int ib_max = std::max(1, std::min(64, (48 * 1024) / (p.R * 256 * (1 + NS))));
    // about one tile per CU, but blocks of at least ~8 users and ~32 items: the workers of a tile share its item rows
    int64_t B = std::min<int64_t>(256, std::max<int64_t>(1, std::min<int64_t>(U / 8, I / 32)));
    if (const char *eb = getenv("CYMF_RELMF_TILE_B")) B = std::max(1, atoi(eb));              // developer: tile count per side
    B = std::max<int64_t>(B, (I + ib_max - 1) / ib_max);
    B = std::max<int64_t>(B, ((int64_t)U + 4095) / 4096);    // the tile kernel sorts up to 4096 users per block
    while (B < 4096 && ((U + B - 1) / B) * ((I + B - 1) / B) > 12288) ++B;   // q of a tile's cells: at most 48 KB of LDS
    if (B > 4096) return false;
    p.B = (int32_t)B;
    p.ub = (int32_t)((U + B - 1) / B);
    p.ib = (int32_t)((I + B - 1) / B);
    if (p.ib > ib_max) return false;
    // wavefronts per workgroup = 4 row workers each (at most 8: the kernel is compiled for 512 threads, i.e. up to 256
    // VGPRs -- a worker holds its user's row and state, two draws' item rows and the pending swaps); the users of a block
    // are dealt round-robin, so pick the count that leaves the fewest workers idle in the last round, preferring two
    // wavefronts per SIMD
    // Lanes per draw: 16 with four row elements each.  Eight lanes with eight elements (K <= 64; CYMF_RELMF_TILE_LPD=8) share a
    // step's bookkeeping (next item, pending swaps, masks, branches: ~3/4 of its instructions) between eight draws instead
    // of four and were expected to be faster; measured on 20000 x 8000, K = 64: 30.8 against 25.6 ms per epoch (SGD), 41 against
    // 34 (AdaGrad) -- the 80 users of a block then need ten wavefronts instead of sixteen, and a step is a latency chain
    // (LDS read -> dot -> DPP sum -> swap) that the fewer wavefronts hide less well.  Kept as a tested variant.
    p.lpd = 16;
    if (const char *el = getenv("CYMF_RELMF_TILE_LPD")) p.lpd = (atoi(el) == 8 && p.R == 1) ? 8 : 16;   // developer
    const int wpw = 64 / p.lpd;
    int best_nw = 4;
    double best = 1e30;
    // 16: the 1024-thread build holds K <= 64 without spilling; eight lanes per draw with AdaGrad / Adam state: 12 (the 768-thread build)
    const int nw_max = p.R == 1 ? ((p.lpd == 8 && opt != CYMF_OPT_SGD) ? 12 : 16) : 8;
    for (int nw = 4; nw <= nw_max; nw += 2) {   // (even counts: an odd one leaves a SIMD with a wavefront less -- 7 measured 21.1 ms where 8 takes 19.7)
        // time ~ rounds x cost of a round.  A round's cost rises slowly up to two wavefronts per SIMD and jumps beyond: with the
        // integer-atomic write-back (FX) two wavefronts already keep a SIMD's VALU busy.  Measured on 20000 x 8000, K = 64, SGD
        // (80 users per block), ms per epoch / rounds: 6 wavefronts 21.8 / 4, 8: 19.7 / 3, 10: 21.7 / 2, 12: 21.1 / 2, 16: 23.6 / 2
        // (Adam: 8: 33.9, 10: 39.4, 12: 35.1, 16: 38.4).
        const int workers = wpw * nw, rounds = (p.ub + workers - 1) / workers;
        const double cost = nw <= 8 ? 3.0 + 0.45 * nw : 10.0 + 0.2 * (nw - 10);
        if (rounds * cost < best - 1e-9) { best = rounds * cost; best_nw = nw; }
    }
    p.threads = 64 * best_nw;
    if (const char *ew = getenv("CYMF_RELMF_TILE_WAVES")) p.threads = 64 * std::min(16, std::max(1, atoi(ew)));   // developer
    p.lds_bytes = tile_lds_bytes(p);
    if (p.lds_bytes > 150 * 1024) return false;
    *plan = p;
    return true;
}

int relmf_tile_bucket(const RelTilePlan &p, const uint32_t *cells, RelTileBufs &bufs, int parity, hipStream_t s) {
    const int B = p.B;
    const uint32_t N = (uint32_t)p.N;
    CYMF_TRY(bufs.pass1.alloc((size_t)p.N));
    CYMF_TRY(bufs.sorted[parity].alloc((size_t)p.N));
    CYMF_TRY(bufs.cnt1.alloc((size_t)B));
    CYMF_TRY(bufs.off1.alloc((size_t)B + 1));
    CYMF_TRY(bufs.cur1.alloc((size_t)B));
    CYMF_TRY(bufs.cnt2.alloc((size_t)B * B));
    CYMF_TRY(bufs.cur2.alloc((size_t)B * B));
    CYMF_TRY(bufs.toff[parity].alloc((size_t)B * B + 1));
    CYMF_TRY(bufs.cnt1.zero(s));
    CYMF_TRY(bufs.cnt2.zero(s));
    const size_t lds_count = (size_t)B * sizeof(uint32_t);
    const size_t lds_scatter = ((size_t)3 * B + 32 + BK_SEG) * sizeof(uint32_t);
    CYMF_TRY(allow_lds(tile_count_kernel<1>, lds_count));
    CYMF_TRY(allow_lds(tile_count_kernel<2>, lds_count));
    CYMF_TRY(allow_lds(tile_scatter_kernel<1>, lds_scatter));
    CYMF_TRY(allow_lds(tile_scatter_kernel<2>, lds_scatter));
    const int segs1 = (int)std::max<int64_t>(1, std::min<int64_t>((p.N + BK_SEG - 1) / BK_SEG, 256 * 64));
    // pass 1: by user block
    hipLaunchKernelGGL(tile_count_kernel<1>, dim3(segs1), dim3(BK_THREADS), lds_count, s, cells, N, nullptr, (uint32_t)p.I,
                       (uint32_t)p.ub, (uint32_t)p.ib, B, bufs.cnt1.p);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(64), 0, s, bufs.cnt1.p, B, nullptr, bufs.off1.p, bufs.cur1.p);
    hipLaunchKernelGGL(tile_scatter_kernel<1>, dim3(segs1), dim3(BK_THREADS), lds_scatter, s, cells, N, nullptr, (uint32_t)p.I,
                       (uint32_t)p.ub, (uint32_t)p.ib, B, bufs.cur1.p, bufs.pass1.p);
    // pass 2: inside every user block, by item block (the buckets are uniform draws: N / B each, give or take)
    const int segs2 = (int)std::max<int64_t>(1, std::min<int64_t>((p.N / B + BK_SEG - 1) / BK_SEG + 1, 4096));
    hipLaunchKernelGGL(tile_count_kernel<2>, dim3(segs2, B), dim3(BK_THREADS), lds_count, s, bufs.pass1.p, N, bufs.off1.p,
                       (uint32_t)p.I, (uint32_t)p.ub, (uint32_t)p.ib, B, bufs.cnt2.p);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(B), dim3(64), 0, s, bufs.cnt2.p, B, bufs.off1.p, bufs.toff[parity].p, bufs.cur2.p);
    hipLaunchKernelGGL(tile_scatter_kernel<2>, dim3(segs2, B), dim3(BK_THREADS), lds_scatter, s, bufs.pass1.p, N, bufs.off1.p,
                       (uint32_t)p.I, (uint32_t)p.ub, (uint32_t)p.ib, B, bufs.cur2.p, bufs.sorted[parity].p);
    CYMF_HIP(hipGetLastError());
    return 0;
}

int relmf_tile_epoch(const RelTilePlan &p, const RelTileParams &prm, const RelTileBufs &bufs, int parity, int64_t epoch,
                     double *loss_acc, int *err, hipStream_t s) {
    RelTileDev d;
    d.p = prm;
    d.U = p.U; d.I = p.I; d.K = p.K; d.B = p.B; d.ub = p.ub; d.ib = p.ib;
    // sub-step st pairs user block b with item block (b + st + rotation) mod B; the rotation changes with the epoch so
    // that a user block does not meet the item blocks in the same order every time
    const int rot = (int)((epoch * 17) % p.B);
    static const bool want_stamps = getenv("CYMF_RELMF_TILE_STAMPS") && getenv("CYMF_RELMF_TILE_STAMPS")[0] == '1';
    static DevBuf<long long> stamp_buf;
    long long *stamps = nullptr;
    if (want_stamps) {
        CYMF_TRY(stamp_buf.alloc(16));
        CYMF_TRY(stamp_buf.zero(s));
        stamps = stamp_buf.p;
    }
    for (int st = 0; st < p.B; ++st) {
        const int shift = (st + rot) % p.B;
        switch (p.R) {
        case 1: CYMF_TRY(launch_tile<1>(p, d, bufs.sorted[parity].p, bufs.toff[parity].p, shift, loss_acc, err, stamps, s)); break;
        case 2: CYMF_TRY(launch_tile<2>(p, d, bufs.sorted[parity].p, bufs.toff[parity].p, shift, loss_acc, err, stamps, s)); break;
        case 3: CYMF_TRY(launch_tile<3>(p, d, bufs.sorted[parity].p, bufs.toff[parity].p, shift, loss_acc, err, stamps, s)); break;
        default: CYMF_TRY(launch_tile<4>(p, d, bufs.sorted[parity].p, bufs.toff[parity].p, shift, loss_acc, err, stamps, s)); break;
        }
    }
    CYMF_HIP(hipGetLastError());
    if (want_stamps) {
        long long h[16];
        CYMF_HIP(hipStreamSynchronize(s));
        CYMF_HIP(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
        for (int w = 0; w < 2; ++w)
            fprintf(stderr, "[relmf tile stamps, last sub-step, workgroup 0, %s wave] load+sort %lld, first fetch %lld, users %lld, barrier %lld, write-back %lld cycles\n",
                    w ? "last" : "first", h[8 * w + 1] - h[8 * w], h[8 * w + 2] - h[8 * w + 1], h[8 * w + 3] - h[8 * w + 2], h[8 * w + 4] - h[8 * w + 3],
                    h[8 * w + 5] - h[8 * w + 4]);
        fprintf(stderr, "[relmf tile stamps, whole epoch] compare-and-swap retries per draw and word pair: %.4f (%lld retries, %lld lane-draws)\n",
                h[7] ? (double)h[6] / (double)h[7] / 2.0 : 0.0, h[6], h[7]);
    }
    return 0;
}

}  // namespace cymf
