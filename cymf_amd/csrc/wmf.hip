// wmf.hip -- WMF (Hu-Koren-Volinsky) ALS half-sweep on gfx950.
// Replaces WMF._als (cymf/wmf.pyx:136-174) and solvep = LAPACK dgesv (cymf/linalg.pyx:144-163):
//     x_i = (YtY + lambda I + (w-1) sum_{j in row i} y_j y_j^T)^-1 (w sum_{j in row i} y_j)
// Rows are independent (the reference's prange, wmf.pyx:150): one workgroup per row, the K x K
// normal matrix lives in LDS from its construction to the end of the solve and never touches HBM.
// The system is SPD (lambda > 0), so the LU-with-pivoting of dgesv is replaced by an in-LDS
// Cholesky; both solve the same system (SURVEY.md 7 hard-6: 5e-15 apart in fp64).
//
//   gram kernel  : YtY (K x K) by slabs of rows, per-workgroup partial sums, float/double atomics.
//   row kernel   : generic path (f32 and f64): LDS tile of gathered y rows, each thread owns
//                  entries of A.
//   row kernel f32 MFMA path (K % 32 == 0): the gathered Gramian sum_j y_j y_j^T is built with
//                  v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate) straight from the
//                  gathered rows; upper-triangular tiles only.
#include <algorithm>

#include "store.h"

namespace cymf {
struct WmfSeg { int32_t slot, begin, end, pad; };   // segment of a long row (see wmf_row_mfma_kernel)
namespace {

constexpr int WMF_THREADS = 256;
constexpr int WMF_TILE = 16;   // gathered rows staged per pass (generic path)

template <typename T>
__global__ __launch_bounds__(WMF_THREADS) void wmf_gram_kernel(const T *__restrict__ Y, int64_t cols, int K,
                                                              T *__restrict__ G) {
    // each workgroup reduces a slab of rows of Y into a K x K partial (registers), then atomics
    extern __shared__ unsigned char smem_raw[];
    T *tile = reinterpret_cast<T *>(smem_raw);   // [WMF_TILE][K]
    const int tid = threadIdx.x;
    const int KK = K * K;
    constexpr int MAXE = (128 * 128 + WMF_THREADS - 1) / WMF_THREADS;   // K <= 128
    T acc[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) acc[e] = 0;
    for (int64_t base = (int64_t)blockIdx.x * WMF_TILE; base < cols; base += (int64_t)gridDim.x * WMF_TILE) {
        const int nr = (int)(cols - base < WMF_TILE ? cols - base : WMF_TILE);
        __syncthreads();
        for (int e = tid; e < nr * K; e += WMF_THREADS) tile[e] = Y[base * K + e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            const int idx = tid + e * WMF_THREADS;
            if (idx < KK) {
                const int k = idx / K, k2 = idx - k * K;
                T s = 0;
                for (int r = 0; r < nr; ++r) s += tile[r * K + k] * tile[r * K + k2];
                acc[e] += s;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int idx = tid + e * WMF_THREADS;
        if (idx < KK && acc[e] != (T)0) atomicAdd(G + idx, acc[e]);
    }
}

// YtY for any K (cymf/wmf.pyx:142 via np.dot): a workgroup forms one 16 x 16 tile of G over a slab of rows of Y,
// both 16-column panels staged through LDS, and adds it to G with atomics.  Used when K > 128.
template <typename T>
__global__ __launch_bounds__(256) void wmf_gram_wide_kernel(const T *__restrict__ Y, int64_t cols, int K, int n_kb,
                                                           T *__restrict__ G) {
    __shared__ T pa[16][17], pb[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int ka = (blockIdx.x / n_kb) * 16, kb = (blockIdx.x % n_kb) * 16;
    T acc = 0;
    for (int64_t base = (int64_t)blockIdx.y * 16; base < cols; base += (int64_t)gridDim.y * 16) {
        __syncthreads();
        const int64_t r = base + ty;
        pa[ty][tx] = (r < cols && ka + tx < K) ? Y[r * K + ka + tx] : (T)0;
        pb[ty][tx] = (r < cols && kb + tx < K) ? Y[r * K + kb + tx] : (T)0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += pa[q][ty] * pb[q][tx];
    }
    if (ka + ty < K && kb + tx < K && acc != (T)0) atomicAdd(G + (int64_t)(ka + ty) * K + kb + tx, acc);
}

template <typename T>
__global__ void wmf_add_diag_kernel(T *G, int K, T lambda) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) G[k * K + k] += lambda;
}

__device__ __forceinline__ float dsqrt(float x) { return __fsqrt_rn(x); }
__device__ __forceinline__ double dsqrt(double x) { return sqrt(x); }

// In-LDS Cholesky A = L L^T (lower, in place) and solve of A x = b; A is [K][lda], lda = K+1
// (odd stride: a column walk touches every LDS bank once).  Two barriers per column in the
// factorization (scale column c; rank-1 update of the trailing lower triangle on a 16x16 thread
// tile, no integer division); the two triangular solves run on wavefront 0 alone with b in LDS, so
// they need no workgroup barrier at all (LDS operations of one wavefront complete in program order).
template <typename T>
__device__ void chol_solve_lds(T *A, T *b, T *dg, int K, int lda) {
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    for (int c = 0; c < K; ++c) {
        __syncthreads();                         // trailing update of column c-1 is complete
        const T dcc = A[c * lda + c];
        const T inv = (T)1 / dsqrt(dcc);
        if (tid == 0) dg[c] = dsqrt(dcc);        // the pivot itself stays in place: nobody waits to overwrite it
        for (int r = c + 1 + tid; r < K; r += WMF_THREADS) A[r * lda + c] *= inv;
        __syncthreads();
        // A[r][q] -= L[r][c] L[q][c] for c < q <= r < K
        for (int r = c + 1 + ty; r < K; r += 16) {
            const T lrc = A[r * lda + c];
            for (int q = c + 1 + tx; q <= r; q += 16) A[r * lda + q] -= lrc * A[q * lda + c];
        }
    }
    __syncthreads();
    if (tid < 64) {
        // lanes of ONE wavefront exchange b through LDS without a barrier: volatile keeps the compiler from
        // caching or reordering these accesses (the hardware executes a wave's LDS operations in order)
        volatile T *vb = b;
        // L z = b
        for (int c = 0; c < K; ++c) {
            const T bc = vb[c] / dg[c];
            for (int r = c + 1 + tid; r < K; r += 64) vb[r] = vb[r] - A[r * lda + c] * bc;
            if (tid == 0) vb[c] = bc;
        }
        // L^T x = z
        for (int c = K - 1; c >= 0; --c) {
            const T bc = vb[c] / dg[c];
            for (int r = tid; r < c; r += 64) vb[r] = vb[r] - A[c * lda + r] * bc;
            if (tid == 0) vb[c] = bc;
        }
    }
    __syncthreads();
}

// Generic row kernel (any K, f32 or f64).  The K x K system lives in LDS while it fits (A_scratch == nullptr); for
// larger K (cymf/wmf.pyx:44 takes any num_components) each workgroup works in its own K x (K+1) slice of a global
// scratch buffer instead -- same code, the workgroup barriers order its global accesses as they order the LDS ones.
template <typename T>
__global__ __launch_bounds__(WMF_THREADS) void wmf_row_kernel(int32_t rows, int K, const int32_t *__restrict__ indptr,
                                                             const int32_t *__restrict__ indices,
                                                             T *__restrict__ X, const T *__restrict__ Y,
                                                             const T *__restrict__ A0, T weight, T *A_scratch) {
    extern __shared__ unsigned char smem_raw[];
    const int lda = K + 1;
    T *lds = reinterpret_cast<T *>(smem_raw);
    T *A = A_scratch ? A_scratch + (size_t)blockIdx.x * K * lda : lds;   // [K][K+1]
    T *b = A_scratch ? lds : lds + K * lda;    // [K]
    T *dg = b + K;                             // [K] Cholesky diagonal
    T *tile = dg + K;                          // [WMF_TILE][K]
    const int tid = threadIdx.x;
    const int KK = K * K;
    for (int32_t i = blockIdx.x; i < rows; i += gridDim.x) {
        const int32_t p0 = indptr[i], p1 = indptr[i + 1];
        if (p0 == p1) {                                         // wmf.pyx:154-156
            for (int k = tid; k < K; k += WMF_THREADS) X[(int64_t)i * K + k] = 0;
            continue;
        }
        __syncthreads();
        for (int e = tid; e < KK; e += WMF_THREADS) A[(e / K) * lda + (e % K)] = A0[e];   // wmf.pyx:158
        for (int k = tid; k < K; k += WMF_THREADS) b[k] = 0;
        for (int32_t p = p0; p < p1; p += WMF_TILE) {          // wmf.pyx:161-166
            const int nr = p1 - p < WMF_TILE ? p1 - p : WMF_TILE;
            __syncthreads();
            for (int e = tid; e < nr * K; e += WMF_THREADS) {
                const int r = e / K, k = e - r * K;
                tile[e] = Y[(int64_t)indices[p + r] * K + k];
            }
            __syncthreads();
            for (int e = tid; e < KK; e += WMF_THREADS) {
                const int k = e / K, k2 = e - k * K;
                T s = 0;
                for (int r = 0; r < nr; ++r) s += tile[r * K + k] * tile[r * K + k2];
                A[k * lda + k2] += s * (weight - (T)1);
            }
            for (int k = tid; k < K; k += WMF_THREADS) {
                T s = 0;
                for (int r = 0; r < nr; ++r) s += tile[r * K + k];
                b[k] += s * weight;
            }
        }
        __syncthreads();
        chol_solve_lds<T>(A, b, dg, K, lda);                    // wmf.pyx:168
        for (int k = tid; k < K; k += WMF_THREADS) X[(int64_t)i * K + k] = b[k];   // wmf.pyx:170-171
    }
}

// f32 MFMA row kernel, K = 32*T32.  Wave w owns upper-triangular 32x32 tiles t = w, w+4, ...
// of G = sum_j y_j y_j^T.  v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; here k indexes the gathered row (two per instruction), so both operands
// are read directly from the gathered rows: lane l loads y[row_{2s + (l>>5)}][32*tile + (l&31)].
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Work items: item s < n_segs is segment s of a LONG row (see below); item n_segs + r is row r in full
// (rows longer than long_threshold are skipped there).  A segment is seg = {row slot, begin, end}: its partial Gramian and
// partial sum are ADDED to scratch[slot] (K*K + K floats per long row, zeroed by the host) and the
// solve is left to wmf_long_finish_kernel.  A row of 10^5 entries is thus built by ~50 workgroups
// instead of being one workgroup's serial tail.
template <int T32, bool SEG>
__global__ __launch_bounds__(WMF_THREADS) void wmf_row_mfma_kernel(int32_t rows, const int32_t *__restrict__ indptr,
                                                                  const int32_t *__restrict__ indices,
                                                                  float *__restrict__ X, const float *__restrict__ Y,
                                                                  const float *__restrict__ A0, float weight,
                                                                  int32_t long_threshold, const WmfSeg *__restrict__ segs,
                                                                  int32_t n_segs, float *__restrict__ scratch) {
    constexpr int K = 32 * T32;
    constexpr int NT = T32 * (T32 + 1) / 2;          // upper-triangular tiles
    constexpr int TPW = (NT + 3) / 4;                // tiles per wave
    constexpr int lda = K + 1;
    extern __shared__ unsigned char smem_raw[];
    float *A = reinterpret_cast<float *>(smem_raw);  // [K][K+1]
    float *b = A + K * lda;                           // [K]
    float *dg = b + K;                                // [K] Cholesky diagonal
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    const int32_t n_items = SEG ? n_segs : rows;    // two instantiations: whole rows / segments of long rows
    for (int32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        constexpr bool whole = !SEG;
        const int32_t i = whole ? item : 0;
        int32_t p0, p1, slot = -1;
        if (whole) {
            p0 = indptr[i]; p1 = indptr[i + 1];
            if (p0 == p1) {
                for (int k = tid; k < K; k += WMF_THREADS) X[(int64_t)i * K + k] = 0;
                continue;
            }
            if (long_threshold > 0 && p1 - p0 > long_threshold) continue;   // built from segments
        } else {
            const WmfSeg sg = segs[item];
            slot = sg.slot; p0 = sg.begin; p1 = sg.end;
        }
        __syncthreads();
        f32x16 acc[TPW];
        float bsum[TPW];
        int tm[TPW], tn[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            acc[q] = (f32x16)(0.0f);
            bsum[q] = 0.0f;
            int m = 0, rem = wave + 4 * q;   // t-th pair (m <= n) in row-major upper-triangular order
            while (m < T32 && rem >= T32 - m) { rem -= T32 - m; ++m; }
            tm[q] = m;
            tn[q] = m + rem;
        }
        // 64 gathered rows per batch: lane l fetches the index of row l, the MFMA steps read the
        // two rows of a step through ds_bpermute; 8 steps of loads are in flight before their MFMAs
        for (int32_t pb = p0; pb < p1; pb += 64) {
            const int32_t myp = pb + lane;
            const int32_t myidx = myp < p1 ? indices[myp] : -1;
            const int nb = p1 - pb < 64 ? p1 - pb : 64;
            const int steps = (nb + 1) >> 1;
            for (int s0 = 0; s0 < steps; s0 += 8) {
                float av[8][TPW], bv[8][TPW];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int32_t idx = __shfl(myidx, 2 * (s0 + u) + lh, 64);
                    const bool ok = idx >= 0;
                    const float *yrow = Y + (int64_t)(ok ? idx : 0) * K;
#pragma unroll
                    for (int q = 0; q < TPW; ++q) {
                        av[u][q] = 0.0f;
                        bv[u][q] = 0.0f;
                        if (wave + 4 * q < NT) {   // wave-uniform
                            av[u][q] = ok ? yrow[32 * tm[q] + li] : 0.0f;
                            bv[u][q] = ok ? yrow[32 * tn[q] + li] : 0.0f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int q = 0; q < TPW; ++q) {
                        if (wave + 4 * q < NT) {
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][q], bv[u][q], acc[q], 0, 0, 0);
                            if (tm[q] == tn[q]) bsum[q] += av[u][q];
                        }
                    }
                }
            }
        }
        if constexpr (!whole) {   // segment of a long row: add the partial sums, no solve here
            float *G = scratch + (size_t)slot * (K * K + K);
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                if (wave + 4 * q < NT) {
                    if (tm[q] == tn[q]) {
                        const float tot = bsum[q] + __shfl_xor(bsum[q], 32, 64);
                        if (lh == 0) atomicAdd(G + K * K + 32 * tm[q] + li, tot);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * tm[q] + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int col = 32 * tn[q] + li;
                        atomicAdd(G + row * K + col, acc[q][r]);
                        if (tm[q] != tn[q]) atomicAdd(G + col * K + row, acc[q][r]);
                    }
                }
            }
            continue;
        }
        // b = w * sum_j y_j, from the diagonal tiles' A operands (two half-waves = two rows per step)
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            if (wave + 4 * q < NT && tm[q] == tn[q]) {
                const float tot = bsum[q] + __shfl_xor(bsum[q], 32, 64);
                if (lh == 0) b[32 * tm[q] + li] = tot * weight;
            }
        }
        // A = A0 + (w-1) G ; C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            if (wave + 4 * q < NT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * tm[q] + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int col = 32 * tn[q] + li;
                    const float g = acc[q][r] * (weight - 1.0f);
                    A[row * lda + col] = A0[row * K + col] + g;
                    if (tm[q] != tn[q]) A[col * lda + row] = A0[col * K + row] + g;   // mirror
                }
            }
        }
        __syncthreads();
        if (weight > -1e30f) chol_solve_lds<float>(A, b, dg, K, lda);   // (a huge negative weight is the no-solve timing probe)
        for (int k = tid; k < K; k += WMF_THREADS) X[(int64_t)i * K + k] = b[k];
    }
}

// ---------------------------------------------------------------------------------------------
// Register-resident solve.  The in-LDS Cholesky above spends its time in barriers and LDS latency
// (two workgroup barriers and a read-modify-write of the trailing triangle per column): measured
// 24 us of CU time per K=64 row on C4, 5-10x the Gramian that precedes it.  Here one lane owns one ROW of the
// (full, symmetric) system, in registers: K <= 64 -> one wavefront per row (no barrier at all), K <= 128 -> two.
// Gauss-Jordan without pivoting (the matrix is SPD, lambda > 0 on the diagonal):
//   column step c : every lane publishes its A[j][c] through a double-buffered LDS column; the pivot
//                   d = A[c][c] comes back by v_readlane (one wave) or LDS, the pivot ROW A[c][q] is, by the symmetry of
//                   the not yet eliminated block, the published COLUMN and comes back as wave-uniform
//                   (broadcast) ds_read_b128:
//                       A[j][q] -= (A[j][c]/d) * A[c][q],  b_j -= (A[j][c]/d) * b_c   for every row j != c, q > c
//                   -- one FMA per element, the same arithmetic for rows below AND above the pivot, so there is
//                   no triangle to skip, no square root and no back substitution: x_j = b_j / pivot_j at the end.
// Everything is unrolled over c and q so that the row is addressed by static register numbers.
template <int NW>
__device__ __forceinline__ void group_sync() {
    if constexpr (NW > 1) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // one wave: LDS executes in order; only stop compiler motion
}

__device__ __forceinline__ float rcp_nr(float d) {   // v_rcp_f32 + one Newton step
    const float y = __builtin_amdgcn_rcpf(d);
    return y * (2.0f - d * y);
}

using f32x2 = __attribute__((ext_vector_type(2))) float;

// column step C (template recursion instead of a loop: the row must be addressed by static register numbers,
// and a `#pragma unroll` the optimizer declines turns a[] into v_movrel indexing).  The row is kept as K/2
// register pairs and updated with v_pk_fma_f32 (two FMAs per instruction): half the issue slots and, as
// important at K=128, half the code -- the unrolled elimination is straight-line code that every wave streams
// through once per row, and it has to stay near the 64 KB instruction cache.  Pairs are updated whole: an entry
// left of the pivot column is dead, so touching it is harmless.
template <int K, int NW, int C>
__device__ __forceinline__ void gj_column(f32x2 (&a)[K / 2], float &bj, float &mypiv, int j, float *colbuf, float *bbuf) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    constexpr int KP = 64 * NW;   // one slot per lane: idle lanes (j >= K) publish into slots nobody reads, no branch
    float *cb = colbuf + (C & 1) * KP;          // column C was published by the previous step (column 0: by solve_reg)
    const float raw = a[C / 2][C & 1];
    float d, bc;
    group_sync<NW>();
    if constexpr (NW == 1) {      // the pivot row is a lane of this wave: no LDS round trip in front of the reciprocal
        d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, raw), C));
        bc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bj), C));
    } else {
        d = cb[C];
        bc = (bbuf + (C & 1) * KP)[C];
    }
    const float rd = rcp_nr(d);
    const float nt = j == C ? 0.0f : -(raw * rd);
    mypiv = j == C ? rd : mypiv;
    bj += nt * bc;
    const f32x2 nt2 = {nt, nt};
    const f4 *cb4 = reinterpret_cast<const f4 *>(cb);
    // The broadcast reads of the pivot row are issued a chunk (CH x 16 B) ahead of the FMAs that use them, into two
    // alternating register sets: the pins below are `asm volatile`, which no load may cross, so a read issued next to its
    // FMAs would be waited for on the spot (one full LDS latency per two v_pk_fma_f32 -- measured: the whole solve).
    constexpr int G0 = (C + 1) / 4, G1 = K / 4, CH = 8, NCH = (G1 - G0 + CH - 1) / CH;
    f4 buf[2][CH];
#pragma unroll
    for (int u = 0; u < CH; ++u)
        if (G0 + u < G1) buf[0][u] = cb4[G0 + u];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch + 1 < NCH) {
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (G0 + (ch + 1) * CH + u < G1) buf[(ch + 1) & 1][u] = cb4[G0 + (ch + 1) * CH + u];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int g = G0 + ch * CH + u;
            if (g < G1) {
                const f4 v = buf[ch & 1][u];
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (4 * g + 2 * h + 1 > C) {
                        const f32x2 vh = {v[2 * h], v[2 * h + 1]};
                        a[2 * g + h] = __builtin_elementwise_fma(nt2, vh, a[2 * g + h]);
                        // the update is pinned to its column: left alone, the compiler sinks the FMAs of far columns
                        // to their first use and keeps every broadcast value alive for them (K=32 took 390 VGPRs)
                        asm volatile("" : "+v"(a[2 * g + h]));
                        // look-ahead: the next pivot column is final as soon as its pair is updated -- it is published now
                        // (other buffer), so that the LDS write and the wait for the other wave run under the remaining FMAs
                        if (C + 1 < K && 2 * g + h == (C + 1) / 2) {
                            (colbuf + ((C + 1) & 1) * KP)[j] = a[(C + 1) / 2][(C + 1) & 1];
                            if (NW > 1) (bbuf + ((C + 1) & 1) * KP)[j] = bj;
                        }
                    }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (C + 1 < K) gj_column<K, NW, C + 1>(a, bj, mypiv, j, colbuf, bbuf);
}

template <int K, int NW>
__device__ __forceinline__ float solve_reg(f32x2 (&a)[K / 2], float bj, int j, float *colbuf, float *bbuf) {
    float mypiv = 0.0f;   // 1 / pivot of this lane's row
    colbuf[j] = a[0][0];  // column 0; every later column is published by the step before it
    if (NW > 1) bbuf[j] = bj;
    gj_column<K, NW, 0>(a, bj, mypiv, j, colbuf, bbuf);
    return bj * mypiv;
}

// t-th pair (m <= n) of the row-major upper triangle of a T32 x T32 tile grid
// (column by column: the last column block of the row registers comes to life only with the last T32 tiles, while most of the
// accumulators are already gone -- the order of the layout change is what bounds the register peak)
constexpr int tile_n(int t) { return (t >= 1) + (t >= 3) + (t >= 6); }   // T32 <= 4 (no loop: it has to fold inside `#pragma unroll` bodies)
constexpr int tile_m(int t) { return t - tile_n(t) * (tile_n(t) + 1) / 2; }

constexpr int WMF_STAGE_LD = 36;                  // stride of a staged tile row: 16-byte aligned rows, conflict-free ds_read_b128
constexpr int WMF_STAGE = 32 * WMF_STAGE_LD;      // one 32x32 tile

template <int T32, int NW>
constexpr size_t wmf_reg_smem() { return sizeof(float) * ((size_t)NW * WMF_STAGE + 4 * 64 * NW + 32 * T32 + 2 * NW + 4); }

// Round Q of the layout change: every wave stages its Q-th tile, then every lane takes what belongs to its
// row out of the NW staged tiles.  Template recursion over rounds and tiles: the tile coordinates of a staged
// tile must be compile-time constants, they select the registers a[32 n + e] the values go to.
// Lane j (row block jb = j / 32) gets column block c of its row from exactly ONE tile, (min(jb,c), max(jb,c)) -- its row
// jl when jb is the tile's row block, its column jl (the transpose) otherwise -- so the take is an assignment: the row
// registers come to life block by block while the accumulators die tile by tile, and the two never have to be held
// together (holding all of both cost 450 spilled VGPRs per lane and row at K=128).  A per-lane select keeps the lanes of the
// other row blocks unchanged.  (Skipping, by a wave-uniform branch, the parts of a tile none of the wave's row blocks uses was
// tried: with the branch the compiler leaves half of the row in a stack array -- 7000 scratch accesses per row.)
template <int T32, int NW, int W, int Q>
__device__ __forceinline__ void stage_take(f32x2 (&a)[16 * T32], const float *stage, int jb, int jl, int wave) {
    using f4 = __attribute__((ext_vector_type(4))) float;
    constexpr int NT = T32 * (T32 + 1) / 2;
    constexpr int t = W + NW * Q;
    if constexpr (t < NT) {
        constexpr int m = tile_m(t), n = tile_n(t);
        const float *st = stage + W * WMF_STAGE;
        {
            const bool hit = jb == m;
            const f4 *row = reinterpret_cast<const f4 *>(st + jl * WMF_STAGE_LD);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f4 v = row[g];
                a[16 * n + 2 * g] = hit ? f32x2{v[0], v[1]} : a[16 * n + 2 * g];
                a[16 * n + 2 * g + 1] = hit ? f32x2{v[2], v[3]} : a[16 * n + 2 * g + 1];
            }
        }
        if constexpr (m != n) {             // the mirrored block reads the tile's transpose
            const bool hit = jb == n;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const f32x2 v = {st[2 * e * WMF_STAGE_LD + jl], st[(2 * e + 1) * WMF_STAGE_LD + jl]};
                a[16 * m + e] = hit ? v : a[16 * m + e];
            }
        }
    }
    if constexpr (W + 1 < NW) stage_take<T32, NW, W + 1, Q>(a, stage, jb, jl, wave);
}

// Every wave holds a partial sum of EVERY tile (the gathered rows are dealt to the waves step by step, see the kernel):
// wave 0 writes its partials of the round's NW tiles into the stage, then the other wave adds its own (plain read-add-write,
// nobody else touches the stage in that phase), then all lanes take.  (ds_add_f32 instead of the read-add-write was tried:
// ~700 cycles per wave instruction, 6 ms per K=128 user sweep.  A second stage region per wave, summed by the take, doubles
// the take's reads in flight and with them the register peak: 940 spilled VGPRs.)
template <int T32, int NW, int PH, int Q>
__device__ __forceinline__ void stage_put(const f32x16 (&acc)[T32 * (T32 + 1) / 2], const float (&bsum)[T32], float *stage,
                                          float *bvec, float weight, int li, int lh) {
    constexpr int NT = T32 * (T32 + 1) / 2;
#pragma unroll
    for (int W = 0; W < NW; ++W) {
        const int t = NW * Q + W;
        if (t < NT) {
            const int m = tile_m(t), n = tile_n(t);
            float *st = stage + W * WMF_STAGE;
            if (m == n) {
                const float tot = (bsum[m] + __shfl_xor(bsum[m], 32, 64)) * weight;
                if (lh == 0) bvec[32 * m + li] = PH == 0 ? tot : bvec[32 * m + li] + tot;
            }
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = PH == 0 ? 0.0f : st[((r & 3) + 8 * (r >> 2) + 4 * lh) * WMF_STAGE_LD + li];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                st[((r & 3) + 8 * (r >> 2) + 4 * lh) * WMF_STAGE_LD + li] = __builtin_fmaf(acc[t][r], weight - 1.0f, old[r]);
        }
    }
}

template <int T32, int NW, int Q>
__device__ __forceinline__ void stage_rounds(f32x2 (&a)[16 * T32], const f32x16 (&acc)[T32 * (T32 + 1) / 2], const float (&bsum)[T32],
                                             float *stage, float *bvec, float weight, int jb, int jl, int wave, int li, int lh) {
    constexpr int NT = T32 * (T32 + 1) / 2;
    group_sync<NW>();   // the stage (and, first round, the previous row's column buffers / bvec) is free
    if (wave == 0) stage_put<T32, NW, 0, Q>(acc, bsum, stage, bvec, weight, li, lh);
    if constexpr (NW > 1) {
        group_sync<NW>();
        if (wave != 0) stage_put<T32, NW, 1, Q>(acc, bsum, stage, bvec, weight, li, lh);
    }
    group_sync<NW>();
    stage_take<T32, NW, 0, Q>(a, stage, jb, jl, wave);
    if constexpr (NW * (Q + 1) < NT) stage_rounds<T32, NW, Q + 1>(a, acc, bsum, stage, bvec, weight, jb, jl, wave, li, lh);
}

// One row per workgroup of NW wavefronts (K = 32*T32 <= 64*NW).  Every wave builds partial sums of all Gramian tiles over
// its share of the gathered rows; the tiles go through a small LDS stage to turn the MFMA C layout into "lane j holds row j"
// (rows read the tile, the mirrored block reads its transpose), and the system is solved in registers.  Little LDS and
// <= 168 VGPRs at K <= 64 on purpose: the column steps are a latency chain, three waves per SIMD fill it.
template <int T32, int NW>
__global__ __launch_bounds__(64 * NW, NW == 1 ? 3 : 2) void wmf_row_reg_kernel(int32_t rows, const int32_t *__restrict__ indptr,
                                                             const int32_t *__restrict__ indices, float *__restrict__ X,
                                                             const float *__restrict__ Y, const float *__restrict__ A0,
                                                             float weight, int32_t long_threshold, int probe,
                                                             const int32_t *__restrict__ order, const float *__restrict__ A0t) {
    constexpr int K = 32 * T32;
    constexpr int NT = T32 * (T32 + 1) / 2;
    extern __shared__ unsigned char smem_raw[];
    float *stage = reinterpret_cast<float *>(smem_raw);   // [NW][32][WMF_STAGE_LD]
    float *colbuf = stage + ((NW * WMF_STAGE + 3) & ~3);   // [2][64 NW], 16-byte aligned
    float *bbuf = colbuf + 2 * 64 * NW;                    // [2][64 NW]
    float *bvec = bbuf + 2 * 64 * NW;                      // [K]
    const int tid = threadIdx.x, lane = tid & 63, wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    for (int32_t wi = blockIdx.x; wi < rows; wi += gridDim.x) {
        const int32_t i = order ? order[wi] : wi;                // work item wi: the wi-th longest row
        const int32_t p0 = indptr[i];
        int32_t p1 = indptr[i + 1];
        int j = tid;
        asm volatile("" : "+v"(j));   // opaque per row: keeps the 2K lane comparisons of the solve from being hoisted into (spilled) SGPR pairs
        if (p0 == p1) {                                        // wmf.pyx:154-156
            if (j < K) X[(int64_t)i * K + j] = 0.0f;
            continue;
        }
        if (long_threshold > 0 && p1 - p0 > long_threshold) continue;   // built from segments
        if (probe == 2) p1 = p0;                                         // timing probe: no gather, no MFMA
        // Every wave accumulates EVERY tile, over its share of the gathered rows (the k=2 MFMA steps are dealt to the waves in
        // turn): a gathered row is read from memory by one wave only -- with the tiles dealt out instead, both waves of a K=128
        // row read all of it, and the gather (L2 misses into a 14-70 MB table) is what bounds this phase.  The partial tiles meet
        // in the LDS stage (stage_rounds).  The accumulators start from the tile of A0 = YtY + lambda I (tile t on wave t % NW), pre-divided by
        // (w - 1): what is staged later, acc (w - 1), is then the finished A = A0 + (w - 1) G.
        f32x16 acc[NT];
        float bsum[T32];
        const float inv_w1 = 1.0f / (weight - 1.0f);             // (the host sends w == 1 to the other kernel)
#pragma unroll
        for (int m = 0; m < T32; ++m) bsum[m] = 0.0f;
        {
            // the accumulators start from A0 / (w - 1) in the tile layout of wmf_tile_layout_kernel: four 16-byte loads per tile
            using f4 = __attribute__((ext_vector_type(4))) float;
            int a0off = lane * 4;
            asm volatile("" : "+v"(a0off));
            const f4 *a0t = reinterpret_cast<const f4 *>(A0t) + a0off;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (wave == t % NW) {                            // each tile of A0 enters the sum once; the waves share the loads
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f4 v = a0t[t * 256 + q];
                        acc[t][4 * q] = v[0]; acc[t][4 * q + 1] = v[1]; acc[t][4 * q + 2] = v[2]; acc[t][4 * q + 3] = v[3];
                    }
                } else {
                    acc[t] = (f32x16)(0.0f);
                }
            }
        }
        // 64 gathered rows per batch (lane l holds the index of entry l), 32 k=2 steps per batch dealt to the waves in turn, 8 steps
        // per group: each of the T32 32-column chunks of a gathered row is loaded once per step and feeds every tile that uses it
        // (tile (m, n) multiplies chunk m by chunk n).  (A variant that issued the loads of group g + 1 before the MFMAs of group g
        // costs a wave per SIMD in registers and measured slower: K=128 user sweep 13.1 -> 15.9 ms, K=64 unchanged.)
        // (one 32-bit byte offset per gathered row, chunk offsets as immediates, unconditional loads selected afterwards, no MFMAs
        // past the row's end: as in wmf_row_blk_kernel; the host sends tables beyond 4 GB to the older kernels)
        const char *Yb = reinterpret_cast<const char *>(Y);
        for (int32_t pb = p0; pb < p1; pb += 64) {
            const int32_t myp = pb + lane;
            const int32_t myidx = myp < p1 ? indices[myp] : -1;
            const int nb = p1 - pb < 64 ? p1 - pb : 64;
            const int steps = (nb + 1) >> 1;
            for (int s0 = wave; s0 < steps; s0 += 8 * NW) {     // steps wave, wave + NW, ... (at most 32 per batch)
                float ch[8][T32];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int32_t idx = __shfl(myidx, 2 * (s0 + NW * u) + lh, 64);
                    const bool ok = idx >= 0 && s0 + NW * u < steps;
                    const uint32_t off = (uint32_t)(ok ? idx : 0) * (uint32_t)(K * 4) + (uint32_t)(li * 4);
#pragma unroll
                    for (int m = 0; m < T32; ++m) {
                        const float v = *reinterpret_cast<const float *>(Yb + off + 128 * m);
                        ch[u][m] = ok ? v : 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (s0 + NW * u < steps) {                  // (uniform)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ch[u][tile_m(t)], ch[u][tile_n(t)], acc[t], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < T32; ++m) bsum[m] += ch[u][m];
                    }
                }
            }
        }
        // lane j <- row j of A = A0 + (w-1) G
        const int jr = j < K ? j : 0;
        const int jb = jr >> 5, jl = jr & 31;
        f32x2 a[K / 2];
#pragma unroll
        for (int g = 0; g < K / 2; ++g) a[g] = f32x2{0.0f, 0.0f};   // (constants: folded into the first select of each block)
        stage_rounds<T32, NW, 0>(a, acc, bsum, stage, bvec, weight, jb, jl, wave, li, lh);
        group_sync<NW>();
        const float bj0 = bvec[jr];
        const float x = probe == 1 ? a[0][0] + bj0 : solve_reg<K, NW>(a, bj0, j, colbuf, bbuf);   // (1: timing probe, no solve)
        if (j < K) X[(int64_t)i * K + j] = x;                  // wmf.pyx:170-171
    }
}

// ---------------------------------------------------------------------------------------------
// Blocked Cholesky in the MFMA accumulator layout (K = 96, 128): ONE wavefront per row, no LDS, no layout change.
// The system stays where the Gramian was accumulated -- upper-triangular 32 x 32 tiles T(m, n), m <= n, in the C/D layout of
// v_mfma_f32_32x32x2_f32 (lane l: column l & 31; register r: row (r & 3) + 8 (r >> 2) + 4 (l >> 5)) -- and is factored
// block row by block row, A = U^T U (SPD, no pivoting):
//     block row p, 32 steps   row c of every tile T(p, n >= p) is scaled and subtracted from the rows below it: one rank-1
//                             MFMA per tile and step (chol_step); afterwards T(p,p) = U_pp, T(p,n) = S(p,n) = U_pp^-T T(p,n)
//     T(m,n) -= S(p,m)^T S(p,n), p < m <= n   16 MFMAs per tile: registers r of S(p,m) and S(p,n) ARE the A and B operands
//                             (lane i of register r holds S[k][i], k this half's row of the pair) -- VGPR operands as they
//                             lie, no broadcast through LDS
//     y = U^-T b rides along in lane layout; x_p = U_pp^-1 (y_p - sum_{n>p} S(p,n) x_n) on the way back, with U_pp transposed
//     through the matrix unit so that its columns come in lane layout too (back_step).
// Per K=128 row: 320 + 160 + 64 MFMAs (the register Gauss-Jordan above spends 2 x 8192 v_pk_fma_f32 and a 128-step latency
// chain on two waves), and two independent waves per SIMD.  History: the first blocked version eliminated with explicit
// inverses of the 32 x 32 pivot blocks (symmetric sweep, R = -P^-1 T): 256 + 128 MFMAs, but 5-6 x the error of the
// Gauss-Jordan on lambda-dominated systems (tools/wmf_accuracy.py: 1.1e-4 against 1.9e-5 relative to the f64 oracle at K=128)
// -- products with P^-1 carry cond(P), products with U^-1 its square root.
constexpr int blk_tix(int m, int n) { return n * (n + 1) / 2 + m; }   // = the order of tile_m / tile_n

// A0 / (w - 1) re-laid for wmf_row_blk_kernel: tile t, lane l, register r at [(t * 64 + l) * 16 + r], so that a row's 16 starting
// values per tile are four 16-byte loads (160 scalar loads and multiplies per row otherwise; the matrix is 64 KB, L2-resident).
template <int T32>
__global__ void wmf_tile_layout_kernel(const float *__restrict__ A0, float inv_w1, float *__restrict__ out) {
    constexpr int K = 32 * T32, NT = T32 * (T32 + 1) / 2;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NT * 64 * 16) return;
    const int r = e & 15, l = (e >> 4) & 63, t = e >> 10;
    int m = 0, n = 0;
    for (int q = 0; q < NT; ++q)
        if (q == t) { m = tile_m(q); n = tile_n(q); }
    const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = 32 * n + (l & 31);
    out[e] = A0[row * K + col] * inv_w1;
}

// Cross-half moves without the LDS pipe (gfx950): v_permlane32_swap exchanges the upper half of one register with the lower
// half of another, so swapping a value with a copy of itself yields {its lower half in both halves, its upper half in both
// halves}; v_permlane16_swap does the same for the odd / even rows of 16 lanes.  Inline assembly on purpose: a swap issued
// directly after the VALU instruction that wrote its operand reads stale lanes, and hipcc 7.2 inserts no wait state
// (tools/micro/permlane.hip: half-wave sums off by one row; one s_nop 0 in front is enough for both).
__device__ __forceinline__ void swap32_self(float v, float &lo_all, float &hi_all) {
    // volatile: the swap must execute with all lanes enabled -- sunk into the not-taken side of a per-half select it would run
    // under a half-wave exec mask and move nothing
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2\n\ts_nop 0\n\tv_permlane32_swap_b32 %0, %1" : "=&v"(lo_all), "=&v"(hi_all) : "v"(v));
}
__device__ __forceinline__ float half_bcast(float v, int half) {   // lanes l and l + 32 <- v of lane l + 32 * half
    float lo, hi;
    swap32_self(v, lo, hi);
    return half ? hi : lo;
}
__device__ __forceinline__ float other_half(float v, int lh) {     // v of lane l ^ 32
    float lo, hi;
    swap32_self(v, lo, hi);
    return lh ? lo : hi;
}

__device__ __forceinline__ float rsq_nr(float d) {   // v_rsq_f32 + one Newton step
    const float y = __builtin_amdgcn_rsqf(d);
    return y * (1.5f - 0.5f * d * y * y);
}
__device__ __forceinline__ float lane_value(float v, int l) {   // v_readlane_b32
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// sum over the 32 lanes of each half-wave, returned in every lane of the half
__device__ __forceinline__ float half_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    float even, odd;   // {rows 0 0 2 2, rows 1 1 3 3}
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2\n\ts_nop 0\n\tv_permlane16_swap_b32 %0, %1" : "=&v"(even), "=&v"(odd) : "v"(v));
    return even + odd;
}

// Step C of the Cholesky factorisation T(P,P) = U^T U of a diagonal tile in the C layout, with the same row operations applied
// to a tile E that starts as the identity (afterwards E = U^-T, lower triangular).  Row C of a tile lies in register RC of
// the lanes of half LC, element j in lane j -- which is exactly the operand layout of v_mfma_f32_32x32x2_f32 (lane l supplies
// A[l & 31][l >> 5] and B[l >> 5][l & 31]); with the other half's operands zero, one MFMA per tile is the rank-1 update of the
// step:  row C <- row C / sqrt(d),  rows i > C -= U[C][i] * (row C).  U[C][i] is taken from row C of the tile itself: the
// trailing part is symmetric up to rounding.  Rows <= C are not touched (A operand masked); what the updates write below the
// diagonal of the tile is never read.
// Two pivots per MFMA: rows C and C + 1 (C even) lie in registers RC and RC + 1 of the SAME half; row C + 1 takes the update of
// step C on the VALU (one FMA per tile), and the two rank-1 updates of the rows below go through the matrix unit together --
// the k=2 instruction has room for two outer products: this half supplies row C, the other half (by v_permlane32_swap) row C + 1.
template <int C>
__device__ __forceinline__ void chol_step(f32x16 &D, f32x16 &E, int li, int lh) {
    static_assert((C & 1) == 0, "pivot pairs start at even rows");
    constexpr int RC = (C & 3) + 4 * (C >> 3), LC = (C >> 2) & 1;
    const bool mine = lh == LC;
    const float p0 = D[RC];
    const float rs0 = rsq_nr(lane_value(p0, C + 32 * LC));
    const float u0 = p0 * rs0, e0 = E[RC] * rs0;
    const float f = lane_value(u0, C + 1 + 32 * LC);            // U[C][C+1]
    const float p1 = __builtin_fmaf(-f, u0, D[RC + 1]);
    const float rs1 = rsq_nr(lane_value(p1, C + 1 + 32 * LC));
    const float u1 = p1 * rs1, e1 = __builtin_fmaf(-f, e0, E[RC + 1]) * rs1;
    D[RC] = mine ? u0 : D[RC];
    D[RC + 1] = mine ? u1 : D[RC + 1];
    E[RC] = mine ? e0 : E[RC];
    E[RC + 1] = mine ? e1 : E[RC + 1];
    const float u1o = other_half(u1, lh), e1o = other_half(e1, lh);   // (all lanes: see swap32_self)
    const float ub = mine ? u0 : u1o;                           // B operand: row C in half LC, row C + 1 in the other half
    const float eb = mine ? e0 : e1o;
    const float aop = li > C + 1 ? -ub : 0.0f;                  // rows below the pair only
    D = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, ub, D, 0, 0, 0);
    E = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, eb, E, 0, 0, 0);
    if constexpr (C + 2 < 32) chol_step<C + 2>(D, E, li, lh);
}

// value of lane-layout vector v (element j in lanes j and j + 32) at this lane's row of register r
__device__ __forceinline__ float to_reg_layout(float v, int r, int lh) { return __shfl(v, (r & 3) + 8 * (r >> 2) + 4 * lh, 64); }

template <int T32, int P>
__device__ __forceinline__ void chol_panels(f32x16 (&acc)[T32 * (T32 + 1) / 2], float (&bl)[T32], float (&y)[T32], int li, int lh) {
    f32x16 &D = acc[blk_tix(P, P)];
    {
        f32x16 E;
#pragma unroll
        for (int r = 0; r < 16; ++r) E[r] = li == (r & 3) + 8 * (r >> 2) + 4 * lh ? 1.0f : 0.0f;
        __builtin_amdgcn_sched_barrier(0);
        chol_step<0>(D, E, li, lh);
        __builtin_amdgcn_sched_barrier(0);
        D = E;                                                  // U is dead; U^-T stays for the back substitution
    }
    if constexpr (P + 1 < T32) {
        // V = U^-1 = E^T through the matrix unit (D[i][j] = sum_k E[k][i] I[k][j]): register r of V is the A operand of E * T
        f32x16 V = (f32x16)(0.0f);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            V = __builtin_amdgcn_mfma_f32_32x32x2f32(D[r], li == (r & 3) + 8 * (r >> 2) + 4 * lh ? 1.0f : 0.0f, V, 0, 0, 0);
        float part = 0.0f;                                      // y_P = U^-T b_P = E b_P: column sums of V against b in register layout
#pragma unroll
        for (int r = 0; r < 16; ++r) part = __builtin_fmaf(V[r], to_reg_layout(bl[P], r, lh), part);
        y[P] = part + other_half(part, lh);
#pragma unroll
        for (int n = P + 1; n < T32; ++n) {                     // S(P,n) = U^-T T(P,n) = E T(P,n); b_n -= S(P,n)^T y_P
            f32x16 R = (f32x16)(0.0f);
#pragma unroll
            for (int r = 0; r < 16; ++r) R = __builtin_amdgcn_mfma_f32_32x32x2f32(V[r], acc[blk_tix(P, n)][r], R, 0, 0, 0);
            float yp = y[P];
            asm volatile("" : "+v"(yp));                        // y_P in register layout is re-fetched per tile: held across the tiles it costs 16 registers
            float pn = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) pn = __builtin_fmaf(R[r], to_reg_layout(yp, r, lh), pn);
            bl[n] -= pn + other_half(pn, lh);
            acc[blk_tix(P, n)] = R;
        }
        __builtin_amdgcn_sched_barrier(0);
        // T(m,n) -= S(P,m)^T S(P,n), P < m <= n: registers r of S(P,m) and of S(P,n) are the operands as they lie.  The tiles
        // of the later block rows are kept NEGATED until their own block row is factored (the kernel negates them once after
        // the Gramian, the block row is negated back below): the update is then an addition and needs no negated copy of S(P,m).
#pragma unroll
        for (int m = P + 1; m < T32; ++m)
#pragma unroll
            for (int n = m; n < T32; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[blk_tix(m, n)] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[blk_tix(P, m)][r], acc[blk_tix(P, n)][r], acc[blk_tix(m, n)], 0, 0, 0);
#pragma unroll
        for (int n = P + 1; n < T32; ++n) acc[blk_tix(P + 1, n)] = -acc[blk_tix(P + 1, n)];   // block row P + 1 comes next
        chol_panels<T32, P + 1>(acc, bl, y, li, lh);
    } else {
        // last block row: y_P = E b_P directly, row sums of E against b in lane layout
        float sel = 0.0f;
        const int rj = (li & 3) + 4 * (li >> 3), lj = (li >> 2) & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sr = half_sum(D[r] * bl[P]);
            sel = r == rj ? sr : sel;
        }
        const float other = other_half(sel, lh);
        y[P] = lh == lj ? sel : other;
    }
}

// ---- K = 128: the same elimination with idle tiles parked in LDS -------------------------------------------------------
// Ten accumulator tiles are 160 of the 256 registers two waves per SIMD leave a lane; with the working tiles of a block row
// (E, V, R) on top the compiler kept whole tiles in scratch and fetched them back around every phase (1 KB per lane, and a
// scratch round trip is a memory latency inside what is already a latency chain).  The schedule below says explicitly which
// four tiles are NOT needed next and keeps them in the workgroup's 16 KB of LDS (one wave per workgroup, eight per CU: 128 of
// 160 KB): at most six tiles plus the three working tiles are in registers at any time.
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void park_put(f32x4 *park, int slot, int lane, const f32x16 &t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) park[(slot * 4 + q) * 64 + lane] = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
}
__device__ __forceinline__ void park_get(const f32x4 *park, int slot, int lane, f32x16 &t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = park[(slot * 4 + q) * 64 + lane];
        t[4 * q] = v[0]; t[4 * q + 1] = v[1]; t[4 * q + 2] = v[2]; t[4 * q + 3] = v[3];
    }
}
// D <- U^-T of the diagonal tile D (chol_step), V <- U^-1, returns y = U^-T b (lane layout)
__device__ __forceinline__ float blk_factor(f32x16 &D, f32x16 &V, float b, int li, int lh) {
    {
        f32x16 E;
#pragma unroll
        for (int r = 0; r < 16; ++r) E[r] = li == (r & 3) + 8 * (r >> 2) + 4 * lh ? 1.0f : 0.0f;
        __builtin_amdgcn_sched_barrier(0);
        chol_step<0>(D, E, li, lh);
        __builtin_amdgcn_sched_barrier(0);
        D = E;
    }
    V = (f32x16)(0.0f);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        V = __builtin_amdgcn_mfma_f32_32x32x2f32(D[r], li == (r & 3) + 8 * (r >> 2) + 4 * lh ? 1.0f : 0.0f, V, 0, 0, 0);
    float part = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part = __builtin_fmaf(V[r], to_reg_layout(b, r, lh), part);
    return part + other_half(part, lh);
}
// T <- S = U^-T T (A operand: registers of V), bn -= S^T y
__device__ __forceinline__ void blk_panel_tile(const f32x16 &V, f32x16 &T, float y, float &bn, int lh) {
    f32x16 R = (f32x16)(0.0f);
#pragma unroll
    for (int r = 0; r < 16; ++r) R = __builtin_amdgcn_mfma_f32_32x32x2f32(V[r], T[r], R, 0, 0, 0);
    asm volatile("" : "+v"(y));
    float pn = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) pn = __builtin_fmaf(R[r], to_reg_layout(y, r, lh), pn);
    bn -= pn + other_half(pn, lh);
    T = R;
}
// C += A^T B (C is carried negated: see chol_panels)
__device__ __forceinline__ void blk_trail(const f32x16 &A, const f32x16 &B, f32x16 &C) {
#pragma unroll
    for (int r = 0; r < 16; ++r) C = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r], B[r], C, 0, 0, 0);
}
__device__ __forceinline__ void blk_trail_parked(const f32x16 &A, const f32x16 &B, f32x4 *park, int slot, int lane) {
    f32x16 C;
    park_get(park, slot, lane, C);
    blk_trail(A, B, C);
    park_put(park, slot, lane, C);
}
// s[r] += S[r] * x (row sums of S against x are finished by blk_rows_to_lanes)
__device__ __forceinline__ void blk_row_acc(const f32x16 &S, float x, float (&s)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_fmaf(S[r], x, s[r]);
}
__device__ __forceinline__ float blk_rows_to_lanes(float (&s)[16], int li, int lh) {   // sum over the lanes of each register's row -> lane layout
    const int rj = (li & 3) + 4 * (li >> 3), lj = (li >> 2) & 1;
    float sel = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float sr = half_sum(s[r]);
        sel = r == rj ? sr : sel;
    }
    const float other = other_half(sel, lh);
    return lh == lj ? sel : other;
}
__device__ __forceinline__ float blk_apply_Et(const f32x16 &E, float w, int lh) {      // U^-1 w = E^T w: column sums of E against w in register layout
    float part = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part = __builtin_fmaf(E[r], to_reg_layout(w, r, lh), part);
    return part + other_half(part, lh);
}

// acc: the ten tiles of a K=128 system (later block rows negated), bl: b in lane layout; x <- the solution (lane layout per block)
__device__ __forceinline__ void blk_solve_parked4(f32x16 (&acc)[10], float (&bl)[4], float (&x)[4], f32x4 *park, int lane, int li, int lh) {
    f32x16 &t00 = acc[blk_tix(0, 0)], &t01 = acc[blk_tix(0, 1)], &t02 = acc[blk_tix(0, 2)], &t03 = acc[blk_tix(0, 3)];
    f32x16 &t11 = acc[blk_tix(1, 1)], &t12 = acc[blk_tix(1, 2)], &t13 = acc[blk_tix(1, 3)];
    f32x16 &t22 = acc[blk_tix(2, 2)], &t23 = acc[blk_tix(2, 3)], &t33 = acc[blk_tix(3, 3)];
    float y[4];
    park_put(park, 0, lane, t13);
    park_put(park, 1, lane, t22);
    park_put(park, 2, lane, t23);
    park_put(park, 3, lane, t33);
    {   // block row 0
        f32x16 V;
        y[0] = blk_factor(t00, V, bl[0], li, lh);
        blk_panel_tile(V, t01, y[0], bl[1], lh);
        blk_panel_tile(V, t02, y[0], bl[2], lh);
        blk_panel_tile(V, t03, y[0], bl[3], lh);
    }
    __builtin_amdgcn_sched_barrier(0);
    blk_trail(t01, t01, t11);
    blk_trail(t01, t02, t12);
    blk_trail_parked(t01, t03, park, 0, lane);
    blk_trail_parked(t02, t02, park, 1, lane);
    blk_trail_parked(t02, t03, park, 2, lane);
    blk_trail_parked(t03, t03, park, 3, lane);
    park_get(park, 0, lane, t13);
    park_put(park, 0, lane, t00);                              // E_0 rests until the back substitution
    t11 = -t11; t12 = -t12; t13 = -t13;
    __builtin_amdgcn_sched_barrier(0);
    {   // block row 1
        f32x16 V;
        y[1] = blk_factor(t11, V, bl[1], li, lh);
        blk_panel_tile(V, t12, y[1], bl[2], lh);
        blk_panel_tile(V, t13, y[1], bl[3], lh);
    }
    __builtin_amdgcn_sched_barrier(0);
    blk_trail_parked(t12, t12, park, 1, lane);
    blk_trail_parked(t12, t13, park, 2, lane);
    blk_trail_parked(t13, t13, park, 3, lane);
    park_get(park, 1, lane, t22);
    park_get(park, 2, lane, t23);
    park_put(park, 1, lane, t01);
    park_put(park, 2, lane, t02);
    t22 = -t22; t23 = -t23;
    __builtin_amdgcn_sched_barrier(0);
    {   // block row 2
        f32x16 V;
        y[2] = blk_factor(t22, V, bl[2], li, lh);
        blk_panel_tile(V, t23, y[2], bl[3], lh);
    }
    __builtin_amdgcn_sched_barrier(0);
    blk_trail_parked(t23, t23, park, 3, lane);
    park_get(park, 3, lane, t33);
    park_put(park, 3, lane, t03);
    t33 = -t33;
    __builtin_amdgcn_sched_barrier(0);
    {   // block row 3: y_3 = E b_3 directly (row sums of E against b in lane layout)
        f32x16 E;
#pragma unroll
        for (int r = 0; r < 16; ++r) E[r] = li == (r & 3) + 8 * (r >> 2) + 4 * lh ? 1.0f : 0.0f;
        __builtin_amdgcn_sched_barrier(0);
        chol_step<0>(t33, E, li, lh);
        __builtin_amdgcn_sched_barrier(0);
        t33 = E;
        float s[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
        blk_row_acc(t33, bl[3], s);
        y[3] = blk_rows_to_lanes(s, li, lh);
    }
    // back substitution
    x[3] = blk_apply_Et(t33, y[3], lh);
    __builtin_amdgcn_sched_barrier(0);
    {
        float s[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
        blk_row_acc(t23, x[3], s);
        x[2] = blk_apply_Et(t22, y[2] - blk_rows_to_lanes(s, li, lh), lh);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        float s[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
        blk_row_acc(t12, x[2], s);
        blk_row_acc(t13, x[3], s);
        x[1] = blk_apply_Et(t11, y[1] - blk_rows_to_lanes(s, li, lh), lh);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        float s[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
        f32x16 T;
        park_get(park, 1, lane, T);
        blk_row_acc(T, x[1], s);
        park_get(park, 2, lane, T);
        blk_row_acc(T, x[2], s);
        park_get(park, 3, lane, T);
        blk_row_acc(T, x[3], s);
        const float w = y[0] - blk_rows_to_lanes(s, li, lh);
        park_get(park, 0, lane, T);
        x[0] = blk_apply_Et(T, w, lh);
    }
}

template <int T32>
__global__ __launch_bounds__(64, 2) void wmf_row_blk_kernel(int32_t rows, const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices, float *__restrict__ X,
                                                           const float *__restrict__ Y, const float *__restrict__ A0,
                                                           float weight, int32_t long_threshold, int probe,
                                                           const int32_t *__restrict__ order, unsigned long long *__restrict__ phase_ticks) {
    constexpr int K = 32 * T32;
    constexpr int NT = T32 * (T32 + 1) / 2;
    constexpr int GS = T32 >= 4 ? 4 : 8;
    for (int32_t wi = blockIdx.x; wi < rows; wi += gridDim.x) {
        const int32_t i = order ? order[wi] : wi;                // work item wi: the wi-th longest row
        unsigned long long tk[5] = {0, 0, 0, 0, 0};              // CYMF_WMF_PROBE=5: s_memtime at the phase boundaries (tools/wmf_probe.py)
        if (probe == 5) tk[0] = __builtin_amdgcn_s_memtime();
        // the lane number is made opaque per row and again per phase: everything derived from it (permute index vectors, tile
        // corner offsets, the 32 pivot-lane masks of a sweep) would otherwise be hoisted out of the row loop and held in
        // registers across all phases (108 VGPRs at K=32, 970 bytes of scratch per lane at K=128)
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));
        int li = lane & 31, lh = lane >> 5;
        const int32_t p0 = indptr[i];
        int32_t p1 = indptr[i + 1];
        if (p0 == p1) {                                        // wmf.pyx:154-156
#pragma unroll
            for (int m = 0; m < T32; ++m) X[(int64_t)i * K + 32 * m + li] = 0.0f;
            continue;
        }
        if (long_threshold > 0 && p1 - p0 > long_threshold) continue;   // built from segments
        if (probe == 2) p1 = p0;                                         // timing probe: no gather, no Gramian
        // T = A0 / (w - 1) + G: the system scaled by 1 / (w - 1), b scaled with it
        f32x16 acc[NT];
        float bsum[T32];
        const float inv_w1 = 1.0f / (weight - 1.0f);
#pragma unroll
        for (int m = 0; m < T32; ++m) bsum[m] = 0.0f;
        {
            using f4 = __attribute__((ext_vector_type(4))) float;
            const f4 *a0t = reinterpret_cast<const f4 *>(A0) + lane * 4;   // A0: the tile layout of wmf_tile_layout_kernel
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 v = a0t[t * 256 + q];
                    acc[t][4 * q] = v[0]; acc[t][4 * q + 1] = v[1]; acc[t][4 * q + 2] = v[2]; acc[t][4 * q + 3] = v[3];
                }
        }
        // 64 gathered rows per batch (lane l holds the index of entry l), GS k=2 steps per group (8; 4 at K=128, where the 160
        // accumulator registers leave no room for 32 operands: the compiler then spills a whole tile and reloads it around
        // every step, 3500 cycles per step instead of 1280): each of the T32 32-column chunks
        // of a gathered row is loaded once per step and feeds every tile that uses it.  One 32-bit byte offset per gathered row,
        // the chunk offsets are immediates (the host sends tables beyond 4 GB to the other kernel); the loads are unconditional
        // -- entry 0 of the table stands in for a missing row -- and selected after.
        const char *Yb = reinterpret_cast<const char *>(Y);
        if (probe == 5) { asm volatile("" : "+v"(acc[NT - 1][15])); tk[1] = __builtin_amdgcn_s_memtime(); }
        for (int32_t pb = p0; pb < p1; pb += 64) {
            const int32_t myp = pb + lane;
            const int32_t myidx = myp < p1 ? indices[myp] : -1;
            const int nb = p1 - pb < 64 ? p1 - pb : 64;
            const int steps = (nb + 1) >> 1;
            for (int s0 = 0; s0 < steps; s0 += GS) {
                float ch[GS][T32];
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    const int32_t idx = __shfl(myidx, 2 * (s0 + u) + lh, 64);
                    const bool ok = idx >= 0;
                    const uint32_t off = (uint32_t)(ok ? idx : 0) * (uint32_t)(K * 4) + (uint32_t)(li * 4);
#pragma unroll
                    for (int m = 0; m < T32; ++m) {
                        const float v = *reinterpret_cast<const float *>(Yb + off + 128 * m);
                        ch[u][m] = ok ? v : 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    if (s0 + u < steps) {                       // (uniform)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ch[u][tile_m(t)], ch[u][tile_n(t)], acc[t], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < T32; ++m) bsum[m] += ch[u][m];
                    }
                }
            }
        }
        asm volatile("" : "+v"(lane));
        li = lane & 31;
        lh = lane >> 5;
        if (probe == 5) { asm volatile("" : "+v"(acc[0][0])); tk[2] = __builtin_amdgcn_s_memtime(); }
        float bl[T32], y[T32];                                  // b_m, y_m = (U^-T b)_m: element j in lanes j and j + 32
#pragma unroll
        for (int m = 0; m < T32; ++m) bl[m] = (bsum[m] + other_half(bsum[m], lh)) * (weight * inv_w1);
        if (probe == 1) {                                       // timing probe: no elimination
            float sacc = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[t][r];
            const float out = sacc + bl[0];
            if (lh == 0) X[(int64_t)i * K + li] = out;
            continue;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (tile_m(t) >= 1) acc[t] = -acc[t];              // (see chol_panels: later block rows are carried negated)
        float x[T32];
        if constexpr (T32 == 4) {
            extern __shared__ f32x4 park_lds[];                 // [4 tiles][4][64 lanes]
            blk_solve_parked4(acc, bl, x, park_lds, lane, li, lh);
            if (probe == 5) { asm volatile("" : "+v"(x[0])); tk[3] = __builtin_amdgcn_s_memtime(); }
        } else {
            chol_panels<T32, 0>(acc, bl, y, li, lh);
            if (probe == 5) { asm volatile("" : "+v"(y[T32 - 1])); tk[3] = __builtin_amdgcn_s_memtime(); }
            // back substitution: x_p = U_pp^-1 (y_p - sum_{n > p} S(p,n) x_n) = E_p^T (...): the sums over n are row sums of the S tiles
            // against x in lane layout, the product with E^T a column sum against the result in register layout
            const int rj = (li & 3) + 4 * (li >> 3), lj = (li >> 2) & 1;   // where row li of a tile lives
#pragma unroll
            for (int p = T32 - 1; p >= 0; --p) {
                __builtin_amdgcn_sched_barrier(0);
                float w = y[p];
                if (p + 1 < T32) {
                    float sel = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float sr = 0.0f;
#pragma unroll
                        for (int n = p + 1; n < T32; ++n) sr = __builtin_fmaf(acc[blk_tix(p, n)][r], x[n], sr);
                        sr = half_sum(sr);                          // (S(p,.) x)[k(r, lh)], in every lane of the half
                        sel = r == rj ? sr : sel;
                    }
                    const float other = other_half(sel, lh);
                    w -= lh == lj ? sel : other;
                }
                float part = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) part = __builtin_fmaf(acc[blk_tix(p, p)][r], to_reg_layout(w, r, lh), part);
                x[p] = part + other_half(part, lh);
            }
        }
        if (lh == 0) {
#pragma unroll
            for (int m = 0; m < T32; ++m) X[(int64_t)i * K + 32 * m + li] = x[m];   // wmf.pyx:170-171
        }
        if (probe == 5) {
            asm volatile("" : "+v"(x[0]));
            tk[4] = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) atomicAdd(phase_ticks + q, tk[q + 1] - tk[q]);
                atomicAdd(phase_ticks + 4, 1ull);
            }
        }
    }
}

// Segment of a long row (or of the identity list: YtY), 4 waves: as in wmf_row_reg_kernel every wave accumulates every tile
// over its share of the k=2 steps (wave w takes steps w, w+4, ...), so a gathered row is fetched by ONE wave -- with the tiles
// dealt to the waves (wmf_row_mfma_kernel<T32, true>, kept for CYMF_WMF_LDS_SOLVE) every wave fetched the chunks of its own
// tiles, 20 chunk loads per gathered row pair where 4 suffice.  No LDS, no barrier; the partial tiles and column sums are
// added to scratch[slot] (K*K + K floats) with float atomics, the mirrored block as well (the finish kernel and the readers of
// YtY take the full matrix).
template <int T32>
__global__ __launch_bounds__(WMF_THREADS, 2) void wmf_seg_kernel(const int32_t *__restrict__ indices, const float *__restrict__ Y,
                                                             const WmfSeg *__restrict__ segs, int32_t n_segs,
                                                             float *__restrict__ scratch) {
    constexpr int K = 32 * T32;
    constexpr int NT = T32 * (T32 + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    for (int32_t item = blockIdx.x; item < n_segs; item += gridDim.x) {
        const WmfSeg sg = segs[item];
        const int32_t p0 = sg.begin, p1 = sg.end;
        f32x16 acc[NT];
        float bsum[T32];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x16)(0.0f);
#pragma unroll
        for (int m = 0; m < T32; ++m) bsum[m] = 0.0f;
        const char *Yb = reinterpret_cast<const char *>(Y);
        for (int32_t pb = p0; pb < p1; pb += 64) {
            const int32_t myp = pb + lane;
            const int32_t myidx = myp < p1 ? indices[myp] : -1;
            const int nb = p1 - pb < 64 ? p1 - pb : 64;
            const int steps = (nb + 1) >> 1;                     // <= 32: one pass of 8 steps per wave
            if (wave < steps) {
                // one 32-bit byte offset per gathered row, chunk offsets as immediates, unconditional loads selected afterwards
                // (as in wmf_row_blk_kernel; the host sends tables beyond 4 GB to the tile-dealing kernel)
                float ch[8][T32];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int32_t idx = __shfl(myidx, 2 * (wave + 4 * u) + lh, 64);
                    const bool ok = idx >= 0 && wave + 4 * u < steps;
                    const uint32_t off = (uint32_t)(ok ? idx : 0) * (uint32_t)(K * 4) + (uint32_t)(li * 4);
#pragma unroll
                    for (int m = 0; m < T32; ++m) {
                        const float v = *reinterpret_cast<const float *>(Yb + off + 128 * m);
                        ch[u][m] = ok ? v : 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (wave + 4 * u < steps) {                  // (uniform)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ch[u][tile_m(t)], ch[u][tile_n(t)], acc[t], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < T32; ++m) bsum[m] += ch[u][m];
                    }
                }
            }
        }
        float *G = scratch + (size_t)sg.slot * (K * K + K);
#pragma unroll
        for (int m = 0; m < T32; ++m) {
            const float tot = bsum[m] + __shfl_xor(bsum[m], 32, 64);
            if (lh == 0) atomicAdd(G + K * K + 32 * m + li, tot);
        }
        float *gd = G + 4 * lh * K + li, *gm = G + li * K + 4 * lh;   // this lane's corner of a tile / of its mirror image
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                atomicAdd(gd + (32 * tile_m(t) + rr) * K + 32 * tile_n(t), acc[t][r]);
                if (tile_m(t) != tile_n(t)) atomicAdd(gm + 32 * tile_n(t) * K + 32 * tile_m(t) + rr, acc[t][r]);
            }
        }
    }
}

// Long rows, finished with the register solve: one workgroup of NW waves per row, lane j loads row j of
// A = A0 + (w-1) G and b_j = w * sum_j straight from the scratch the segments accumulated (64 scattered row reads per
// instruction, but only ~10^3 rows) and runs solve_reg.  The in-LDS Cholesky finish below took 0.5-0.7 ms per item sweep
// (24-100 us per row, two workgroups per CU), a tenth of the K=64 epoch.
template <int T32, int NW>
__global__ __launch_bounds__(64 * NW, NW == 1 ? 3 : 2) void wmf_long_reg_kernel(const int32_t *__restrict__ long_rows,
                                                                              float *__restrict__ X, const float *__restrict__ A0,
                                                                              float *scratch, float weight) {
    constexpr int K = 32 * T32;
    using f4 = __attribute__((ext_vector_type(4))) float;
    extern __shared__ unsigned char smem_raw[];
    float *colbuf = reinterpret_cast<float *>(smem_raw);   // [2][64 NW]
    float *bbuf = colbuf + 2 * 64 * NW;                    // [2][64 NW]
    int j = threadIdx.x;
    asm volatile("" : "+v"(j));
    const int jr = j < K ? j : 0;
    const int32_t i = long_rows[blockIdx.x];
    float *G = scratch + (size_t)blockIdx.x * (K * K + K);
    const f4 *a0 = reinterpret_cast<const f4 *>(A0 + (size_t)jr * K);
    f4 *g4 = reinterpret_cast<f4 *>(G + (size_t)jr * K);
    f32x2 a[K / 2];
#pragma unroll
    for (int g = 0; g < K / 4; ++g) {
        const f4 v = a0[g] + (weight - 1.0f) * g4[g];
        a[2 * g] = f32x2{v[0], v[1]};
        a[2 * g + 1] = f32x2{v[2], v[3]};
    }
    const float bj = weight * G[K * K + jr];
    // the slot is left zeroed for the next half-sweep's segments: no fill has to run (and be waited for) in front of them
    // (lanes past K read row 0 again and may see these zeros: their values are never used)
    if (j < K) {
#pragma unroll
        for (int g = 0; g < K / 4; ++g) g4[g] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        G[K * K + j] = 0.0f;
    }
    const float x = solve_reg<K, NW>(a, bj, j, colbuf, bbuf);
    if (j < K) X[(int64_t)i * K + j] = x;
}

// long rows: A = A0 + (w-1) G, b = w * sum, both from the scratch the segments accumulated
__global__ __launch_bounds__(WMF_THREADS) void wmf_long_finish_kernel(int K, const int32_t *__restrict__ long_rows,
                                                                     float *__restrict__ X, const float *__restrict__ A0,
                                                                     const float *__restrict__ scratch, float weight) {
    extern __shared__ unsigned char smem_raw[];
    const int lda = K + 1;
    float *A = reinterpret_cast<float *>(smem_raw);
    float *b = A + K * lda;
    float *dg = b + K;
    const int tid = threadIdx.x;
    const int32_t i = long_rows[blockIdx.x];
    const float *G = scratch + (size_t)blockIdx.x * (K * K + K);
    for (int e = tid; e < K * K; e += WMF_THREADS) A[(e / K) * lda + (e % K)] = A0[e] + (weight - 1.0f) * G[e];
    for (int k = tid; k < K; k += WMF_THREADS) b[k] = weight * G[K * K + k];
    __syncthreads();
    chol_solve_lds<float>(A, b, dg, K, lda);
    for (int k = tid; k < K; k += WMF_THREADS) X[(int64_t)i * K + k] = b[k];
}

}  // namespace
}  // namespace cymf

using namespace cymf;

template <typename T>
struct WmfStore {
    DevBuf<T> W, H, G;
    DevBuf<T> A_wide;   // K > ~130 (f64) / ~190 (f32): per-workgroup K x (K+1) systems of the generic row kernel
    // workgroup-private scratch, every element written before it is read inside one kernel by the same CU: plain cached memory
    WmfStore() { A_wide.fine = 0; }
};

namespace cymf {
int comm_allgatherv(cymf_comm *c, void *d_buf, const int64_t *row_bounds, int64_t row_bytes, hipStream_t s);   // comm.hip
int comm_world(cymf_comm *c);
int comm_rank(cymf_comm *c);
}  // namespace cymf

struct cymf_wmf {
    int32_t U = 0, I = 0, K = 0;
    int dtype = 0, device = 0;
    double weight = 10.0, wd = 0.01;
    hipStream_t stream = nullptr;
    hipStream_t seg_stream = nullptr;       // the long rows' segments run beside the whole rows (CYMF_WMF_SEG_STREAM=0: on `stream`)
    hipEvent_t ev_ready = nullptr, ev_segs = nullptr;
    WmfStore<float> f32;
    WmfStore<double> f64;
    DevBuf<int32_t> d_indptr, d_indices, d_tindptr, d_tindices;
    bool have_data = false, have_params = false;
    bool use_mfma = true;
    // multi-GPU: rows of each side are cut into one contiguous range per rank, balanced by
    // entries + a constant per solve; a rank solves its rows and the ranges are all-gathered after the sweep
    cymf_comm *comm = nullptr;
    int shard_rank = 0, shard_world = 1;   // from comm; CYMF_WMF_FAKE_SHARD="r/w" (tests) sets them without a communicator: no gather
    std::vector<int64_t> bounds[2];   // [world + 1] row boundaries per side (empty = single GPU)
    int probe = 0;           // CYMF_WMF_PROBE: 1 skips the solve, 2 the Gramian (timing only, results invalid)
    int row_order = 1;       // CYMF_WMF_ROW_ORDER=0: rows in index order instead of longest first
    int blocked = -1;        // CYMF_WMF_BLOCKED: 1/0 force/forbid the blocked elimination (wmf_row_blk_kernel); default: K >= 96
    bool reg_solve = true;   // register-resident solve (wmf_row_reg_kernel); CYMF_WMF_LDS_SOLVE=1 selects the in-LDS one
    // rows with more than long_threshold entries are built from segments (MFMA path)
    int32_t long_threshold = 2048;
    int32_t seg_len = 0;            // 0: default (see set_data); CYMF_WMF_SEG
    DevBuf<cymf::WmfSeg> d_segs[2];
    DevBuf<int32_t> d_long_rows[2];
    DevBuf<float> d_a0t;            // (YtY + lambda I) / (w - 1) in the accumulator tile layout (wmf_tile_layout_kernel)
    DevBuf<unsigned long long> d_ticks;   // CYMF_WMF_PROBE=5: [side][8] phase ticks of wmf_row_blk_kernel
    DevBuf<int32_t> d_order[2];     // this rank's whole rows (offsets from its first row), longest first: the row kernels' work list
    int32_t n_order[2] = {0, 0};
    int32_t n_segs[2] = {0, 0}, n_long[2] = {0, 0};
    DevBuf<float> d_scratch;
    bool scratch_dirty = false;     // the last finish kernel did not leave its slots zeroed (the in-LDS finish)
    // YtY on the MFMA path: identity index list and segments over the rows of each table (0 = W, 1 = H)
    DevBuf<int32_t> d_iota;
    DevBuf<cymf::WmfSeg> d_gram_segs[2];
    int32_t n_gram_segs[2] = {0, 0};
};

template <typename F>
static int allow_lds(F kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        CYMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

template <typename T>
static int wmf_half(cymf_wmf *h, WmfStore<T> &st, int side) {
    const int K = h->K;
    const int32_t rows = side == 0 ? h->U : h->I, cols = side == 0 ? h->I : h->U;
    T *X = side == 0 ? st.W.p : st.H.p;
    const T *Y = side == 0 ? st.H.p : st.W.p;
    const int32_t *ip = side == 0 ? h->d_indptr.p : h->d_tindptr.p;
    const int32_t *ix = side == 0 ? h->d_indices.p : h->d_tindices.p;
    // this rank's rows [lo, hi): the whole-row kernels get the CSR and the output offset by lo (indptr values are
    // absolute positions in indices); the long-row lists were already cut to the range in set_data
    int32_t lo = 0, hi = rows;
    if (!h->bounds[side].empty()) {
        lo = (int32_t)h->bounds[side][h->shard_rank];
        hi = (int32_t)h->bounds[side][h->shard_rank + 1];
    }
    T *const X_all = X;
    X += (size_t)lo * K;
    ip += lo;
    const int32_t rows_all = rows;
    (void)rows_all;
    const int32_t my_rows = hi - lo;
    CYMF_TRY(st.G.alloc((size_t)K * K + K));   // K*K Gramian (+ K: the column sums the MFMA path also produces)
    CYMF_TRY(st.G.zero(h->stream));
    const bool mfma_ok = sizeof(T) == 4 && h->use_mfma && K % 32 == 0 && K <= 128;
    // the one-wave kernels address the gathered table with 32-bit byte offsets: tables of 4 GB and more take the older kernels
    const bool reg_ok = h->reg_solve && (int64_t)cols * K * 4 < ((int64_t)1 << 32);
    if (mfma_ok) {
        // YtY on the MFMA units: Y as ONE long "row" over the identity index list, cut into segments, each
        // workgroup adding its partial 32x32 tiles into G (the segment instantiation of the row kernel)
        if constexpr (sizeof(T) == 4) {
            const int g = side == 0 ? 1 : 0;   // Y is the item table for the user sweep and vice versa
            size_t smem = sizeof(float) * ((size_t)K * (K + 1) + 2 * K);
            const int ns = h->n_gram_segs[g];
#define WMF_GRAM_(T32_)                                                                                                       \
    do {                                                                                                                      \
        if (reg_ok) {                                                                                                         \
            hipLaunchKernelGGL((wmf_seg_kernel<T32_>), dim3(ns), dim3(WMF_THREADS), 0, h->stream, h->d_iota.p,                \
                               reinterpret_cast<const float *>(Y), h->d_gram_segs[g].p, ns, reinterpret_cast<float *>(st.G.p)); \
        } else {                                                                                                              \
            CYMF_TRY(allow_lds(wmf_row_mfma_kernel<T32_, true>, smem));                                                       \
            hipLaunchKernelGGL((wmf_row_mfma_kernel<T32_, true>), dim3(ns), dim3(WMF_THREADS), smem, h->stream, 0, nullptr,    \
                               h->d_iota.p, nullptr, reinterpret_cast<const float *>(Y), nullptr, 0.0f, 0, h->d_gram_segs[g].p, \
                               ns, reinterpret_cast<float *>(st.G.p));                                                        \
        }                                                                                                                     \
    } while (0)
            switch (K / 32) {
            case 1: WMF_GRAM_(1); break;
            case 2: WMF_GRAM_(2); break;
            case 3: WMF_GRAM_(3); break;
            default: WMF_GRAM_(4); break;
            }
#undef WMF_GRAM_
        }
    } else if (K > 128) {   // YtY for any K: 16 x 16 output tiles over slabs of rows
        const int n_kb = (K + 15) / 16;
        const int slabs = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)cols + 255) / 256, 64));
        hipLaunchKernelGGL(wmf_gram_wide_kernel<T>, dim3(n_kb * n_kb, slabs), dim3(256), 0, h->stream, Y, (int64_t)cols, K, n_kb, st.G.p);
    } else {   // YtY + lambda I  (wmf.pyx:142-143)
        int grid = (int)std::min<int64_t>(((int64_t)cols + WMF_TILE - 1) / WMF_TILE, 1024);
        size_t smem = sizeof(T) * WMF_TILE * K;
        hipLaunchKernelGGL(wmf_gram_kernel<T>, dim3(grid), dim3(WMF_THREADS), smem, h->stream, Y, (int64_t)cols, K, st.G.p);
    }
    hipLaunchKernelGGL(wmf_add_diag_kernel<T>, dim3((K + 63) / 64), dim3(64), 0, h->stream, st.G.p, K, (T)h->wd);
    CYMF_HIP(hipGetLastError());
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(my_rows, 256 * 16));
    bool mfma = false;
    if constexpr (sizeof(T) == 4) {
        if (h->use_mfma && K % 32 == 0 && K <= 128) {
            mfma = true;
            size_t smem = sizeof(float) * ((size_t)K * (K + 1) + 2 * K);
            const float *Yf = reinterpret_cast<const float *>(Y);
            float *Xf = reinterpret_cast<float *>(X);
            float *Xf_all = reinterpret_cast<float *>(X_all);
            const float *Gf = reinterpret_cast<const float *>(st.G.p);
            const int32_t nseg = h->n_segs[side], nlong = h->n_long[side];
            const WmfSeg *segs = h->d_segs[side].p;
            // The long rows' segments and the whole rows are independent until the long rows are finished: the segments go to a
            // second stream, so that the two kernels' workgroups mix on the CUs and only one tail is paid.
            hipStream_t ss = (h->seg_stream && reg_ok && nseg > 0) ? h->seg_stream : h->stream;
            if (ss != h->stream) {
                CYMF_HIP(hipEventRecord(h->ev_ready, h->stream));   // YtY + lambda I is complete, the previous sweep's table too
                CYMF_HIP(hipStreamWaitEvent(ss, h->ev_ready, 0));
            }
            // The register finish kernel leaves its slots zeroed; the in-LDS finish does not -- and neither does a half-sweep that
            // launched segments and then returned early (an error between here and the finish launch): the scratch counts as dirty
            // from the moment segments may be written until the register finish kernel is enqueued behind them.
            if (nlong > 0 && (!reg_ok || h->scratch_dirty))
                CYMF_HIP(hipMemsetAsync(h->d_scratch.p, 0, (size_t)nlong * ((size_t)K * K + K) * sizeof(float), ss));
            if (nlong > 0) h->scratch_dirty = true;
            struct SegJoin {   // an early return must not leave the segment stream running beside the next call
                cymf_wmf *h; hipStream_t ss; bool armed;
                void join() {
                    if (!armed) return;
                    armed = false;
                    if (hipEventRecord(h->ev_segs, ss) == hipSuccess) (void)hipStreamWaitEvent(h->stream, h->ev_segs, 0);
                }
                ~SegJoin() { join(); }
            } seg_join{h, ss, ss != h->stream};
            const int grid_seg = (int)std::min<int64_t>(nseg, 256 * 16);
            const bool sorted = h->row_order && h->d_order[side].p;
            const int32_t *order = sorted ? h->d_order[side].p : nullptr;
            const int32_t n_work = sorted ? h->n_order[side] : my_rows;
#define WMF_LAUNCH_(T32_)                                                                                                   \
    do {                                                                                                                    \
        CYMF_TRY(allow_lds(wmf_row_mfma_kernel<T32_, false>, smem));                                                        \
        CYMF_TRY(allow_lds(wmf_row_mfma_kernel<T32_, true>, smem));                                                         \
        if (nseg > 0 && reg_ok)   /* segments of the long rows first: the longest work starts earliest */                   \
            hipLaunchKernelGGL((wmf_seg_kernel<T32_>), dim3(grid_seg), dim3(WMF_THREADS), 0, ss, ix, Yf, segs, nseg,         \
                               h->d_scratch.p);                                                                             \
        else if (nseg > 0)                                                                                                  \
            hipLaunchKernelGGL((wmf_row_mfma_kernel<T32_, true>), dim3(grid_seg), dim3(WMF_THREADS), smem, h->stream, my_rows, \
                               ip, ix, Xf, Yf, Gf, (float)h->weight, h->long_threshold, segs, nseg, h->d_scratch.p);        \
        if (my_rows > 0 && reg_ok && h->weight != 1.0) {   /* A0 / (w - 1) in the accumulator tile layout for the one-row kernels */ \
            constexpr int n_a0t = (T32_) * ((T32_) + 1) / 2 * 1024;                                                         \
            CYMF_TRY(h->d_a0t.alloc(n_a0t));                                                                                \
            hipLaunchKernelGGL((wmf_tile_layout_kernel<T32_>), dim3(n_a0t / 256), dim3(256), 0, h->stream, Gf,               \
                               (float)(1.0 / (h->weight - 1.0)), h->d_a0t.p);                                               \
        }                                                                                                                   \
        if (my_rows <= 0) {                                                                                                 \
        } else if (reg_ok && h->weight != 1.0 && (h->blocked >= 0 ? h->blocked != 0 : (T32_) >= 3)) {                      \
            const int grid_b = (int)std::max<int64_t>(1, std::min<int64_t>(n_work, 256 * 64));                              \
            hipLaunchKernelGGL((wmf_row_blk_kernel<T32_>), dim3(grid_b), dim3(64), (T32_) == 4 ? 16384 : 0, h->stream, n_work, ip, ix, Xf, Yf, h->d_a0t.p, \
                               (float)h->weight, nlong > 0 ? h->long_threshold : 0, h->probe, order, h->d_ticks.p);         \
        } else if (reg_ok && h->weight != 1.0) {                                                                           \
            constexpr int NW_ = (T32_) <= 2 ? 1 : 2;                                                                        \
            const size_t smem_r = wmf_reg_smem<T32_, NW_>();                                                                \
            const int grid_r = (int)std::max<int64_t>(1, std::min<int64_t>(n_work, 256 * 64));                              \
            CYMF_TRY(allow_lds(wmf_row_reg_kernel<T32_, NW_>, smem_r));                                                     \
            hipLaunchKernelGGL((wmf_row_reg_kernel<T32_, NW_>), dim3(grid_r), dim3(64 * NW_), smem_r, h->stream,             \
                               n_work, ip, ix, Xf, Yf, Gf, (float)h->weight, nlong > 0 ? h->long_threshold : 0, h->probe, order, h->d_a0t.p); \
        } else {                                                                                                            \
            hipLaunchKernelGGL((wmf_row_mfma_kernel<T32_, false>), dim3(grid), dim3(WMF_THREADS), smem, h->stream, my_rows, ip, \
                               ix, Xf, Yf, Gf, (float)h->weight, nlong > 0 ? h->long_threshold : 0, segs, nseg, h->d_scratch.p); \
        }                                                                                                                   \
    } while (0)
            switch (K / 32) {
            case 1: WMF_LAUNCH_(1); break;
            case 2: WMF_LAUNCH_(2); break;
            case 3: WMF_LAUNCH_(3); break;
            default: WMF_LAUNCH_(4); break;
            }
#undef WMF_LAUNCH_
            seg_join.join();
            if (nlong > 0 && reg_ok) {
#define WMF_FINISH_(T32_)                                                                                                   \
    do {                                                                                                                    \
        constexpr int NW_ = (T32_) <= 2 ? 1 : 2;                                                                            \
        hipLaunchKernelGGL((wmf_long_reg_kernel<T32_, NW_>), dim3(nlong), dim3(64 * NW_), sizeof(float) * 4 * 64 * NW_,      \
                           h->stream, h->d_long_rows[side].p, Xf_all, Gf, h->d_scratch.p, (float)h->weight);                \
    } while (0)
                switch (K / 32) {
                case 1: WMF_FINISH_(1); break;
                case 2: WMF_FINISH_(2); break;
                case 3: WMF_FINISH_(3); break;
                default: WMF_FINISH_(4); break;
                }
#undef WMF_FINISH_
                h->scratch_dirty = false;   // enqueued: its slots are zero again when it has run
            } else if (nlong > 0) {
                CYMF_TRY(allow_lds(wmf_long_finish_kernel, smem));
                hipLaunchKernelGGL(wmf_long_finish_kernel, dim3(nlong), dim3(WMF_THREADS), smem, h->stream, K, h->d_long_rows[side].p, Xf_all, Gf,
                                   h->d_scratch.p, (float)h->weight);
            }
        }
    }
    if (!mfma) {
        size_t smem = sizeof(T) * ((size_t)K * (K + 1) + 2 * K + (size_t)WMF_TILE * K);
        T *scratch = nullptr;
        int grid_g = grid;
        if (smem > 160 * 1024) {   // the system does not fit the CU's LDS: one global K x (K+1) slice per workgroup
            grid_g = (int)std::max<int64_t>(1, std::min<int64_t>(my_rows, 256 * 4));
            CYMF_TRY(st.A_wide.alloc((size_t)grid_g * K * (K + 1)));
            scratch = st.A_wide.p;
            smem = sizeof(T) * (2 * (size_t)K + (size_t)WMF_TILE * K);
        }
        CYMF_TRY(allow_lds(wmf_row_kernel<T>, smem));
        if (my_rows > 0)
            hipLaunchKernelGGL(wmf_row_kernel<T>, dim3(grid_g), dim3(WMF_THREADS), smem, h->stream, my_rows, K, ip, ix, X, Y, st.G.p, (T)h->weight, scratch);
    }
    CYMF_HIP(hipGetLastError());
    if (h->probe == 5) {   // phase ticks of wmf_row_blk_kernel (diagnostic: synchronises)
        unsigned long long t[8] = {0};
        CYMF_HIP(hipMemcpyAsync(t, h->d_ticks.p, sizeof(t), hipMemcpyDeviceToHost, h->stream));
        CYMF_HIP(hipStreamSynchronize(h->stream));
        if (t[4])
            fprintf(stderr, "[wmf probe 5] side %d: %llu rows; s_memtime ticks per row: A0 %.0f, Gramian %.0f, factorisation %.0f, back substitution %.0f\n",
                    side, t[4], (double)t[0] / t[4], (double)t[1] / t[4], (double)t[2] / t[4], (double)t[3] / t[4]);
        CYMF_TRY(h->d_ticks.zero(h->stream));
    }
    if (h->comm && !h->bounds[side].empty())   // every rank ends the half-sweep with the whole updated table
        CYMF_TRY(comm_allgatherv(h->comm, X_all, h->bounds[side].data(), (int64_t)K * (int64_t)sizeof(T), h->stream));
    return 0;
}

extern "C" int cymf_wmf_create(cymf_wmf **out, int32_t U, int32_t I, int32_t K, double weight, double weight_decay,
                               int dtype, int device) {
    if (!out) return fail(CYMF_ERR_INVALID, "cymf_wmf_create: out is NULL");
    *out = nullptr;
    if (U <= 0 || I <= 0 || K <= 0) return fail(CYMF_ERR_INVALID, "cymf_wmf_create: U, I, K must be positive");
    if ((int64_t)K * K > ((int64_t)1 << 26)) return fail(CYMF_ERR_UNSUPPORTED, "cymf_wmf_create: K=%d: a K x K system per workgroup beyond 2^26 entries", K);
    if (dtype != CYMF_F32 && dtype != CYMF_F64) return fail(CYMF_ERR_INVALID, "cymf_wmf_create: dtype %d", dtype);
    CYMF_TRY(use_device(device));
    cymf_wmf *h = new cymf_wmf();
    h->U = U; h->I = I; h->K = K; h->weight = weight; h->wd = weight_decay; h->dtype = dtype; h->device = device;
    const char *env = getenv("CYMF_WMF_NO_MFMA");
    h->use_mfma = !(env && env[0] == '1');
    if (const char *e4 = getenv("CYMF_WMF_PROBE")) h->probe = atoi(e4);
    if (const char *e7 = getenv("CYMF_WMF_BLOCKED")) h->blocked = atoi(e7);
    if (const char *e8 = getenv("CYMF_WMF_ROW_ORDER")) h->row_order = atoi(e8);
    if (const char *e5 = getenv("CYMF_WMF_FAKE_SHARD")) {
        int r = 0, w = 1;
        if (sscanf(e5, "%d/%d", &r, &w) == 2 && w >= 1 && r >= 0 && r < w) { h->shard_rank = r; h->shard_world = w; }
    }
    if (const char *e3 = getenv("CYMF_WMF_LDS_SOLVE")) h->reg_solve = !(e3[0] == '1');
    if (const char *e2 = getenv("CYMF_WMF_LONG")) h->long_threshold = std::max(64, atoi(e2));
    if (const char *e3 = getenv("CYMF_WMF_SEG")) h->seg_len = std::max(64, atoi(e3));
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(CYMF_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    const char *e9 = getenv("CYMF_WMF_SEG_STREAM");
    if (!(e9 && e9[0] == '0')) {
        if (create_side_stream(&h->seg_stream) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_segs, hipEventDisableTiming) != hipSuccess) {
            cymf_wmf_destroy(h);
            return fail(CYMF_ERR_HIP, "cymf_wmf_create: side stream");
        }
    }
    *out = h;
    return 0;
}

extern "C" int cymf_wmf_attach_comm(cymf_wmf *h, cymf_comm *c) {
    if (!h || !c) return fail(CYMF_ERR_INVALID, "cymf_wmf_attach_comm: bad arguments");
    if (h->have_data) return fail(CYMF_ERR_INVALID, "cymf_wmf_attach_comm must precede cymf_wmf_set_data");
    h->comm = c;
    h->shard_rank = comm_rank(c);
    h->shard_world = comm_world(c);
    return 0;
}

extern "C" int cymf_wmf_row_range(cymf_wmf *h, int side, int32_t *lo, int32_t *hi) {
    if (!h || (side != 0 && side != 1) || !lo || !hi || !h->have_data) return fail(CYMF_ERR_INVALID, "cymf_wmf_row_range: bad arguments / no data");
    *lo = 0;
    *hi = side == 0 ? h->U : h->I;
    if (!h->bounds[side].empty()) { *lo = (int32_t)h->bounds[side][h->shard_rank]; *hi = (int32_t)h->bounds[side][h->shard_rank + 1]; }
    return 0;
}

static int check_csr(const int32_t *indptr, const int32_t *indices, int32_t rows, int32_t cols, const char *what) {
    if (indptr[0] != 0) return fail(CYMF_ERR_INVALID, "cymf_wmf_set_data: %s indptr[0] != 0", what);
    for (int32_t r = 0; r < rows; ++r)
        if (indptr[r] > indptr[r + 1]) return fail(CYMF_ERR_INVALID, "cymf_wmf_set_data: %s indptr not monotone", what);
    for (int32_t p = 0; p < indptr[rows]; ++p)
        if (indices[p] < 0 || indices[p] >= cols) return fail(CYMF_ERR_INVALID, "cymf_wmf_set_data: %s index out of range", what);
    return 0;
}

extern "C" int cymf_wmf_set_data(cymf_wmf *h, const int32_t *indptr, const int32_t *indices, const int32_t *t_indptr,
                                 const int32_t *t_indices) {
    if (!h || !indptr || !t_indptr) return fail(CYMF_ERR_INVALID, "cymf_wmf_set_data: bad arguments");
    CYMF_TRY(use_device(h->device));
    const int64_t nnz = indptr[h->U];
    if (nnz != t_indptr[h->I] || (nnz > 0 && (!indices || !t_indices))) return fail(CYMF_ERR_INVALID, "cymf_wmf_set_data: CSR / transposed CSR disagree");
    CYMF_TRY(check_csr(indptr, indices, h->U, h->I, "X"));
    CYMF_TRY(check_csr(t_indptr, t_indices, h->I, h->U, "X^T"));
    CYMF_TRY(h->d_indptr.upload(indptr, (size_t)h->U + 1, h->stream));
    CYMF_TRY(h->d_indices.upload(indices, (size_t)nnz, h->stream));
    CYMF_TRY(h->d_tindptr.upload(t_indptr, (size_t)h->I + 1, h->stream));
    CYMF_TRY(h->d_tindices.upload(t_indices, (size_t)nnz, h->stream));
    // long rows -> segments of long_threshold entries (both sides)
    size_t max_long = 0;
    for (int side = 0; side < 2; ++side) {
        const int32_t *ip = side == 0 ? indptr : t_indptr;
        const int32_t rows = side == 0 ? h->U : h->I;
        int32_t r_lo = 0, r_hi = rows;
        h->bounds[side].clear();
        if (h->shard_world > 1) {   // contiguous row ranges of equal cost (entries + a constant per solve), the same on every rank
            const int world = h->shard_world;
            const int64_t per_solve = std::max(16, h->K / 2);
            const int64_t total = (int64_t)ip[rows] + per_solve * rows;
            h->bounds[side].assign((size_t)world + 1, rows);
            h->bounds[side][0] = 0;
            int next = 1;
            for (int32_t r = 0; r < rows && next < world; ++r) {
                const int64_t before = (int64_t)ip[r] + per_solve * r;
                while (next < world && before >= total * next / world) h->bounds[side][next++] = r;
            }
            r_lo = (int32_t)h->bounds[side][h->shard_rank];
            r_hi = (int32_t)h->bounds[side][h->shard_rank + 1];
        }
        std::vector<WmfSeg> segs;
        std::vector<int32_t> longs;
        for (int32_t r = r_lo; r < r_hi; ++r) {
            const int32_t n = ip[r + 1] - ip[r];
            if (n <= h->long_threshold) continue;
            const int32_t slot = (int32_t)longs.size();
            longs.push_back(r);
            // wmf_seg_kernel: every wave of a segment adds all tiles to the scratch, four times the atomics of the
            // tile-dealing kernel per segment -- so its segments are four times as long
            const int32_t seg_len = h->seg_len > 0 ? h->seg_len : (h->reg_solve ? 4 : 1) * h->long_threshold;
            for (int32_t b = ip[r]; b < ip[r + 1]; b += seg_len)
                segs.push_back(WmfSeg{slot, b, std::min(b + seg_len, ip[r + 1]), 0});
        }
        // Work list of the whole-row kernels: longest row first.  Row lengths follow a power law (C4 users: mean 144, 1 % above
        // 736, up to the threshold), and with rows dealt out in index order the sweep ended when the unluckiest workgroup did:
        // +33 % over the mean on C4 (greedy schedule of the measured lengths).  Longest-first is the classic LPT rule.
        std::vector<int32_t> order;
        order.reserve((size_t)(r_hi - r_lo));
        for (int32_t r = r_lo; r < r_hi; ++r)
            if (ip[r + 1] - ip[r] <= h->long_threshold) order.push_back(r - r_lo);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return ip[r_lo + a + 1] - ip[r_lo + a] > ip[r_lo + b + 1] - ip[r_lo + b];
        });
        h->n_order[side] = (int32_t)order.size();
        CYMF_TRY(h->d_order[side].upload(order.data(), order.size(), h->stream));
        // (segments stay in row order: sorted longest first, the segments of one very long row run together and contend for
        // the atomics on its scratch slot -- measured 3.3 instead of 2.9 ms for the C4 item sweep at K=128)
        h->n_segs[side] = (int32_t)segs.size();
        h->n_long[side] = (int32_t)longs.size();
        CYMF_TRY(h->d_segs[side].upload(segs.data(), segs.size(), h->stream));
        CYMF_TRY(h->d_long_rows[side].upload(longs.data(), longs.size(), h->stream));
        max_long = std::max(max_long, longs.size());
    }
    CYMF_TRY(h->d_scratch.alloc(std::max<size_t>(1, max_long * ((size_t)h->K * h->K + h->K))));
    CYMF_TRY(h->d_scratch.zero(h->stream));
    CYMF_TRY(h->d_ticks.alloc(8));
    CYMF_TRY(h->d_ticks.zero(h->stream));
    {
        std::vector<int32_t> iota((size_t)std::max(h->U, h->I));
        for (size_t v = 0; v < iota.size(); ++v) iota[v] = (int32_t)v;
        CYMF_TRY(h->d_iota.upload(iota.data(), iota.size(), h->stream));
        for (int g = 0; g < 2; ++g) {
            const int32_t n = g == 0 ? h->U : h->I;
            std::vector<WmfSeg> segs;
            for (int32_t b = 0; b < n; b += 512) segs.push_back(WmfSeg{0, b, std::min(b + 512, n), 0});
            h->n_gram_segs[g] = (int32_t)segs.size();
            CYMF_TRY(h->d_gram_segs[g].upload(segs.data(), segs.size(), h->stream));
        }
    }
    CYMF_HIP(hipStreamSynchronize(h->stream));
    h->have_data = true;
    return 0;
}

extern "C" int cymf_wmf_upload(cymf_wmf *h, const double *W, const double *H) {
    if (!h || !W || !H) return fail(CYMF_ERR_INVALID, "cymf_wmf_upload: bad arguments");
    CYMF_TRY(use_device(h->device));
    const size_t nW = (size_t)h->U * h->K, nH = (size_t)h->I * h->K;
    if (h->dtype == CYMF_F32) { CYMF_TRY(upload_f64(h->f32.W, W, nW, h->stream)); CYMF_TRY(upload_f64(h->f32.H, H, nH, h->stream)); }
    else { CYMF_TRY(upload_f64(h->f64.W, W, nW, h->stream)); CYMF_TRY(upload_f64(h->f64.H, H, nH, h->stream)); }
    CYMF_HIP(hipStreamSynchronize(h->stream));
    h->have_params = true;
    return 0;
}

extern "C" int cymf_wmf_download(cymf_wmf *h, double *W, double *H) {
    if (!h || !W || !H || !h->have_params) return fail(CYMF_ERR_INVALID, "cymf_wmf_download: bad arguments / no params");
    CYMF_TRY(use_device(h->device));
    const size_t nW = (size_t)h->U * h->K, nH = (size_t)h->I * h->K;
    if (h->dtype == CYMF_F32) { CYMF_TRY(download_f64(h->f32.W, W, nW, h->stream)); CYMF_TRY(download_f64(h->f32.H, H, nH, h->stream)); }
    else { CYMF_TRY(download_f64(h->f64.W, W, nW, h->stream)); CYMF_TRY(download_f64(h->f64.H, H, nH, h->stream)); }
    return 0;
}

extern "C" int cymf_wmf_half_sweep(cymf_wmf *h, int side) {
    if (!h || (side != 0 && side != 1)) return fail(CYMF_ERR_INVALID, "cymf_wmf_half_sweep: bad arguments");
    if (!h->have_data || !h->have_params) return fail(CYMF_ERR_INVALID, "cymf_wmf_half_sweep before set_data/upload");
    CYMF_TRY(use_device(h->device));
    if (h->dtype == CYMF_F32) CYMF_TRY(wmf_half<float>(h, h->f32, side)); else CYMF_TRY(wmf_half<double>(h, h->f64, side));
    return 0;
}

extern "C" int cymf_wmf_epochs(cymf_wmf *h, int32_t n_epochs) {
    if (!h || n_epochs < 0) return fail(CYMF_ERR_INVALID, "cymf_wmf_epochs: bad arguments");
    for (int32_t e = 0; e < n_epochs; ++e) {   // wmf.pyx:110-112
        CYMF_TRY(cymf_wmf_half_sweep(h, 0));
        CYMF_TRY(cymf_wmf_half_sweep(h, 1));
    }
    CYMF_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cymf_wmf_destroy(cymf_wmf *h) {
    if (!h) return 0;
    if (!cymf::runtime_alive(h->device)) return 0;   // process exit / runtime already torn down: leak quietly
    if (h->seg_stream) { (void)hipStreamSynchronize(h->seg_stream); (void)hipStreamDestroy(h->seg_stream); }
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
    if (h->ev_segs) (void)hipEventDestroy(h->ev_segs);
    delete h;
    return 0;
}
