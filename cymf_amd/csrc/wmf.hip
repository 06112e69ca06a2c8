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

template <typename T>
__global__ void wmf_add_diag_kernel(T *G, int K, T lambda) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) G[k * K + k] += lambda;
}

__device__ __forceinline__ float dsqrt(float x) { return __fsqrt_rn(x); }
__device__ __forceinline__ double dsqrt(double x) { return sqrt(x); }

// In-LDS Cholesky A = L L^T (lower, in place) and solve of A x = b; A is [K][lda].
template <typename T>
__device__ void chol_solve_lds(T *A, T *b, int K, int lda) {
    const int tid = threadIdx.x;
    for (int c = 0; c < K; ++c) {
        if (tid == 0) A[c * lda + c] = dsqrt(A[c * lda + c]);
        __syncthreads();
        const T inv = (T)1 / A[c * lda + c];
        for (int r = c + 1 + tid; r < K; r += WMF_THREADS) A[r * lda + c] *= inv;
        __syncthreads();
        // trailing update of the lower triangle: A[r][q] -= A[r][c] A[q][c], c < q <= r < K
        const int m = K - c - 1;
        for (int e = tid; e < m * m; e += WMF_THREADS) {
            const int r = c + 1 + e / m, q = c + 1 + e % m;
            if (q <= r) A[r * lda + q] -= A[r * lda + c] * A[q * lda + c];
        }
        __syncthreads();
    }
    // L z = b
    for (int c = 0; c < K; ++c) {
        if (tid == 0) b[c] /= A[c * lda + c];
        __syncthreads();
        const T bc = b[c];
        for (int r = c + 1 + tid; r < K; r += WMF_THREADS) b[r] -= A[r * lda + c] * bc;
        __syncthreads();
    }
    // L^T x = z
    for (int c = K - 1; c >= 0; --c) {
        if (tid == 0) b[c] /= A[c * lda + c];
        __syncthreads();
        const T bc = b[c];
        for (int r = tid; r < c; r += WMF_THREADS) b[r] -= A[c * lda + r] * bc;
        __syncthreads();
    }
}

// Generic row kernel (any K <= 128, f32 or f64).
template <typename T>
__global__ __launch_bounds__(WMF_THREADS) void wmf_row_kernel(int32_t rows, int K, const int32_t *__restrict__ indptr,
                                                             const int32_t *__restrict__ indices,
                                                             T *__restrict__ X, const T *__restrict__ Y,
                                                             const T *__restrict__ A0, T weight) {
    extern __shared__ unsigned char smem_raw[];
    const int lda = K + 1;
    T *A = reinterpret_cast<T *>(smem_raw);   // [K][K+1]
    T *b = A + K * lda;                        // [K]
    T *tile = b + K;                           // [WMF_TILE][K]
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
        chol_solve_lds<T>(A, b, K, lda);                        // wmf.pyx:168
        for (int k = tid; k < K; k += WMF_THREADS) X[(int64_t)i * K + k] = b[k];   // wmf.pyx:170-171
    }
}

// f32 MFMA row kernel, K = 32*T32.  Wave w owns upper-triangular 32x32 tiles t = w, w+4, ...
// of G = sum_j y_j y_j^T.  v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; here k indexes the gathered row (two per instruction), so both operands
// are read directly from the gathered rows: lane l loads y[row_{2s + (l>>5)}][32*tile + (l&31)].
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int T32>
__global__ __launch_bounds__(WMF_THREADS) void wmf_row_mfma_kernel(int32_t rows, const int32_t *__restrict__ indptr,
                                                                  const int32_t *__restrict__ indices,
                                                                  float *__restrict__ X, const float *__restrict__ Y,
                                                                  const float *__restrict__ A0, float weight) {
    constexpr int K = 32 * T32;
    constexpr int NT = T32 * (T32 + 1) / 2;          // upper-triangular tiles
    constexpr int TPW = (NT + 3) / 4;                // tiles per wave
    constexpr int lda = K + 1;
    extern __shared__ unsigned char smem_raw[];
    float *A = reinterpret_cast<float *>(smem_raw);  // [K][K+1]
    float *b = A + K * lda;                           // [K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    for (int32_t i = blockIdx.x; i < rows; i += gridDim.x) {
        const int32_t p0 = indptr[i], p1 = indptr[i + 1];
        if (p0 == p1) {
            for (int k = tid; k < K; k += WMF_THREADS) X[(int64_t)i * K + k] = 0;
            continue;
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
        chol_solve_lds<float>(A, b, K, lda);
        for (int k = tid; k < K; k += WMF_THREADS) X[(int64_t)i * K + k] = b[k];
    }
}

}  // namespace
}  // namespace cymf

using namespace cymf;

template <typename T>
struct WmfStore {
    DevBuf<T> W, H, G;
};

struct cymf_wmf {
    int32_t U = 0, I = 0, K = 0;
    int dtype = 0, device = 0;
    double weight = 10.0, wd = 0.01;
    hipStream_t stream = nullptr;
    WmfStore<float> f32;
    WmfStore<double> f64;
    DevBuf<int32_t> d_indptr, d_indices, d_tindptr, d_tindices;
    bool have_data = false, have_params = false;
    bool use_mfma = true;
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
    CYMF_TRY(st.G.alloc((size_t)K * K));
    CYMF_TRY(st.G.zero(h->stream));
    {   // YtY + lambda I  (wmf.pyx:142-143)
        int grid = (int)std::min<int64_t>(((int64_t)cols + WMF_TILE - 1) / WMF_TILE, 1024);
        size_t smem = sizeof(T) * WMF_TILE * K;
        hipLaunchKernelGGL(wmf_gram_kernel<T>, dim3(grid), dim3(WMF_THREADS), smem, h->stream, Y, (int64_t)cols, K, st.G.p);
        hipLaunchKernelGGL(wmf_add_diag_kernel<T>, dim3((K + 63) / 64), dim3(64), 0, h->stream, st.G.p, K, (T)h->wd);
        CYMF_HIP(hipGetLastError());
    }
    const int grid = (int)std::min<int64_t>(rows, 256 * 16);
    bool mfma = false;
    if constexpr (sizeof(T) == 4) {
        if (h->use_mfma && K % 32 == 0 && K <= 128) {
            mfma = true;
            size_t smem = sizeof(float) * ((size_t)K * (K + 1) + K);
            const float *Yf = reinterpret_cast<const float *>(Y);
            float *Xf = reinterpret_cast<float *>(X);
            const float *Gf = reinterpret_cast<const float *>(st.G.p);
            CYMF_TRY(allow_lds(wmf_row_mfma_kernel<1>, smem)); CYMF_TRY(allow_lds(wmf_row_mfma_kernel<2>, smem));
            CYMF_TRY(allow_lds(wmf_row_mfma_kernel<3>, smem)); CYMF_TRY(allow_lds(wmf_row_mfma_kernel<4>, smem));
            switch (K / 32) {
            case 1: hipLaunchKernelGGL(wmf_row_mfma_kernel<1>, dim3(grid), dim3(WMF_THREADS), smem, h->stream, rows, ip, ix, Xf, Yf, Gf, (float)h->weight); break;
            case 2: hipLaunchKernelGGL(wmf_row_mfma_kernel<2>, dim3(grid), dim3(WMF_THREADS), smem, h->stream, rows, ip, ix, Xf, Yf, Gf, (float)h->weight); break;
            case 3: hipLaunchKernelGGL(wmf_row_mfma_kernel<3>, dim3(grid), dim3(WMF_THREADS), smem, h->stream, rows, ip, ix, Xf, Yf, Gf, (float)h->weight); break;
            default: hipLaunchKernelGGL(wmf_row_mfma_kernel<4>, dim3(grid), dim3(WMF_THREADS), smem, h->stream, rows, ip, ix, Xf, Yf, Gf, (float)h->weight); break;
            }
        }
    }
    if (!mfma) {
        size_t smem = sizeof(T) * ((size_t)K * (K + 1) + K + (size_t)WMF_TILE * K);
        CYMF_TRY(allow_lds(wmf_row_kernel<T>, smem));
        hipLaunchKernelGGL(wmf_row_kernel<T>, dim3(grid), dim3(WMF_THREADS), smem, h->stream, rows, K, ip, ix, X, Y, st.G.p, (T)h->weight);
    }
    CYMF_HIP(hipGetLastError());
    return 0;
}

extern "C" int cymf_wmf_create(cymf_wmf **out, int32_t U, int32_t I, int32_t K, double weight, double weight_decay,
                               int dtype, int device) {
    if (!out) return fail(CYMF_ERR_INVALID, "cymf_wmf_create: out is NULL");
    *out = nullptr;
    if (U <= 0 || I <= 0 || K <= 0) return fail(CYMF_ERR_INVALID, "cymf_wmf_create: U, I, K must be positive");
    if (K > 128) return fail(CYMF_ERR_UNSUPPORTED, "cymf_wmf_create: K=%d > 128 (the K x K system is LDS-resident)", K);
    if (dtype != CYMF_F32 && dtype != CYMF_F64) return fail(CYMF_ERR_INVALID, "cymf_wmf_create: dtype %d", dtype);
    CYMF_TRY(use_device(device));
    cymf_wmf *h = new cymf_wmf();
    h->U = U; h->I = I; h->K = K; h->weight = weight; h->wd = weight_decay; h->dtype = dtype; h->device = device;
    const char *env = getenv("CYMF_WMF_NO_MFMA");
    h->use_mfma = !(env && env[0] == '1');
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(CYMF_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    *out = h;
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
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
    return 0;
}
