// expomf.hip -- Exposure MF (Liang et al. 2016) as cymf trains it (cymf/expomf.pyx:105-207), float64.
//
// One epoch (expomf.pyx:132-147):
//   E-step   : n_ui = sqrt(lam_y / 2.0 * pi) * exp(-lam_y (w_u . h_i)^2 / 2)   [the reference's operator order: sqrt(lam_y*pi/2)]
//              E_ui = (n_ui + 1e-8) / (n_ui + 1e-8 + (1 - mu_i) / mu_i);  E_ui = 1 where X_ui != 0          (:141-144)
//   users    : w_u = (wd/lam_y I + lam_y sum_j E_uj h_j h_j^T)^-1 (lam_y sum_{j in X_u} E_uj h_j); 0 for users without
//              positives -- the sum in the matrix runs over ALL items, weighted by the dense exposure row  (:163-204)
//   items    : the same with E^T and the updated W                                                        (:147)
//   mu_i     = (alpha_1 + sum_u E_ui - 1) / (alpha_1 + alpha_2 + U - 2),  alpha_1 = alpha_2 = 1            (:149)
// The dense U x I exposure makes the sweeps O(U I K^2): the reference runs this on ml-100k-sized data and so does
// this build -- one workgroup per row, the weighted Gramian accumulated from LDS tiles of Y, an in-LDS Cholesky in
// place of dgesv (same SPD system).  SURVEY.md 8f-4; parity is unpinned (expomf.pyx needs cblas to build here).
#include "store.h"

#include <algorithm>
#include <cmath>

namespace cymf {
namespace {

constexpr int EXPO_THREADS = 256;
constexpr int EXPO_TILE = 16;

__global__ __launch_bounds__(256) void expo_estep_kernel(const double *__restrict__ W, const double *__restrict__ H,
                                                        const double *__restrict__ mu, double *__restrict__ E, int32_t U, int32_t I,
                                                        int K, double lam_y, double coef) {
    const int64_t n = (int64_t)U * I;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const int64_t u = t / I, i = t - u * I;
        double d = 0.0;
        for (int k = 0; k < K; ++k) d += W[u * K + k] * H[i * K + k];
        const double nui = coef * exp(-lam_y * (d * d) / 2.0);
        const double m = mu[i];
        E[t] = (nui + 1e-8) / (nui + 1e-8 + (1.0 - m) / m);
    }
}

__global__ void expo_mark_kernel(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, double *__restrict__ E,
                                 int32_t U, int32_t I) {
    const int lane = lane_id();
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave0; u < U; u += n_waves)
        for (int32_t p = indptr[u] + lane; p < indptr[u + 1]; p += 64) E[u * I + indices[p]] = 1.0;
}

__global__ __launch_bounds__(256) void expo_mu_kernel(const double *__restrict__ E, double *__restrict__ mu, int32_t U, int32_t I,
                                                     double alpha1, double alpha2) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    double s = 0.0;
    for (int32_t u = 0; u < U; ++u) s += E[(int64_t)u * I + i];     // coalesced across the threads of a workgroup
    mu[i] = (alpha1 + s - 1.0) / (alpha1 + alpha2 + (double)U - 2.0);
}

// X[i] <- solve for row i.  The exposure of (row i, column j) is E[i * es_row + j * es_col] (users: es_row = I, es_col = 1;
// items: es_row = 1, es_col = I).
__global__ __launch_bounds__(EXPO_THREADS) void expo_row_kernel(int32_t rows, int32_t cols, int K, const int32_t *__restrict__ indptr,
                                                               const int32_t *__restrict__ indices, const double *__restrict__ E,
                                                               int64_t es_row, int64_t es_col, double *__restrict__ X,
                                                               const double *__restrict__ Y, double lam_y, double ridge,
                                                               double *A_scratch) {
    extern __shared__ unsigned char smem_raw[];
    const int lda = K + 1;
    // the K x K system sits in LDS while it fits; for larger K (cymf/expomf.pyx:45 takes any num_components) in the
    // workgroup's own slice of a global scratch buffer -- the workgroup barriers order both alike
    double *lds = reinterpret_cast<double *>(smem_raw);
    double *A = A_scratch ? A_scratch + (size_t)blockIdx.x * K * lda : lds;   // [K][K+1]
    double *b = A_scratch ? lds : lds + K * lda;         // [K]
    double *tile = b + K;                                // [EXPO_TILE][K]
    double *ew = tile + EXPO_TILE * K;                   // [EXPO_TILE]
    const int tid = threadIdx.x;
    const int KK = K * K;
    for (int32_t i = blockIdx.x; i < rows; i += gridDim.x) {
        const int32_t p0 = indptr[i], p1 = indptr[i + 1];
        if (p0 == p1) {                                            // expomf.pyx:178-182
            for (int k = tid; k < K; k += EXPO_THREADS) X[(int64_t)i * K + k] = 0.0;
            continue;
        }
        __syncthreads();
        for (int e = tid; e < KK; e += EXPO_THREADS) A[(e / K) * lda + (e % K)] = (e / K == e % K) ? ridge : 0.0;   // (wd / lam_y) I, :171
        for (int k = tid; k < K; k += EXPO_THREADS) b[k] = 0.0;
        __syncthreads();
        // b = lam_y sum_{j in X_i} E_ij y_j                                                                          (:187-190)
        for (int k = tid; k < K; k += EXPO_THREADS) {
            double s = 0.0;
            for (int32_t p = p0; p < p1; ++p) {
                const int32_t j = indices[p];
                s += Y[(int64_t)j * K + k] * E[(int64_t)i * es_row + (int64_t)j * es_col] * lam_y;
            }
            b[k] = s;
        }
        // A += lam_y sum_j E_ij y_j y_j^T over ALL columns j                                                          (:192-196)
        for (int32_t j0 = 0; j0 < cols; j0 += EXPO_TILE) {
            const int nr = cols - j0 < EXPO_TILE ? cols - j0 : EXPO_TILE;
            __syncthreads();
            for (int e = tid; e < nr * K; e += EXPO_THREADS) tile[e] = Y[(int64_t)j0 * K + e];
            for (int r = tid; r < nr; r += EXPO_THREADS) ew[r] = E[(int64_t)i * es_row + (int64_t)(j0 + r) * es_col] * lam_y;
            __syncthreads();
            for (int e = tid; e < KK; e += EXPO_THREADS) {
                const int k = e / K, k2 = e - k * K;
                double s = 0.0;
                for (int r = 0; r < nr; ++r) s += tile[r * K + k] * tile[r * K + k2] * ew[r];
                A[k * lda + k2] += s;
            }
        }
        __syncthreads();
        // Cholesky A = L L^T in place (lower), then the two triangular solves; the system is SPD (ridge > 0)
        for (int c = 0; c < K; ++c) {
            __syncthreads();
            const double dcc = sqrt(A[c * lda + c]);
            __syncthreads();
            if (tid == 0) A[c * lda + c] = dcc;
            for (int r = c + 1 + tid; r < K; r += EXPO_THREADS) A[r * lda + c] /= dcc;
            __syncthreads();
            for (int e = tid; e < (K - c - 1) * (K - c - 1); e += EXPO_THREADS) {
                const int r = c + 1 + e / (K - c - 1), q = c + 1 + e % (K - c - 1);
                if (q <= r) A[r * lda + q] -= A[r * lda + c] * A[q * lda + c];
            }
        }
        __syncthreads();
        if (tid == 0) {   // the O(K^2) substitutions on one thread are noise next to the O(cols K^2) Gramian
            for (int c = 0; c < K; ++c) {
                double s = b[c];
                for (int r = 0; r < c; ++r) s -= A[c * lda + r] * b[r];
                b[c] = s / A[c * lda + c];
            }
            for (int c = K - 1; c >= 0; --c) {
                double s = b[c];
                for (int r = c + 1; r < K; ++r) s -= A[r * lda + c] * b[r];
                b[c] = s / A[c * lda + c];
            }
        }
        __syncthreads();
        for (int k = tid; k < K; k += EXPO_THREADS) X[(int64_t)i * K + k] = b[k];     // :202-203
    }
}

inline int ew_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace
}  // namespace cymf

using namespace cymf;

struct cymf_expomf {
    int32_t U = 0, I = 0, K = 0;
    int device = 0;
    double lam_y = 1.0, wd = 0.01;
    hipStream_t stream = nullptr;
    DevBuf<double> W, H, E, mu;
    DevBuf<double> A_wide;   // K > ~130: per-workgroup systems of expo_row_kernel (workgroup-private scratch)
    DevBuf<int32_t> d_indptr, d_indices, d_tindptr, d_tindices;
    bool have_data = false, have_params = false;
};

extern "C" int cymf_expomf_create(cymf_expomf **out, int32_t U, int32_t I, int32_t K, double lam_y, double weight_decay, int device) {
    if (!out) return fail(CYMF_ERR_INVALID, "cymf_expomf_create: out is NULL");
    *out = nullptr;
    if (U <= 0 || I <= 0 || K <= 0) return fail(CYMF_ERR_INVALID, "cymf_expomf_create: U, I, K must be positive");
    if ((int64_t)K * K > ((int64_t)1 << 26)) return fail(CYMF_ERR_UNSUPPORTED, "cymf_expomf_create: K=%d: a K x K system per workgroup beyond 2^26 entries", K);
    if (!(lam_y > 0) || !(weight_decay > 0)) return fail(CYMF_ERR_INVALID, "cymf_expomf_create: lam_y and weight_decay must be positive");
    if ((uint64_t)U * (uint64_t)I > (1ull << 32)) return fail(CYMF_ERR_UNSUPPORTED, "cymf_expomf_create: the dense %d x %d exposure matrix is not a realistic input", U, I);
    CYMF_TRY(use_device(device));
    cymf_expomf *h = new cymf_expomf();
    h->U = U; h->I = I; h->K = K; h->lam_y = lam_y; h->wd = weight_decay; h->device = device;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(CYMF_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    *out = h;
    return 0;
}

extern "C" int cymf_expomf_set_data(cymf_expomf *h, const int32_t *indptr, const int32_t *indices, const int32_t *t_indptr,
                                    const int32_t *t_indices) {
    if (!h || !indptr || !t_indptr) return fail(CYMF_ERR_INVALID, "cymf_expomf_set_data: bad arguments");
    CYMF_TRY(use_device(h->device));
    const int64_t nnz = indptr[h->U];
    if (indptr[0] != 0 || t_indptr[0] != 0 || nnz != t_indptr[h->I] || (nnz > 0 && (!indices || !t_indices)))
        return fail(CYMF_ERR_INVALID, "cymf_expomf_set_data: CSR / transposed CSR disagree");
    for (int32_t u = 0; u < h->U; ++u) if (indptr[u] > indptr[u + 1]) return fail(CYMF_ERR_INVALID, "cymf_expomf_set_data: indptr not monotone");
    for (int32_t i = 0; i < h->I; ++i) if (t_indptr[i] > t_indptr[i + 1]) return fail(CYMF_ERR_INVALID, "cymf_expomf_set_data: transposed indptr not monotone");
    for (int64_t p = 0; p < nnz; ++p)
        if (indices[p] < 0 || indices[p] >= h->I || t_indices[p] < 0 || t_indices[p] >= h->U) return fail(CYMF_ERR_INVALID, "cymf_expomf_set_data: index out of range");
    CYMF_TRY(h->d_indptr.upload(indptr, (size_t)h->U + 1, h->stream));
    CYMF_TRY(h->d_indices.upload(indices, (size_t)nnz, h->stream));
    CYMF_TRY(h->d_tindptr.upload(t_indptr, (size_t)h->I + 1, h->stream));
    CYMF_TRY(h->d_tindices.upload(t_indices, (size_t)nnz, h->stream));
    CYMF_TRY(h->E.alloc((size_t)h->U * h->I));
    std::vector<double> mu0((size_t)h->I, 0.01);                    // expomf.pyx:120
    CYMF_TRY(h->mu.upload(mu0.data(), mu0.size(), h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    h->have_data = true;
    return 0;
}

extern "C" int cymf_expomf_upload(cymf_expomf *h, const double *W, const double *H) {
    if (!h || !W || !H) return fail(CYMF_ERR_INVALID, "cymf_expomf_upload: bad arguments");
    CYMF_TRY(use_device(h->device));
    CYMF_TRY(upload_f64(h->W, W, (size_t)h->U * h->K, h->stream));
    CYMF_TRY(upload_f64(h->H, H, (size_t)h->I * h->K, h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    h->have_params = true;
    return 0;
}

extern "C" int cymf_expomf_download(cymf_expomf *h, double *W, double *H) {
    if (!h || !W || !H || !h->have_params) return fail(CYMF_ERR_INVALID, "cymf_expomf_download: bad arguments / no params");
    CYMF_TRY(use_device(h->device));
    CYMF_TRY(download_f64(h->W, W, (size_t)h->U * h->K, h->stream));
    CYMF_TRY(download_f64(h->H, H, (size_t)h->I * h->K, h->stream));
    return 0;
}

extern "C" int cymf_expomf_epochs(cymf_expomf *h, int32_t n_epochs) {
    if (!h || n_epochs < 0) return fail(CYMF_ERR_INVALID, "cymf_expomf_epochs: bad arguments");
    if (!h->have_data || !h->have_params) return fail(CYMF_ERR_INVALID, "cymf_expomf_epochs before set_data/upload");
    CYMF_TRY(use_device(h->device));
    const int K = h->K;
    const double coef = std::sqrt(h->lam_y / 2.0 * M_PI);          // expomf.pyx:141, evaluated left to right
    size_t smem = sizeof(double) * ((size_t)K * (K + 1) + K + (size_t)EXPO_TILE * K + EXPO_TILE);
    double *scratch = nullptr;
    const int grid_u = std::min(h->U, smem > 160 * 1024 ? 1024 : 4096), grid_i = std::min(h->I, smem > 160 * 1024 ? 1024 : 4096);
    if (smem > 160 * 1024) {   // does not fit the CU's LDS: global slices
        CYMF_TRY(h->A_wide.alloc((size_t)std::max(grid_u, grid_i) * K * (K + 1)));
        scratch = h->A_wide.p;
        smem = sizeof(double) * ((size_t)K + (size_t)EXPO_TILE * K + EXPO_TILE);
    }
    if (smem > 48 * 1024)
        CYMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(expo_row_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int32_t e = 0; e < n_epochs; ++e) {
        hipLaunchKernelGGL(expo_estep_kernel, dim3(ew_blocks((int64_t)h->U * h->I)), dim3(256), 0, h->stream, h->W.p, h->H.p, h->mu.p, h->E.p,
                           h->U, h->I, K, h->lam_y, coef);
        hipLaunchKernelGGL(expo_mark_kernel, dim3(std::max(1, std::min(h->U / 4 + 1, 1024))), dim3(256), 0, h->stream, h->d_indptr.p,
                           h->d_indices.p, h->E.p, h->U, h->I);
        hipLaunchKernelGGL(expo_row_kernel, dim3(grid_u), dim3(EXPO_THREADS), smem, h->stream, h->U, h->I, K, h->d_indptr.p,
                           h->d_indices.p, h->E.p, (int64_t)h->I, (int64_t)1, h->W.p, h->H.p, h->lam_y, h->wd / h->lam_y, scratch);
        hipLaunchKernelGGL(expo_row_kernel, dim3(grid_i), dim3(EXPO_THREADS), smem, h->stream, h->I, h->U, K, h->d_tindptr.p,
                           h->d_tindices.p, h->E.p, (int64_t)1, (int64_t)h->I, h->H.p, h->W.p, h->lam_y, h->wd / h->lam_y, scratch);
        hipLaunchKernelGGL(expo_mu_kernel, dim3((h->I + 255) / 256), dim3(256), 0, h->stream, h->E.p, h->mu.p, h->U, h->I, 1.0, 1.0);
        CYMF_HIP(hipGetLastError());
    }
    CYMF_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cymf_expomf_destroy(cymf_expomf *h) {
    if (!h) return 0;
    if (!cymf::runtime_alive(h->device)) return 0;   // process exit / runtime already torn down: leak quietly
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
    return 0;
}
