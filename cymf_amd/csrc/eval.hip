// eval.hip -- the sampled-negative ranking harness on the device.
//
// Replaces the per-user Python loop of Evaluator.evaluate (cymf/evaluator.pyx:57-139):
//   candidates : the user's held-out items (feedback 1) followed by num_negatives draws of
//                UniformGenerator(0, I, seed) with redraw while the item is a known positive
//                (:80-88).  The redraws make the stream position of user u depend on all users
//                before it, so ONE wavefront walks the stream, testing 64 draws per round
//                (eval_walk_kernel); the result is cached per (seed, num_negatives) because the
//                reference re-creates the same generator on every call (:82).
//   scores     : np.dot(H[items], W[user]) (:90) -- one lane per candidate, fp64, k ascending.
//   ranking    : argsort()[::-1] (:90) is only consumed through its first k entries
//                (cymf/metrics.pyx:24-147), so the kernel selects the top max(k) by repeated
//                wave-wide argmax; equal scores take the larger candidate position first (what a
//                reversed stable sort gives; ties only occur between repeated negatives).
//   metrics    : DCG / Recall / MAP @k and their IPS variants, which index the propensities by
//                candidate POSITION, not item id (evaluator.pyx:92) -- kept.
#include "common.h"
#include "rng.h"

#include <algorithm>
#include <cstdlib>

namespace cymf {

struct EvalWalk {
    int32_t eu;       // next evaluated user (index into eval_users)
    int32_t acc;      // negatives already accepted for that user
    int64_t used;     // draws consumed over all launches (diagnostic)
};

// One wavefront.  Round: lanes test draws[pos + lane] against the current user's positives
// (binary search in the sorted CSR row), the accepted ones are appended in stream order until the
// user has num_neg; the draw after the last accepted one is the next user's first.
__global__ void __launch_bounds__(64) eval_walk_kernel(const uint32_t *__restrict__ draws, int64_t n_draws,
                                                      const int32_t *__restrict__ eval_users, int32_t n_eval,
                                                      const int32_t *__restrict__ all_indptr,
                                                      const int32_t *__restrict__ all_indices, int32_t num_neg,
                                                      int32_t *__restrict__ neg_out, EvalWalk *st) {
    const int lane = lane_id();
    int32_t eu = st->eu, acc = st->acc;
    int64_t pos = 0;
    while (eu < n_eval && pos < n_draws) {
        const int32_t u = eval_users[eu];
        const int32_t lo0 = all_indptr[u], hi0 = all_indptr[u + 1];
        const bool valid = pos + lane < n_draws;
        const int32_t d = valid ? (int32_t)draws[pos + lane] : 0;
        int32_t lo = lo0, hi = hi0;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (all_indices[mid] < d) lo = mid + 1; else hi = mid;
        }
        const bool ok = valid && !(lo < hi0 && all_indices[lo] == d);
        const uint64_t mask = __ballot(ok);
        const int32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
        const int32_t need = num_neg - acc;
        const int32_t cnt = __popcll(mask);
        if (cnt >= need) {
            const uint64_t last = __ballot(ok && rank == need - 1);
            const int32_t L = __ffsll((unsigned long long)last) - 1;
            if (ok && lane <= L) neg_out[(int64_t)eu * num_neg + acc + rank] = d;
            pos += L + 1;
            eu += 1;
            acc = 0;
        } else {
            if (ok) neg_out[(int64_t)eu * num_neg + acc + rank] = d;
            acc += cnt;
            const int64_t left = n_draws - pos;
            pos += left < 64 ? left : 64;
        }
    }
    if (lane == 0) { st->eu = eu; st->acc = acc; st->used += pos; }
}

// candidate c of evaluated user eu: held-out items first, then the sampled negatives
__device__ __forceinline__ int32_t candidate(int32_t c, int32_t n_test, const int32_t *test_row, const int32_t *neg_row) {
    return c < n_test ? test_row[c] : neg_row[c - n_test];
}

__global__ void __launch_bounds__(256) eval_score_kernel(const double *__restrict__ W, const double *__restrict__ H, int32_t K,
                                                        const int32_t *__restrict__ eval_users, int32_t n_eval,
                                                        const int64_t *__restrict__ cand_off,
                                                        const int32_t *__restrict__ test_indptr,
                                                        const int32_t *__restrict__ test_indices,
                                                        const int32_t *__restrict__ neg, int32_t num_neg,
                                                        double *__restrict__ scores) {
    const int lane = lane_id();
    const int32_t eu = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (eu >= n_eval) return;
    const int32_t u = eval_users[eu];
    const int32_t t0 = test_indptr[u], n_test = test_indptr[u + 1] - t0;
    const int32_t n = n_test + num_neg;
    const double *w = W + (size_t)u * K;
    for (int32_t c = lane; c < n; c += 64) {
        const int32_t item = candidate(c, n_test, test_indices + t0, neg + (int64_t)eu * num_neg);
        const double *h = H + (size_t)item * K;
        double s = 0.0;
        for (int32_t k = 0; k < K; ++k) s += h[k] * w[k];
        scores[cand_off[eu] + c] = s;
    }
}

__device__ __forceinline__ bool ranks_before(double s, int32_t i, double t, int32_t j) {   // (s,i) sorts ahead of (t,j)
    return s > t || (s == t && i > j);
}

// out layout: [metric m (DCG, Recall, MAP)][ki][U]
// WIDE = false: k <= 64, lane r keeps the candidate ranked r (four users per workgroup).  WIDE = true: any k
// (cymf/evaluator.pyx:29 takes any list of k): the ranked list lives in LDS, one user (one wavefront) per workgroup.
template <bool WIDE>
__global__ void __launch_bounds__(256) eval_rank_kernel(const double *__restrict__ scores, const int32_t *__restrict__ eval_users,
                                                       int32_t n_eval, const int64_t *__restrict__ cand_off,
                                                       const int32_t *__restrict__ test_indptr, int32_t num_neg,
                                                       const double *__restrict__ prop, int32_t n_prop, int unbiased,
                                                       const int32_t *__restrict__ ks, int32_t nk, int32_t kmax,
                                                       const double *__restrict__ disc, int32_t U, double *__restrict__ out) {
    extern __shared__ int32_t s_rank[];      // WIDE: candidate position by rank
    const int lane = lane_id();
    const int32_t eu = __builtin_amdgcn_readfirstlane((int)(WIDE ? blockIdx.x : blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (eu >= n_eval) return;
    const int32_t u = eval_users[eu];
    const int32_t n_test = test_indptr[u + 1] - test_indptr[u];
    const int32_t n = n_test + num_neg;
    const double *sc = scores + cand_off[eu];
    const int32_t rounds = kmax < n ? kmax : n;

    double prev_s = __builtin_inf();
    int32_t prev_i = 0x7fffffff;
    int32_t mine = -1;                       // lane r keeps the candidate position ranked r
    int32_t ranked = 0;                      // WIDE: ranks filled in s_rank
    for (int32_t r = 0; r < rounds; ++r) {
        double bs = 0.0;
        int32_t bi = -1;
        for (int32_t c = lane; c < n; c += 64) {
            const double s = sc[c];
            if (ranks_before(prev_s, prev_i, s, c) && (bi < 0 || ranks_before(s, c, bs, bi))) { bs = s; bi = c; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double os = __shfl_xor(bs, off, 64);
            const int32_t oi = __shfl_xor(bi, off, 64);
            if (oi >= 0 && (bi < 0 || ranks_before(os, oi, bs, bi))) { bs = os; bi = oi; }
        }
        if (bi < 0) break;                   // only NaN scores left: they never rank
        if constexpr (WIDE) { if (lane == 0) s_rank[r] = bi; }
        else if (lane == r) mine = bi;
        prev_s = bs;
        prev_i = bi;
        if constexpr (WIDE) ranked = r + 1;
    }
    const double y = (mine >= 0 && mine < n_test) ? 1.0 : 0.0;
    double p = 1.0;
    if (unbiased && mine >= 0) p = prop[mine < n_prop ? mine : n_prop - 1];
    // normaliser: sum over the whole ranked list of y (or y/p) = over the held-out items
    double norm = 0.0;
    if (unbiased) {
        for (int32_t c = lane; c < n_test; c += 64) norm += 1.0 / prop[c < n_prop ? c : n_prop - 1];
        norm = wave_sum(norm);
    } else {
        norm = (double)n_test;
    }
    const double yp = y / p;
    for (int32_t ki = 0; ki < nk; ++ki) {
        const int32_t k = ks[ki] < rounds ? ks[ki] : rounds;
        double dcg = 0.0, rec = 0.0, ap = 0.0, run = 0.0;
        for (int32_t r = 0; r < k; ++r) {
            double yr, hit;
            if constexpr (WIDE) {
                const int32_t pos = r < ranked ? s_rank[r] : -1;
                hit = (pos >= 0 && pos < n_test) ? 1.0 : 0.0;
                yr = (unbiased && pos >= 0) ? hit / prop[pos < n_prop ? pos : n_prop - 1] : hit;
            } else {
                yr = __shfl(yp, r, 64);
                hit = __shfl(y, r, 64);
            }
            run += yr;                                   // cumsum(y) or cumsum(y/p)
            dcg += yr / disc[r];
            rec += yr;
            if (hit == 1.0) ap += run / (double)(r + 1);
        }
        if (lane == 0) {
            const bool none = norm == 0.0;
            out[((size_t)0 * nk + ki) * U + u] = none ? 0.0 : dcg / norm;
            out[((size_t)1 * nk + ki) * U + u] = none ? 0.0 : rec / norm;
            out[((size_t)2 * nk + ki) * U + u] = none ? 0.0 : ap / norm;
        }
    }
}

}  // namespace cymf

using namespace cymf;

struct cymf_eval {
    int32_t U = 0, I = 0, device = 0;
    hipStream_t stream = nullptr;
    std::vector<int32_t> h_test_indptr, h_eval_users;
    DevBuf<int32_t> d_test_indptr, d_test_indices, d_all_indptr, d_all_indices, d_eval_users;
    DevBuf<int64_t> d_cand_off;
    DevBuf<double> d_prop;
    int32_t n_prop = 0;
    // negatives of the last (seed, num_negatives)
    bool have_neg = false;
    uint32_t neg_seed = 0;
    int32_t neg_num = 0;
    int64_t draws_used = 0;
    DevBuf<int32_t> d_neg;
    DevBuf<double> d_scores, d_W, d_H, d_out, d_disc;
    DevBuf<int32_t> d_ks;
};

static int eval_check_csr(const int32_t *indptr, const int32_t *indices, int32_t rows, int32_t cols, bool sorted, const char *what) {
    if (indptr[0] != 0) return fail(CYMF_ERR_INVALID, "cymf_eval_create: %s indptr[0] != 0", what);
    for (int32_t r = 0; r < rows; ++r) {
        if (indptr[r] > indptr[r + 1]) return fail(CYMF_ERR_INVALID, "cymf_eval_create: %s indptr not monotone", what);
        for (int32_t q = indptr[r]; q < indptr[r + 1]; ++q) {
            if (indices[q] < 0 || indices[q] >= cols) return fail(CYMF_ERR_INVALID, "cymf_eval_create: %s index out of range", what);
            if (sorted && q > indptr[r] && indices[q - 1] >= indices[q])
                return fail(CYMF_ERR_INVALID, "cymf_eval_create: %s rows must have sorted, unique indices", what);
        }
    }
    return 0;
}

extern "C" int cymf_eval_create(cymf_eval **out, int32_t U, int32_t I, const int32_t *test_indptr, const int32_t *test_indices,
                                const int32_t *all_indptr, const int32_t *all_indices, const double *propensity,
                                int32_t n_propensity, int device) {
    if (!out) return fail(CYMF_ERR_INVALID, "cymf_eval_create: out is NULL");
    *out = nullptr;
    if (U <= 0 || I <= 0 || !test_indptr || !all_indptr) return fail(CYMF_ERR_INVALID, "cymf_eval_create: bad arguments");
    if ((test_indptr[U] > 0 && !test_indices) || (all_indptr[U] > 0 && !all_indices)) return fail(CYMF_ERR_INVALID, "cymf_eval_create: NULL indices");
    if (propensity && n_propensity <= 0) return fail(CYMF_ERR_INVALID, "cymf_eval_create: empty propensity vector");
    CYMF_TRY(eval_check_csr(test_indptr, test_indices, U, I, false, "X"));
    CYMF_TRY(eval_check_csr(all_indptr, all_indices, U, I, true, "user_positives"));
    CYMF_TRY(use_device(device));
    cymf_eval *h = new cymf_eval();
    h->U = U; h->I = I; h->device = device;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(CYMF_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    h->h_test_indptr.assign(test_indptr, test_indptr + U + 1);
    std::vector<int64_t> off;
    for (int32_t u = 0; u < U; ++u)
        if (test_indptr[u + 1] > test_indptr[u]) {      // users without held-out items are skipped (:73-74)
            if (all_indptr[u + 1] - all_indptr[u] >= I) {
                delete h;
                return fail(CYMF_ERR_INVALID, "cymf_eval_create: user %d has every item as a positive; no negative can be drawn", u);
            }
            h->h_eval_users.push_back(u);
        }
    int rc = 0;
    auto up = [&](int r) { if (!rc) rc = r; };
    up(h->d_test_indptr.upload(test_indptr, (size_t)U + 1, h->stream));
    up(h->d_test_indices.upload(test_indices, (size_t)test_indptr[U], h->stream));
    up(h->d_all_indptr.upload(all_indptr, (size_t)U + 1, h->stream));
    up(h->d_all_indices.upload(all_indices, (size_t)all_indptr[U], h->stream));
    up(h->d_eval_users.upload(h->h_eval_users.data(), h->h_eval_users.size(), h->stream));
    if (propensity) { up(h->d_prop.upload(propensity, (size_t)n_propensity, h->stream)); h->n_prop = n_propensity; }
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(CYMF_ERR_HIP, "cymf_eval_create: upload failed");
    if (rc) { (void)hipStreamDestroy(h->stream); delete h; return rc; }
    *out = h;
    return 0;
}

static int eval_sample(cymf_eval *h, uint32_t seed, int32_t num_neg) {
    if (h->have_neg && h->neg_seed == seed && h->neg_num == num_neg) return 0;
    h->have_neg = false;
    const int32_t n_eval = (int32_t)h->h_eval_users.size();
    CYMF_TRY(h->d_neg.alloc((size_t)n_eval * num_neg));
    std::vector<int64_t> off((size_t)n_eval + 1, 0);
    for (int32_t e = 0; e < n_eval; ++e) {
        const int32_t u = h->h_eval_users[e];
        off[e + 1] = off[e] + (h->h_test_indptr[u + 1] - h->h_test_indptr[u]) + num_neg;
    }
    CYMF_TRY(h->d_cand_off.upload(off.data(), off.size(), h->stream));
    CYMF_TRY(h->d_scores.alloc((size_t)off[n_eval]));
    if (n_eval == 0 || num_neg == 0) { CYMF_HIP(hipStreamSynchronize(h->stream)); h->have_neg = true; h->neg_seed = seed; h->neg_num = num_neg; return 0; }

    DevBuf<EvalWalk> d_st;
    EvalWalk st{0, 0, 0};
    CYMF_TRY(d_st.upload(&st, 1, h->stream));
    const int64_t total = (int64_t)n_eval * num_neg;
    DeviceRng rng;
    CYMF_TRY(rng.init(seed, (uint64_t)h->I, h->stream, /*parallel=*/total >= (int64_t)4 << 20));
    DevBuf<uint32_t> d_draws;
    while (st.eu < n_eval) {
        // what is still needed plus room for the redraws; a short block only costs another round
        const int64_t need = (int64_t)(n_eval - st.eu) * num_neg - st.acc;
        int64_t block = std::min<int64_t>(need + need / 8 + 4096, (int64_t)1 << 28);
        if (const char *e = getenv("CYMF_EVAL_BLOCK")) block = std::max<int64_t>(1, atoll(e));   // tests: blocks that end mid-user
        CYMF_TRY(d_draws.alloc((size_t)block));
        CYMF_TRY(rng.generate(0, block, d_draws.p, h->stream));
        hipLaunchKernelGGL(eval_walk_kernel, dim3(1), dim3(64), 0, h->stream, d_draws.p, block, h->d_eval_users.p, n_eval,
                           h->d_all_indptr.p, h->d_all_indices.p, num_neg, h->d_neg.p, d_st.p);
        CYMF_HIP(hipGetLastError());
        CYMF_HIP(hipMemcpyAsync(&st, d_st.p, sizeof st, hipMemcpyDeviceToHost, h->stream));
        CYMF_HIP(hipStreamSynchronize(h->stream));
    }
    h->draws_used = st.used;
    h->have_neg = true; h->neg_seed = seed; h->neg_num = num_neg;
    return 0;
}

extern "C" int cymf_eval_negatives(cymf_eval *h, uint32_t seed, int32_t num_negatives, int32_t *users_out, int32_t *neg_out,
                                   int64_t *draws_used) {
    if (!h || num_negatives < 0) return fail(CYMF_ERR_INVALID, "cymf_eval_negatives: bad arguments");
    CYMF_TRY(use_device(h->device));
    CYMF_TRY(eval_sample(h, seed, num_negatives));
    const size_t n_eval = h->h_eval_users.size();
    if (users_out) memcpy(users_out, h->h_eval_users.data(), n_eval * sizeof(int32_t));
    if (neg_out && n_eval * num_negatives)
        CYMF_HIP(hipMemcpy(neg_out, h->d_neg.p, n_eval * num_negatives * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (draws_used) *draws_used = h->draws_used;
    return 0;
}

extern "C" int cymf_eval_num_users(cymf_eval *h, int32_t *n_eval) {
    if (!h || !n_eval) return fail(CYMF_ERR_INVALID, "cymf_eval_num_users: bad arguments");
    *n_eval = (int32_t)h->h_eval_users.size();
    return 0;
}

extern "C" int cymf_eval_run(cymf_eval *h, const double *W, const double *H, int32_t K, uint32_t seed, int32_t num_negatives,
                             const int32_t *ks, int32_t nk, const double *discounts, int unbiased, double *out) {
    if (!h || !W || !H || K <= 0 || num_negatives < 0 || !ks || nk <= 0 || !discounts || !out)
        return fail(CYMF_ERR_INVALID, "cymf_eval_run: bad arguments");
    int32_t kmax = 0;
    for (int32_t i = 0; i < nk; ++i) {
        if (ks[i] < 0) return fail(CYMF_ERR_INVALID, "cymf_eval_run: negative k");
        kmax = std::max(kmax, ks[i]);
    }
    int32_t max_test = 0;
    for (int32_t u = 0; u < h->U; ++u) max_test = std::max(max_test, h->h_test_indptr[(size_t)u + 1] - h->h_test_indptr[(size_t)u]);
    const int64_t rounds_max = std::min<int64_t>(kmax, (int64_t)max_test + num_negatives);
    if (rounds_max > 40000) return fail(CYMF_ERR_UNSUPPORTED, "cymf_eval_run: a ranked list of %lld entries exceeds the CU's LDS", (long long)rounds_max);
    if (unbiased && !h->d_prop.p) return fail(CYMF_ERR_INVALID, "cymf_eval_run: unbiased metrics need the propensity vector at create");
    CYMF_TRY(use_device(h->device));
    CYMF_TRY(eval_sample(h, seed, num_negatives));
    const int32_t n_eval = (int32_t)h->h_eval_users.size();
    const size_t n_out = (size_t)3 * nk * h->U;
    CYMF_TRY(h->d_out.alloc(n_out));
    CYMF_TRY(h->d_out.zero(h->stream));
    if (n_eval > 0) {
        CYMF_TRY(h->d_W.upload(W, (size_t)h->U * K, h->stream));
        CYMF_TRY(h->d_H.upload(H, (size_t)h->I * K, h->stream));
        CYMF_TRY(h->d_ks.upload(ks, (size_t)nk, h->stream));
        CYMF_TRY(h->d_disc.upload(discounts, (size_t)std::max(kmax, 1), h->stream));
        const int blocks = (n_eval + 3) / 4;
        hipLaunchKernelGGL(eval_score_kernel, dim3(blocks), dim3(256), 0, h->stream, h->d_W.p, h->d_H.p, K, h->d_eval_users.p, n_eval,
                           h->d_cand_off.p, h->d_test_indptr.p, h->d_test_indices.p, h->d_neg.p, num_negatives, h->d_scores.p);
        CYMF_HIP(hipGetLastError());
        if (kmax <= 64) {
            hipLaunchKernelGGL(eval_rank_kernel<false>, dim3(blocks), dim3(256), 0, h->stream, h->d_scores.p, h->d_eval_users.p, n_eval,
                               h->d_cand_off.p, h->d_test_indptr.p, num_negatives, h->d_prop.p, h->n_prop, unbiased, h->d_ks.p, nk, kmax,
                               h->d_disc.p, h->U, h->d_out.p);
        } else {   // ranked list in LDS, one user per workgroup
            const size_t smem = (size_t)std::max<int64_t>(rounds_max, 1) * sizeof(int32_t);
            if (smem > 48 * 1024)
                CYMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(eval_rank_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(eval_rank_kernel<true>, dim3(n_eval), dim3(64), smem, h->stream, h->d_scores.p, h->d_eval_users.p, n_eval,
                               h->d_cand_off.p, h->d_test_indptr.p, num_negatives, h->d_prop.p, h->n_prop, unbiased, h->d_ks.p, nk, kmax,
                               h->d_disc.p, h->U, h->d_out.p);
        }
        CYMF_HIP(hipGetLastError());
    }
    CYMF_HIP(hipMemcpyAsync(out, h->d_out.p, n_out * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cymf_eval_destroy(cymf_eval *h) {
    if (!h) return 0;
    if (!cymf::runtime_alive(h->device)) return 0;   // process exit / runtime already torn down: leak quietly
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
    return 0;
}
