/*
 * cymf_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this
 * file's shared object; the product (cymf_amd/) never does and fails loudly without its
 * HIP library.  Plain C99, fp64, one thread, the reference's operation order, built with
 * `gcc -O2 -ffp-contract=off` (the reference's x86-64 build has no FMA contraction:
 * /root/reference/setup.py:17 passes no -march).
 *
 * Pinning: the reference's own tests hold no vector for this path, so
 * the pin is the reference itself, compiled in the build container by oracle/build_ref.py
 * (into a directory outside this repository) and compared bit-for-bit in tests/test_oracle_vs_reference.py, plus the
 * golden fixtures under tests/golden/ that tests/golden/make_golden.py generated from it.
 * WMF is the exception: the reference's wmf/linalg modules are unbuildable here (cblas.h),
 * so orc_wmf_* is "parity unpinned" by the reference and cross-checked against numpy's
 * LAPACK dgesv only.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * mt19937 + uniform_int_distribution<long>   (cymf/math.pyx:12-18, cymf/math.pxd:31-39;
 * libstdc++-11 <random>, bits/uniform_int_dist.h:241-270 (Lemire), :281-352 (dispatch))
 * ---------------------------------------------------------------------------------- */
typedef struct {
    uint32_t mt[624];
    int idx;
    uint64_t range;      /* b - a  (number of values); a is always 0 in the reference */
    uint64_t n_raw;      /* raw 32-bit words consumed so far */
} orc_rng;

void orc_rng_init(orc_rng *g, uint32_t seed, uint64_t range)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
    g->range = range;
    g->n_raw = 0;
}

static void mt_twist(orc_rng *g)
{
    uint32_t *mt = g->mt;
    for (int k = 0; k < 624; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
}

uint32_t orc_rng_raw(orc_rng *g)
{
    if (g->idx >= 624) mt_twist(g);
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    g->n_raw++;
    return y;
}

/* _S_nd<uint64>(g, uint32 range): bits/uniform_int_dist.h:245-270 */
static uint64_t lemire32(orc_rng *g, uint32_t range)
{
    uint64_t product = (uint64_t)orc_rng_raw(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (uint32_t)(-range) % range;
        while (low < threshold) {
            product = (uint64_t)orc_rng_raw(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return product >> 32;
}

/* uniform_int_distribution<long>(0, urange)(g): bits/uniform_int_dist.h:281-352 */
static uint64_t uniform_closed(orc_rng *g, uint64_t urange)
{
    const uint64_t urngrange = 0xffffffffull;
    if (urngrange > urange)
        return lemire32(g, (uint32_t)(urange + 1));
    if (urngrange < urange) {
        uint64_t ret, tmp;
        do {
            tmp = (urngrange + 1) * uniform_closed(g, urange / (urngrange + 1));
            ret = tmp + (uint64_t)orc_rng_raw(g);
        } while (ret > urange || ret < tmp);
        return ret;
    }
    return (uint64_t)orc_rng_raw(g);
}

/* UniformGenerator.generate(): cymf/math.pyx:17-18 */
int64_t orc_rng_next(orc_rng *g) { return (int64_t)uniform_closed(g, g->range - 1); }

/* Testable index stream: draws [skip, skip+n) of UniformGenerator(0, range, seed). */
void orc_rng_fill_uniform(uint32_t seed, uint64_t range, int64_t n, int64_t skip, int64_t *out)
{
    orc_rng g;
    orc_rng_init(&g, seed, range);
    for (int64_t i = 0; i < skip; ++i) (void)orc_rng_next(&g);
    for (int64_t i = 0; i < n; ++i) out[i] = orc_rng_next(&g);
}

void orc_rng_fill_raw(uint32_t seed, int64_t n, uint32_t *out)
{
    orc_rng g;
    orc_rng_init(&g, seed, 2);
    for (int64_t i = 0; i < n; ++i) out[i] = orc_rng_raw(&g);
}

/* ------------------------------------------------------------------------------------
 * Optimizers (cymf/optimizer.pyx:52-58 Sgd, :64-82 AdaGrad, :126-160 Adam)
 * ---------------------------------------------------------------------------------- */
enum { ORC_SGD = 0, ORC_ADAGRAD = 1, ORC_ADAM = 2 };

typedef struct {
    int kind;
    double lr;
    /* AdaGrad: s0 = grad_accum (init ones, optimizer.pyx:69-70).
       Adam: s0 = M, s1 = V (init zeros, optimizer.pyx:143-146). */
    double *W0, *W1, *H0, *H1;
} orc_opt;

static inline double sq(double x) { return x * x; }   /* cymf/math.pxd:41-42 */

static inline void opt_update(const orc_opt *o, double *p, double *s0, double *s1, double g)
{
    switch (o->kind) {
    case ORC_SGD:                                         /* optimizer.pyx:53 */
        *p -= o->lr * g;
        break;
    case ORC_ADAGRAD:                                     /* optimizer.pyx:75-76 */
        *s0 += sq(g);
        *p -= o->lr * g / sqrt(*s0);
        break;
    default: {                                            /* optimizer.pyx:151-153 */
        const double beta1 = 0.9, beta2 = 0.999, eps = 1e-8;
        *s0 = beta1 * *s0 + (1 - beta1) * g;
        *s1 = beta2 * *s1 + (1 - beta2) * sq(g);
        *p -= o->lr * (*s0 / (1 - beta1)) / (sqrt(*s1 / (1 - beta2)) + eps);
    }
    }
}

static int opt_alloc(orc_opt *o, int kind, double lr, int64_t nW, int64_t nH)
{
    o->kind = kind; o->lr = lr;
    o->W0 = o->W1 = o->H0 = o->H1 = NULL;
    if (kind == ORC_ADAGRAD) {
        o->W0 = (double *)malloc(sizeof(double) * (size_t)nW);
        o->H0 = (double *)malloc(sizeof(double) * (size_t)nH);
        if (!o->W0 || !o->H0) return -1;
        for (int64_t i = 0; i < nW; ++i) o->W0[i] = 1.0;
        for (int64_t i = 0; i < nH; ++i) o->H0[i] = 1.0;
    } else if (kind == ORC_ADAM) {
        o->W0 = (double *)calloc((size_t)nW, sizeof(double));
        o->W1 = (double *)calloc((size_t)nW, sizeof(double));
        o->H0 = (double *)calloc((size_t)nH, sizeof(double));
        o->H1 = (double *)calloc((size_t)nH, sizeof(double));
        if (!o->W0 || !o->W1 || !o->H0 || !o->H1) return -1;
    }
    return 0;
}

static void opt_free(orc_opt *o) { free(o->W0); free(o->W1); free(o->H0); free(o->H1); }

/* membership of `item` in the sorted CSR row of `u` == std::set<int>::find (bpr.pyx:146-147,166) */
static inline int csr_has(const int32_t *indptr, const int32_t *indices, int32_t u, int32_t item)
{
    int32_t lo = indptr[u], hi = indptr[u + 1];
    while (lo < hi) {
        int32_t mid = lo + ((hi - lo) >> 1);
        int32_t v = indices[mid];
        if (v == item) return 1;
        if (v < item) lo = mid + 1; else hi = mid;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * BPR  (cymf/bpr.pyx:117-171 loop, cymf/model.pyx:47-62 forward, :66-87 backward)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t U, I, K;
    double wd;
    orc_opt opt;
    orc_rng gen;            /* one generator for the whole fit: bpr.pyx:141 */
    double *W, *H;          /* borrowed, trained in place: bpr.pyx:127-128 */
    int64_t n_skipped;      /* draws skipped so far (bpr.pyx:166-167) */
} orc_bpr;

orc_bpr *orc_bpr_create(int32_t U, int32_t I, int32_t K, int opt, double lr, double wd,
                        uint32_t neg_seed, double *W, double *H)
{
    orc_bpr *m = (orc_bpr *)calloc(1, sizeof(orc_bpr));
    if (!m) return NULL;
    m->U = U; m->I = I; m->K = K; m->wd = wd; m->W = W; m->H = H;
    if (opt_alloc(&m->opt, opt, lr, (int64_t)U * K, (int64_t)I * K)) { free(m); return NULL; }
    orc_rng_init(&m->gen, neg_seed, (uint64_t)I);
    return m;
}

void orc_bpr_destroy(orc_bpr *m) { if (m) { opt_free(&m->opt); free(m); } }
int64_t orc_bpr_skipped(const orc_bpr *m) { return m->n_skipped; }

/* One performed triplet: forward (model.pyx:47-62) then backward (model.pyx:66-87). */
static inline double bpr_triplet(orc_bpr *m, int32_t u, int32_t i, int32_t j)
{
    const int K = m->K;
    const double wd = m->wd;
    double *Wu = m->W + (int64_t)u * K, *Hi = m->H + (int64_t)i * K, *Hj = m->H + (int64_t)j * K;
    double x = 0.0, l2 = 0.0;
    for (int k = 0; k < K; ++k) {
        x += Wu[k] * (Hi[k] - Hj[k]);
        l2 += sq(Wu[k]) + sq(Hi[k]) + sq(Hj[k]);
    }
    double loss = -log(1.0 / (1.0 + exp(-x))) + wd * l2;
    double s = 1.0 / (1.0 + exp(x));
    const orc_opt *o = &m->opt;
    const int64_t ou = (int64_t)u * K, oi = (int64_t)i * K, oj = (int64_t)j * K;
    for (int k = 0; k < K; ++k) {
        double gw = -(s * (Hi[k] - Hj[k]) - wd * Wu[k]);
        double gi = -(s * Wu[k] - wd * Hi[k]);
        double gj = -(s * (-Wu[k]) - wd * Hj[k]);
        opt_update(o, &Wu[k], o->W0 ? &o->W0[ou + k] : NULL, o->W1 ? &o->W1[ou + k] : NULL, gw);
        opt_update(o, &Hi[k], o->H0 ? &o->H0[oi + k] : NULL, o->H1 ? &o->H1[oi + k] : NULL, gi);
        opt_update(o, &Hj[k], o->H0 ? &o->H0[oj + k] : NULL, o->H1 ? &o->H1[oj + k] : NULL, gj);
    }
    return loss;
}

/* One epoch over the fixed triplet order (bpr.pyx:160-171). Returns accum_loss / N.
 * If negatives_out != NULL it receives the N draws of this epoch (skipped ones included). */
double orc_bpr_epoch(orc_bpr *m, const int32_t *users, const int32_t *positives, int64_t N,
                     const int32_t *indptr, const int32_t *indices, int32_t *negatives_out)
{
    double accum = 0.0;
    for (int64_t l = 0; l < N; ++l) {
        int32_t u = users[l], i = positives[l];
        int32_t j = (int32_t)orc_rng_next(&m->gen);
        if (negatives_out) negatives_out[l] = j;
        if (csr_has(indptr, indices, u, j)) { m->n_skipped++; continue; }
        accum += bpr_triplet(m, u, i, j);
    }
    return N ? accum / (double)N : 0.0;
}

/* HOGWILD counterpart of orc_bpr_epoch for the cpu_baseline timing leg only: the reference's
 * `prange(N, schedule="guided")` over lock-free shared W/H (bpr.pyx:162), with the epoch's draws taken
 * from the stream up front (the reference shares one unlocked mt19937 between its threads, which is a
 * data race; drawing first gives every triplet the draw of its position).  Nondeterministic by design. */
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

double orc_bpr_epoch_hogwild(orc_bpr *m, const int32_t *users, const int32_t *positives, int64_t N,
                             const int32_t *indptr, const int32_t *indices, int n_threads, int64_t *performed_out)
{
    int32_t *neg = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    for (int64_t l = 0; l < N; ++l) neg[l] = (int32_t)orc_rng_next(&m->gen);
    double accum = 0.0;
    int64_t performed = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(guided) reduction(+ : accum, performed)
#endif
    for (int64_t l = 0; l < N; ++l) {
        if (csr_has(indptr, indices, users[l], neg[l])) continue;
        accum += bpr_triplet(m, users[l], positives[l], neg[l]);
        performed++;
    }
    (void)n_threads;
    free(neg);
    if (performed_out) *performed_out = performed;
    return N ? accum / (double)N : 0.0;
}

/* Same triplet arithmetic, caller-supplied (u,i,j) list in the given order: used by the
 * host-logic tests (user-sharded multi-rank emulation) and the cpu_baseline timing leg. */
double orc_bpr_apply(orc_bpr *m, const int32_t *u, const int32_t *i, const int32_t *j, int64_t n)
{
    double accum = 0.0;
    for (int64_t t = 0; t < n; ++t) accum += bpr_triplet(m, u[t], i[t], j[t]);
    return accum;
}

/* ------------------------------------------------------------------------------------
 * RelMF (cymf/relmf.pyx:142-148 loop, cymf/model.pyx:99-119 forward, :123-142 backward)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t U, I, K;
    double wd, clip;
    orc_opt opt;
    orc_rng gen;           /* UniformGenerator(0, U*I, 1234): relmf.pyx:128 */
    double *W, *H;
} orc_relmf;

orc_relmf *orc_relmf_create(int32_t U, int32_t I, int32_t K, int opt, double lr, double wd,
                            double clip, uint32_t seed, double *W, double *H)
{
    orc_relmf *m = (orc_relmf *)calloc(1, sizeof(orc_relmf));
    if (!m) return NULL;
    m->U = U; m->I = I; m->K = K; m->wd = wd; m->clip = clip; m->W = W; m->H = H;
    if (opt_alloc(&m->opt, opt, lr, (int64_t)U * K, (int64_t)I * K)) { free(m); return NULL; }
    orc_rng_init(&m->gen, seed, (uint64_t)U * (uint64_t)I);
    return m;
}

void orc_relmf_destroy(orc_relmf *m) { if (m) { opt_free(&m->opt); free(m); } }

static inline double dmax(double a, double b) { return a >= b ? a : b; }   /* math.pxd:47-51 */

/* One epoch = U*I draws with replacement (relmf.pyx:143-148). X dense row-major (U,I).
 * Returns the sum of per-draw losses (relmf.pyx:150-152); draws_out (optional) gets the
 * U*I cell indices drawn. */
double orc_relmf_epoch(orc_relmf *m, const double *X, const double *prop, int64_t *draws_out)
{
    const int K = m->K;
    const int64_t I = m->I, N = (int64_t)m->U * I;
    const double wd = m->wd, M = m->clip;
    const orc_opt *o = &m->opt;
    double accum = 0.0;
    for (int64_t l = 0; l < N; ++l) {
        int64_t rnd = orc_rng_next(&m->gen);
        if (draws_out) draws_out[l] = rnd;
        int64_t u = rnd / I, i = rnd % I;
        double r = X[u * I + i], p = prop[i];
        double *Wu = m->W + u * K, *Hi = m->H + i * K;
        double y = 0.0, l2 = 0.0;
        for (int k = 0; k < K; ++k) {                      /* model.pyx:113-115 */
            y += Wu[k] * Hi[k];
            l2 += sq(Wu[k]) + sq(Hi[k]);
        }
        accum += (r / dmax(p, M)) * sq(1. - y) + (1 - r / dmax(p, M)) * sq(y) + wd * l2;
        for (int k = 0; k < K; ++k) {                      /* model.pyx:130-142 */
            double gw = -((r / dmax(p, M)) * (1. - y) * Hi[k] +
                          (1 - r / dmax(p, M)) * (0. - y) * Hi[k]) + wd * Wu[k];
            double gh = -((r / dmax(p, M)) * (1. - y) * Wu[k] +
                          (1 - r / dmax(p, M)) * (0. - y) * Wu[k]) + wd * Hi[k];
            opt_update(o, &Wu[k], o->W0 ? &o->W0[u * K + k] : NULL, o->W1 ? &o->W1[u * K + k] : NULL, gw);
            opt_update(o, &Hi[k], o->H0 ? &o->H0[i * K + k] : NULL, o->H1 ? &o->H1[i * K + k] : NULL, gh);
        }
    }
    return accum;
}

/* ------------------------------------------------------------------------------------
 * GloVe (cymf/glove.pyx:149-156 loop, cymf/model.pyx:166-181 forward, :185-204 backward,
 *        cymf/optimizer.pyx:85-123 GloVeAdaGrad, accumulators init ones :96-99)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t V, Vc, K;
    double lr, x_max, alpha;
    double *W, *H, *bW, *bH;       /* borrowed */
    double *aW, *aH, *abW, *abH;   /* AdaGrad accumulators */
} orc_glove;

orc_glove *orc_glove_create(int32_t V, int32_t Vc, int32_t K, double lr, double x_max, double alpha,
                            double *W, double *bW, double *H, double *bH)
{
    orc_glove *m = (orc_glove *)calloc(1, sizeof(orc_glove));
    if (!m) return NULL;
    m->V = V; m->Vc = Vc; m->K = K; m->lr = lr; m->x_max = x_max; m->alpha = alpha;
    m->W = W; m->H = H; m->bW = bW; m->bH = bH;
    int64_t nW = (int64_t)V * K, nH = (int64_t)Vc * K;
    m->aW = (double *)malloc(sizeof(double) * (size_t)nW);
    m->aH = (double *)malloc(sizeof(double) * (size_t)nH);
    m->abW = (double *)malloc(sizeof(double) * (size_t)V);
    m->abH = (double *)malloc(sizeof(double) * (size_t)V);   /* _bias is sized X.shape[0]: glove.pyx:94 */
    for (int64_t i = 0; i < nW; ++i) m->aW[i] = 1.0;
    for (int64_t i = 0; i < nH; ++i) m->aH[i] = 1.0;
    for (int64_t i = 0; i < V; ++i) m->abW[i] = m->abH[i] = 1.0;
    return m;
}

void orc_glove_destroy(orc_glove *m)
{
    if (m) { free(m->aW); free(m->aH); free(m->abW); free(m->abH); free(m); }
}

static inline void adagrad(double *p, double *acc, double lr, double g)
{
    *acc += sq(g);
    *p -= lr * g / sqrt(*acc);
}

/* One iteration over the fixed (central, context, count) order. Returns sum of loss[l]. */
double orc_glove_epoch(orc_glove *m, const int32_t *central, const int32_t *context,
                       const double *counts, int64_t N)
{
    const int K = m->K;
    const double lr = m->lr;
    double accum = 0.0;
    for (int64_t l = 0; l < N; ++l) {
        const int64_t c = central[l], x = context[l];
        double *Wc = m->W + c * K, *Hx = m->H + x * K;
        double diff = 0.0;
        for (int k = 0; k < K; ++k) diff += Wc[k] * Hx[k];            /* model.pyx:174-175 */
        diff += m->bW[c] + m->bH[x];                                    /* :176 */
        diff -= log(counts[l]);                                         /* :177 */
        double tmp = diff;
        diff *= fmin(pow(counts[l] / m->x_max, m->alpha), 1.0);         /* :179, weight_func :34-35 */
        accum += 0.5 * diff * tmp;                                      /* :180 */
        for (int k = 0; k < K; ++k) {                                   /* :195-204 */
            double gw = diff * Hx[k];
            double gh = diff * Wc[k];
            adagrad(&Wc[k], &m->aW[c * K + k], lr, gw);
            adagrad(&Hx[k], &m->aH[x * K + k], lr, gh);
            adagrad(&m->bW[c], &m->abW[c], lr, diff);                   /* K times per sample */
            adagrad(&m->bH[x], &m->abH[x], lr, diff);
        }
    }
    return accum;
}

/* ------------------------------------------------------------------------------------
 * WMF ALS half-sweep (cymf/wmf.pyx:136-174) + solvep = LAPACK dgesv (cymf/linalg.pyx:144-163)
 * dgesv is restated as unblocked LU with partial pivoting (LAPACK dgetf2 + dgetrs); LAPACK's
 * blocked dgetrf may round differently at the 1e-16 level.  "Parity unpinned" (see header).
 * ---------------------------------------------------------------------------------- */
static int lu_solve(double *A, double *b, int n)   /* A column-major n x n, overwritten */
{
    int info = 0;
    for (int c = 0; c < n; ++c) {
        int p = c; double best = fabs(A[c + (size_t)c * n]);
        for (int r = c + 1; r < n; ++r) {
            double v = fabs(A[r + (size_t)c * n]);
            if (v > best) { best = v; p = r; }
        }
        if (A[p + (size_t)c * n] == 0.0) { if (!info) info = c + 1; continue; }
        if (p != c) {
            for (int q = 0; q < n; ++q) {
                double t = A[c + (size_t)q * n]; A[c + (size_t)q * n] = A[p + (size_t)q * n]; A[p + (size_t)q * n] = t;
            }
            double t = b[c]; b[c] = b[p]; b[p] = t;
        }
        double inv = 1.0 / A[c + (size_t)c * n];
        for (int r = c + 1; r < n; ++r) A[r + (size_t)c * n] *= inv;
        for (int q = c + 1; q < n; ++q) {
            double f = A[c + (size_t)q * n];
            for (int r = c + 1; r < n; ++r) A[r + (size_t)q * n] -= A[r + (size_t)c * n] * f;
        }
    }
    if (info) return info;
    for (int c = 0; c < n; ++c)                       /* L y = b (unit lower) */
        for (int r = c + 1; r < n; ++r) b[r] -= A[r + (size_t)c * n] * b[c];
    for (int c = n - 1; c >= 0; --c) {                /* U x = y */
        b[c] /= A[c + (size_t)c * n];
        for (int r = 0; r < c; ++r) b[r] -= A[r + (size_t)c * n] * b[c];
    }
    return 0;
}

/* X[rows,K] <- solve per row; Y[cols,K] fixed. indptr/indices: CSR pattern of the rows. */
void orc_wmf_half_sweep(int32_t rows, int32_t cols, int32_t K, const int32_t *indptr,
                        const int32_t *indices, double *X, const double *Y,
                        double weight, double weight_decay)
{
    const size_t KK = (size_t)K * K;
    double *A0 = (double *)calloc(KK, sizeof(double));
    double *A = (double *)malloc(KK * sizeof(double));
    double *b = (double *)malloc((size_t)K * sizeof(double));
    for (int64_t j = 0; j < cols; ++j)                 /* YtY = np.dot(Y.T, Y): wmf.pyx:142 */
        for (int k = 0; k < K; ++k)
            for (int k2 = 0; k2 < K; ++k2) A0[(size_t)k * K + k2] += Y[j * K + k] * Y[j * K + k2];
    for (int k = 0; k < K; ++k) A0[(size_t)k * K + k] += weight_decay;   /* wmf.pyx:143 */
    for (int64_t i = 0; i < rows; ++i) {
        if (indptr[i] == indptr[i + 1]) {                                 /* wmf.pyx:154-156 */
            memset(X + i * K, 0, sizeof(double) * (size_t)K);
            continue;
        }
        memcpy(A, A0, KK * sizeof(double));
        memset(b, 0, sizeof(double) * (size_t)K);
        for (int32_t ptr = indptr[i]; ptr < indptr[i + 1]; ++ptr) {       /* wmf.pyx:161-166 */
            const double *y = Y + (int64_t)indices[ptr] * K;
            for (int k = 0; k < K; ++k) {
                b[k] += y[k] * weight;
                for (int k2 = 0; k2 < K; ++k2) A[(size_t)k * K + k2] += y[k] * y[k2] * (weight - 1.0);
            }
        }
        lu_solve(A, b, K);                                                /* wmf.pyx:168 */
        memcpy(X + i * K, b, sizeof(double) * (size_t)K);                 /* wmf.pyx:170-171 */
    }
    free(A0); free(A); free(b);
}

/* ------------------------------------------------------------------------------------
 * Metrics on a 0/1 vector sorted by score (cymf/metrics.pyx:24-43 DCG, :71-85 Recall,
 * :109-125 MAP)
 * ---------------------------------------------------------------------------------- */
double orc_dcg_at_k(const int32_t *y, int n, int k)
{
    double dcg = (double)y[0], counter = 0.0;
    for (int i = 0; i < n; ++i) {
        if (1 <= i && i < k) dcg += (double)y[i] / log2((double)i + 1.0);
        counter += (double)y[i];
    }
    return counter == 0.0 ? 0.0 : dcg / counter;
}

double orc_recall_at_k(const int32_t *y, int n, int k)
{
    double rec = 0.0, counter = 0.0;
    for (int i = 0; i < n; ++i) {
        if (i < k) rec += (double)y[i];
        counter += (double)y[i];
    }
    return counter == 0.0 ? 0.0 : rec / counter;
}

double orc_ap_at_k(const int32_t *y, int n, int k)
{
    double ap = 0.0, counter = 0.0;
    for (int i = 0; i < n; ++i) {
        counter += (double)y[i];
        if (i < k && y[i] == 1) ap += counter / ((double)i + 1.0);
    }
    return counter == 0.0 ? 0.0 : ap / counter;
}
