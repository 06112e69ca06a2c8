/*
 * cymf_amd.h -- C ABI of libcymf_hip.so, the MI355X (gfx950) implementation of cymf's hot path.
 *
 * The reference (minatosato/cymf) has no FFI of its own: its boundary is the Python class
 * surface cymf.BPR / WMF / GloVe / RelMF.  Each group below replaces the
 * Cython `_fit_*` body named beside it; the cymf_amd Python modules bind these entry points with ctypes
 * and keeps the reference's class surface on top (INTEGRATION.md shows the stub a
 * maintainer of the reference would add).  Paths are relative to /root/reference.
 *
 * Conventions
 *   - every function returns 0 on success, a negative cymf_status on failure;
 *     cymf_last_error() gives the message (thread-local).  The reference's native code
 *     never signals errors (dgesv info ignored, cymf/wmf.pyx:168); the Python wrappers
 *     raise RuntimeError on a non-zero status.
 *   - pointers are host pointers, borrowed for the duration of the call; a handle owns
 *     its device memory.  Host factor matrices are C-contiguous float64 (the reference's
 *     `double[:, ::1]`, cymf/bpr.pyx:127-128); indices are int32 (cymf/bpr.pyx:118-119).
 *   - there is NO CPU fallback: without a gfx950 device every compute entry point fails
 *     with CYMF_ERR_NO_DEVICE.
 */
#ifndef CYMF_AMD_H
#define CYMF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CYMF_OK = 0,
    CYMF_ERR_INVALID = -1,     /* bad argument / call order */
    CYMF_ERR_NO_DEVICE = -2,   /* no gfx950 device visible */
    CYMF_ERR_HIP = -3,         /* a HIP runtime call failed */
    CYMF_ERR_RCCL = -4,        /* an RCCL call failed */
    CYMF_ERR_UNSUPPORTED = -5, /* valid request this build does not implement */
    CYMF_ERR_NOMEM = -6
} cymf_status;

/* optimizer ids: cymf/bpr.pyx:149-154 ("sgd" | "adagrad" | "adam") */
enum { CYMF_OPT_SGD = 0, CYMF_OPT_ADAGRAD = 1, CYMF_OPT_ADAM = 2 };
/* device arithmetic / storage type of the factors */
enum { CYMF_F32 = 0, CYMF_F64 = 1 };
/* execution mode.
 * EXACT      : the reference's sequential order (num_threads == 1, the only deterministic
 *              setting) reproduced by level scheduling: triplets that touch
 *              disjoint rows run together, dependent ones in order.
 * THROUGHPUT : the reference's HOGWILD regime (num_threads > 1, cymf/bpr.pyx:75,162):
 *              triplets bucketed by positive item, unordered, lock-free. */
enum { CYMF_MODE_EXACT = 0, CYMF_MODE_THROUGHPUT = 1 };

/* ---------------------------------------------------------------- library / device */
const char *cymf_last_error(void);
int cymf_version(void);
int cymf_device_count(void);                                  /* replaces cymf::cpucount(), cymf/util.h:15 */
int cymf_device_name(int device, char *buf, int buflen);
int cymf_device_sync(int device);
/* measured HBM rate of a float4 device-to-device stream copy (read + write bytes per second, in GB/s):
 * the achievable-bandwidth yardstick printed beside the 8 TB/s datasheet peak */
int cymf_device_stream_copy_gbps(int device, int64_t bytes, int iters, double *gbps_out);
/* Diagnostic (no reference counterpart): does this box keep plain device memory coherent across the seams the
 * trainers rely on?  `rounds` times, in a buffer of memory type `memtype` (0 hipMalloc, 1 fine-grained, 2 uncached):
 *   every CU reads the buffer (all eight XCD L2s hold the old value) -> ONE workgroup rewrites it ->
 *   [0] every CU re-reads it in the next kernel of the same stream, [1] one workgroup re-reads it,
 *   [2] every CU re-reads it on a second stream behind an event, [3] the copy engine brings it to the host.
 * stale_out[4] receives the number of words that still showed the old value at each seam, *words_out the words
 * checked per seam.  All zero on a healthy stack; DESIGN.md section 2 records what the GPU boxes answered. */
int cymf_device_seam_probe(int device, int memtype, int rounds, int64_t *stale_out, int64_t *words_out);

/* ---------------------------------------------------------------- negative-sample index stream
 * UniformGenerator(a=0, b=range, seed): cymf/math.pyx:12-18, cymf/math.pxd:31-39
 * = std::mt19937(seed) + std::uniform_int_distribution<long>(0, range-1) (libstdc++-11:
 * Lemire rejection on 32-bit words below 2^32; one raw word at 2^32; high part + low word with rejection above,
 * bits/uniform_int_dist.h:281-352 -- RelMF draws cells from range U*I, cymf/relmf.pyx:128).
 * Generated ON THE DEVICE; bit-exact.  Fills out[0..n) with draws [skip, skip+n) of the stream.
 * range must be in [1, 2^62]. */
int cymf_rng_fill_uniform(int device, uint32_t seed, uint64_t range, int64_t n, int64_t skip,
                          int64_t *out);

/* ---------------------------------------------------------------- BPR
 * replaces BPR._fit_bpr, cymf/bpr.pyx:117-190 (setup :127-157, epoch loop :160-171) with
 * BprModel.forward/backward (cymf/model.pyx:47-87) and Sgd/AdaGrad/Adam
 * (cymf/optimizer.pyx:40-160) fused into the step kernels. */
typedef struct cymf_bpr cymf_bpr;

int cymf_bpr_create(cymf_bpr **out, int32_t U, int32_t I, int32_t K, int optimizer,
                    double learning_rate, double weight_decay, uint32_t neg_seed,
                    int dtype, int mode, int device);
/* users/positives: the shuffled X.nonzero() order of cymf/bpr.pyx:104-107 (length N);
 * indptr[U+1]/indices: CSR pattern of X with SORTED indices, the device form of
 * `vector<set<int>> user_positives` (cymf/bpr.pyx:140,146-147).
 * Sharded use: this rank holds the triplets of its own users only;
 * global_pos[N] (may be NULL = 0..N-1) are their positions in the global order of
 * N_global triplets, so that triplet l of epoch e consumes draw e*N_global + l of the ONE
 * global stream (cymf/bpr.pyx:141,165). */
int cymf_bpr_set_data(cymf_bpr *h, const int32_t *users, const int32_t *positives, int64_t N,
                      const int32_t *indptr, const int32_t *indices,
                      const int64_t *global_pos, int64_t N_global);
/* THROUGHPUT mode: number of steps an epoch is cut into (windows of the global order).
 * Default 1; 0 = auto (see cymf_bpr_get_steps_per_epoch).  With a communicator attached the item-factor deltas of all ranks are
 * summed after every step (RCCL all-reduce).  Call before cymf_bpr_set_data. */
int cymf_bpr_set_steps_per_epoch(cymf_bpr *h, int32_t steps);
/* steps = 0 above asks for "auto": chosen from the data at cymf_bpr_set_data -- enough windows that the lock-free mode follows
 * the reference's shuffled order (cymf/bpr.pyx:104) closely; this returns the number in use. */
int cymf_bpr_get_steps_per_epoch(cymf_bpr *h, int32_t *steps);
int cymf_bpr_upload(cymf_bpr *h, const double *W, const double *H);       /* H2D, W:(U,K) H:(I,K) */
int cymf_bpr_download(cymf_bpr *h, double *W, double *H);                 /* D2H */
/* n_epochs passes of cymf/bpr.pyx:160-171; loss_out[e] = accum_loss / N (:171), may be NULL */
int cymf_bpr_epochs(cymf_bpr *h, int32_t n_epochs, double *loss_out);
/* THROUGHPUT mode: advance n_steps steps (wrapping over epoch boundaries); asynchronous
 * w.r.t. the host until cymf_bpr_sync / download.  loss_sum_out (may be NULL) forces a sync. */
int cymf_bpr_steps(cymf_bpr *h, int32_t n_steps, double *loss_sum_out);
/* waits for everything the handle has issued: the step stream and the side stream that prepares the next epoch's
 * negatives (index stream, skip tests) */
int cymf_bpr_sync(cymf_bpr *h);
/* counters since create: performed triplet updates, skipped draws (cymf/bpr.pyx:166-167) */
int cymf_bpr_stats(cymf_bpr *h, int64_t *performed, int64_t *skipped);
/* milliseconds and launches of the dominant kernel since the last call (hipEvent-timed on
 * the kernel's own stream; timing is enabled by cymf_bpr_set_profiling(h, 1)) */
int cymf_bpr_set_profiling(cymf_bpr *h, int on);
int cymf_bpr_kernel_time(cymf_bpr *h, double *ms_total, int64_t *launches, int64_t *units);
/* the epoch's draws in original triplet order, -1 where the draw was skipped (test hook) */
int cymf_bpr_last_negatives(cymf_bpr *h, int32_t *out, int64_t n);
int cymf_bpr_destroy(cymf_bpr *h);

/* ---------------------------------------------------------------- multi-GPU (one process per GPU)
 * The reference is single-process (OpenMP only); this is the one exchange
 * step of the user-sharded design: sum of item-factor deltas over RCCL/xGMI. */
typedef struct cymf_comm cymf_comm;
#define CYMF_UNIQUE_ID_BYTES 128
int cymf_comm_unique_id(char id[CYMF_UNIQUE_ID_BYTES]);                     /* rank 0 */
int cymf_comm_create(cymf_comm **out, const char id[CYMF_UNIQUE_ID_BYTES], int rank, int world,
                     int device);
int cymf_comm_destroy(cymf_comm *c);
/* world handles inside ONE process on ONE device whose collectives meet in device memory (slots of max_floats floats per
 * rank): lets the sharded trainers run as several ranks, one host thread each, on a one-GPU box -- RCCL refuses two ranks
 * on one device.  A test and debugging vehicle, not a transport. */
int cymf_comm_create_local_group(cymf_comm **out, int world, int device, int64_t max_floats);
/* host buffer all-reduce (op 0 = sum, 1 = max): rendezvous/timing helper and test hook */
int cymf_comm_allreduce_f32(cymf_comm *c, float *host_inout, int64_t n, int op);
/* call before cymf_bpr_set_data (the per-step item counts are all-reduced there).  Throughput mode, any
 * optimizer: only H is exchanged, optimizer state of the item rows stays private to the rank. */
int cymf_bpr_attach_comm(cymf_bpr *h, cymf_comm *c);
/* optional: the user ranges [bounds[r], bounds[r+1]) of all ranks; cymf_bpr_download then gathers the rows of W, so that
 * every rank returns the whole trained model (without it a rank returns its own rows, the others at their initial values) */
int cymf_bpr_set_user_bounds(cymf_bpr *h, const int64_t *bounds);

/* ---------------------------------------------------------------- RelMF
 * replaces RelMF._fit_relmf, cymf/relmf.pyx:106-171 (loop :142-148) with
 * RelMfModel.forward/backward (cymf/model.pyx:99-142).  X is the dense (U,I) float64
 * matrix the reference builds (cymf/relmf.pyx:79-81), propensities its :88. */
typedef struct cymf_relmf cymf_relmf;
int cymf_relmf_create(cymf_relmf **out, int32_t U, int32_t I, int32_t K, int optimizer,
                      double learning_rate, double weight_decay, double clip_value,
                      uint32_t seed, int dtype, int mode, int device);
int cymf_relmf_set_data(cymf_relmf *h, const double *X, const double *propensities);
int cymf_relmf_upload(cymf_relmf *h, const double *W, const double *H);
int cymf_relmf_download(cymf_relmf *h, double *W, double *H);
int cymf_relmf_epochs(cymf_relmf *h, int32_t n_epochs, double *loss_out);   /* loss_out[e] = sum of loss[l], :150-152 */
int cymf_relmf_destroy(cymf_relmf *h);
/* Multi-GPU: users sharded by the given ranges (every rank draws the whole cell stream and works through its own users), item
 * table replicated, its deltas all-reduced after each of the steps_per_epoch sub-steps; download() gathers the user rows.
 * float32 throughput mode.  attach before cymf_relmf_upload. */
int cymf_relmf_set_steps_per_epoch(cymf_relmf *h, int32_t steps);
int cymf_relmf_attach_comm(cymf_relmf *h, cymf_comm *c, const int64_t *user_bounds);

/* ---------------------------------------------------------------- GloVe
 * replaces GloVe._fit_glove, cymf/glove.pyx:117-162 (loop :149-156) with
 * GloVeModel.forward/backward (cymf/model.pyx:166-204) and GloVeAdaGrad
 * (cymf/optimizer.pyx:85-123). */
typedef struct cymf_glove cymf_glove;
int cymf_glove_create(cymf_glove **out, int32_t V, int32_t Vc, int32_t K, double learning_rate,
                      double x_max, double alpha, int dtype, int mode, int device);
int cymf_glove_set_data(cymf_glove *h, const int32_t *central, const int32_t *context,
                        const double *counts, int64_t N);
int cymf_glove_upload(cymf_glove *h, const double *W, const double *bias, const double *Wc,
                      const double *bias_c);
int cymf_glove_download(cymf_glove *h, double *W, double *bias, double *Wc, double *bias_c);
int cymf_glove_epochs(cymf_glove *h, int32_t n_epochs, double *loss_out);   /* loss_out[e] = sum of loss[l], :155-156 */
int cymf_glove_destroy(cymf_glove *h);
/* Multi-GPU (no counterpart in the reference): pairs are sharded by central word -- rank r passes only the pairs whose
 * central word lies in [central_bounds[r], central_bounds[r+1]) -- the context table is replicated and its deltas
 * (rows, AdaGrad accumulators, bias pairs) are all-reduced after each of the steps_per_epoch steps (windows of the
 * rank's pair order); download() gathers the central rows, so every rank returns the full tables.  float32
 * throughput mode, V == Vc.  Both calls precede cymf_glove_set_data. */
int cymf_glove_set_steps_per_epoch(cymf_glove *h, int32_t steps);
int cymf_glove_attach_comm(cymf_glove *h, cymf_comm *c, const int64_t *central_bounds);

/* ---------------------------------------------------------------- WMF
 * replaces WMF._als, cymf/wmf.pyx:136-174 (Gramian :142-143, per-row accumulate :161-166)
 * and solvep / LAPACK dgesv, cymf/linalg.pyx:144-163.  One call = one half-sweep:
 * X[rows,K] <- argmin given fixed Y[cols,K] and the CSR pattern of the rows. */
typedef struct cymf_wmf cymf_wmf;
int cymf_wmf_create(cymf_wmf **out, int32_t U, int32_t I, int32_t K, double weight,
                    double weight_decay, int dtype, int device);
/* CSR pattern of X (users x items) and of its transpose, both with int32 indices */
int cymf_wmf_set_data(cymf_wmf *h, const int32_t *indptr, const int32_t *indices,
                      const int32_t *t_indptr, const int32_t *t_indices);
int cymf_wmf_upload(cymf_wmf *h, const double *W, const double *H);
int cymf_wmf_download(cymf_wmf *h, double *W, double *H);
/* side 0: users (W given H), side 1: items (H given W): cymf/wmf.pyx:111-112 */
int cymf_wmf_half_sweep(cymf_wmf *h, int side);
int cymf_wmf_epochs(cymf_wmf *h, int32_t n_epochs);
int cymf_wmf_destroy(cymf_wmf *h);
/* Multi-GPU (no counterpart in the reference; its prange over rows, cymf/wmf.pyx:150, is the same independence):
 * the rows of each side are cut into one contiguous range per rank, a rank solves its range and the updated
 * table is all-gathered after every half-sweep, so every rank holds the full W and H -- the results do not
 * depend on the number of ranks.  Every rank passes the full CSR to cymf_wmf_set_data; call before it. */
int cymf_wmf_attach_comm(cymf_wmf *h, cymf_comm *c);
/* rows [lo, hi) of side 0 (users) / 1 (items) this handle solves (the whole side without a communicator) */
int cymf_wmf_row_range(cymf_wmf *h, int side, int32_t *lo, int32_t *hi);

/* ---------------------------------------------------------------- Evaluator
 * replaces the per-user loop of Evaluator.evaluate, cymf/evaluator.pyx:57-139: candidate
 * sampling from UniformGenerator(0, I, seed) with redraw on known positives (:80-88), the
 * scores np.dot(H[items], W[user]) and their descending order (:90), and the metrics of
 * cymf/metrics.pyx:24-147 (DCG, Recall, MAP @k and their IPS variants, which index the
 * propensities by candidate position: evaluator.pyx:92).
 *   X (test) and user_positives (test + train, sorted unique indices) are CSR patterns.
 *   cymf_eval_run: out[(m * nk + ki) * U + user], m = 0 DCG, 1 Recall, 2 MAP; users without
 *   held-out items get 0 (they still count in the caller's mean, :134-136).
 *   discounts[r] = 1 for r = 0, log2(r + 1) after (cymf/metrics.pyx:36-41), max(k) entries. */
typedef struct cymf_eval cymf_eval;
int cymf_eval_create(cymf_eval **out, int32_t U, int32_t I, const int32_t *test_indptr,
                     const int32_t *test_indices, const int32_t *all_indptr,
                     const int32_t *all_indices, const double *propensity, int32_t n_propensity,
                     int device);
int cymf_eval_num_users(cymf_eval *h, int32_t *n_eval);
/* the sampled negatives, [n_eval][num_negatives] in user order, and the evaluated users */
int cymf_eval_negatives(cymf_eval *h, uint32_t seed, int32_t num_negatives, int32_t *users_out,
                        int32_t *neg_out, int64_t *draws_used);
int cymf_eval_run(cymf_eval *h, const double *W, const double *H, int32_t K, uint32_t seed,
                  int32_t num_negatives, const int32_t *ks, int32_t nk, const double *discounts,
                  int unbiased, double *out);
int cymf_eval_destroy(cymf_eval *h);

/* ---------------------------------------------------------------- ExpoMF
 * replaces ExpoMF._fit_als / _als, cymf/expomf.pyx:105-207: per epoch the dense exposure E-step (:141-144), the
 * user and item solves with the exposure-weighted Gramian over ALL columns (:163-204, solvep = dgesv), and the
 * update of the exposure priors mu (:149).  float64; CSR patterns of X and of its transpose as for WMF. */
typedef struct cymf_expomf cymf_expomf;
int cymf_expomf_create(cymf_expomf **out, int32_t U, int32_t I, int32_t K, double lam_y, double weight_decay,
                       int device);
int cymf_expomf_set_data(cymf_expomf *h, const int32_t *indptr, const int32_t *indices,
                         const int32_t *t_indptr, const int32_t *t_indices);
int cymf_expomf_upload(cymf_expomf *h, const double *W, const double *H);
int cymf_expomf_download(cymf_expomf *h, double *W, double *H);
int cymf_expomf_epochs(cymf_expomf *h, int32_t n_epochs);
int cymf_expomf_destroy(cymf_expomf *h);

#ifdef __cplusplus
}
#endif
#endif /* CYMF_AMD_H */
