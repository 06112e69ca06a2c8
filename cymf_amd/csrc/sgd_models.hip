// sgd_models.hip -- the two other per-sample SGD loops of the reference on the BPR skeleton
// (one wavefront per sample, K across lanes, DPP reduction, fused optimizer):
//   RelMF : RelMF._fit_relmf  cymf/relmf.pyx:106-171 (loop :142-148),
//           RelMfModel.forward/backward cymf/model.pyx:99-142
//   GloVe : GloVe._fit_glove  cymf/glove.pyx:117-162 (loop :149-156),
//           GloVeModel.forward/backward cymf/model.pyx:166-204, GloVeAdaGrad cymf/optimizer.pyx:85-123
// EXACT mode = the reference's sequential order by level scheduling over the two rows a sample
// touches; THROUGHPUT mode = HOGWILD over the same samples, one launch per epoch.
#include <algorithm>
#include <cmath>

#include "relmf_tiles.h"
#include "store.h"

namespace cymf {
namespace {

// ------------------------------------------------------------------ level scheduling (host)
// Samples s = 0..n-1 touch row a[s] of table A and row b[s] of table B.  Returns the samples
// ordered by level (stable) and the level offsets; two samples of one level share no row.
void level_schedule(int64_t n, const int32_t *a, const int32_t *b, int32_t nA, int32_t nB,
                    std::vector<int64_t> &order, std::vector<int64_t> &off) {
    std::vector<int32_t> lastA((size_t)nA, 0), lastB((size_t)nB, 0), level((size_t)n);
    int32_t nl = 0;
    for (int64_t s = 0; s < n; ++s) {
        int32_t lv = std::max(lastA[a[s]], lastB[b[s]]) + 1;
        lastA[a[s]] = lastB[b[s]] = lv;
        level[s] = lv;
        nl = std::max(nl, lv);
    }
    off.assign((size_t)nl + 2, 0);
    for (int64_t s = 0; s < n; ++s) off[level[s] + 1]++;
    for (int32_t v = 1; v <= nl + 1; ++v) off[v] += off[v - 1];
    std::vector<int64_t> cur(off.begin(), off.end());
    order.resize((size_t)n);
    for (int64_t s = 0; s < n; ++s) order[(size_t)cur[level[s]]++] = s;
    // off[lv] .. off[lv+1] is level lv (1-based); off[0] == off[1] == 0
}

inline int ew_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ================================================================== RelMF
template <typename T>
struct RelDev {
    T *W, *H, *W0, *W1, *H0, *H1;
    const T *X;       // dense (U,I)
    const T *prop;    // (I)
    int K;
    int64_t I;
    T wd, clip;
    OptParams<T> opt;
};

// One draw: cell index -> (u, i) = (cell / I, cell % I)  (relmf.pyx:144-146)
// CellT = uint32_t while U*I < 2^32 (one word of the index stream per draw), uint64_t beyond (64-bit draws, rng.hip)
template <typename T, int R, bool PACKED, int OPT, bool HOG>
__device__ __forceinline__ double relmf_sample(const RelDev<T> &d, int64_t u, int64_t i, int lane) {
    constexpr int NS = opt_num_states(OPT);
    const int K = d.K;
    const T r = d.X[u * d.I + i], p = d.prop[i];
    const int64_t ou = u * K, oi = i * K;
    Row<T, R, PACKED> w, h, sw[NS ? NS : 1], sh[NS ? NS : 1];
    w.load(d.W + ou, K, lane);
    h.load(d.H + oi, K, lane);
    if constexpr (NS >= 1) { sw[0].load(d.W0 + ou, K, lane); sh[0].load(d.H0 + oi, K, lane); }
    if constexpr (NS >= 2) { sw[1].load(d.W1 + ou, K, lane); sh[1].load(d.H1 + oi, K, lane); }
    T py = 0, pl = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        py += w.v[q] * h.v[q];
        pl += w.v[q] * w.v[q] + h.v[q] * h.v[q];
    }
    const T y = wave_sum(py), l2 = wave_sum(pl);
    const T qq = r / (p >= d.clip ? p : d.clip);                                  // r / dmax(p, M)
    const double loss = (double)(qq * (1 - y) * (1 - y) + (1 - qq) * y * y + d.wd * l2);  // model.pyx:117
    const T c = qq * (1 - y) + (1 - qq) * (0 - y);                                // model.pyx:131-139 (no factor 2)
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const T wv = w.v[q], hv = h.v[q];
        const T gw = -(c * hv) + d.wd * wv;
        const T gh = -(c * wv) + d.wd * hv;
        T dummy = 0;
        opt_update<T, OPT, HOG>(d.opt, w.v[q], OPT >= 1 ? sw[0].v[q] : dummy, OPT == 2 ? sw[1].v[q] : dummy, gw);
        opt_update<T, OPT, HOG>(d.opt, h.v[q], OPT >= 1 ? sh[0].v[q] : dummy, OPT == 2 ? sh[1].v[q] : dummy, gh);
    }
    w.store(d.W + ou, K, lane);
    h.store(d.H + oi, K, lane);
    if constexpr (NS >= 1) { sw[0].store(d.W0 + ou, K, lane); sh[0].store(d.H0 + oi, K, lane); }
    if constexpr (NS >= 2) { sw[1].store(d.W1 + ou, K, lane); sh[1].store(d.H1 + oi, K, lane); }
    return loss;
}

template <typename T, int R, bool PACKED, int OPT, bool HOG, typename CellT>
__global__ __launch_bounds__(256) void relmf_kernel(RelDev<T> d, const CellT *__restrict__ cells, int64_t n,
                                                   double *__restrict__ loss_acc) {
    const int lane = lane_id();
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double loss_sum = 0.0;
    for (int64_t s = wave0; s < n; s += n_waves) {
        const CellT cell = cells[s];
        loss_sum += relmf_sample<T, R, PACKED, OPT, HOG>(d, (int64_t)(cell / (CellT)d.I), (int64_t)(cell % (CellT)d.I), lane);
    }
    if (lane == 0 && loss_sum != 0.0) atomicAdd(loss_acc, loss_sum);
}

// ------------------------------------------------------------------ EXACT: dataflow execution of the sequential order
// The two-row form of bpr_ticket_kernel (bpr.hip).  Sample s touches row a of table A and row b of table B; it
// may run as soon as the earlier samples touching those rows are done.  The host numbers the accesses of every row in
// sequential order (sample s is access ka[s] of its A row, kb[s] of its B row); a wavefront takes the next sample of the order
// from a dispenser, waits until the two per-row counters show its turn numbers, loads the rows behind an agent-scope acquire
// fence, updates and stores them, and bumps the counters behind a release fence.  Samples are handed out in order to whichever
// wavefront asks next, so the smallest unfinished sample is always held by a RUNNING wavefront: no deadlock whatever part of the
// grid is resident; a spin limit turns a broken schedule into an error instead of a hang.  Replaces one launch per level
// (RelMF 30 x 40: 40 launches per epoch; GloVe on 120 words: ~300): the same arithmetic in the same order, bit for bit.
constexpr unsigned int TICKET2_SPIN_LIMIT = 1u << 20;

__device__ __forceinline__ bool ticket_wait2(const unsigned int *c0, unsigned int t0, const unsigned int *c1, unsigned int t1) {
    unsigned int spins = 0;
    while (true) {
        const unsigned int v0 = __hip_atomic_load(c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int v1 = __hip_atomic_load(c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v0 == t0 && v1 == t1) return true;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > TICKET2_SPIN_LIMIT) return false;
    }
}

// Body: double operator()(int64_t s, int lane, int64_t &a, int64_t &b) -- reports the sample's rows, then (after the wait)
// run(s, lane) does the loads, the update and the stores
template <typename Sample>
__device__ __forceinline__ void ticket2_loop(Sample smp, int64_t n, const uint32_t *__restrict__ ka, const uint32_t *__restrict__ kb,
                                             unsigned int *doneA, unsigned int *doneB, unsigned long long *next,
                                             double *__restrict__ loss_acc, int *err) {
    const int lane = lane_id();
    double loss_sum = 0.0;
    auto grab = [&]() -> int64_t {
        unsigned long long v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
        return (int64_t)(((unsigned long long)hi << 32) | lo);
    };
    int64_t s = grab();
    while (s < n) {
        const int64_t s_next = grab();   // asked for early: its latency hides behind this sample's wait and loads
        int64_t a, b;
        smp.rows(s, a, b);
        if (!ticket_wait2(doneA + a, ka[s], doneB + b, kb[s])) {   // cannot happen with a consistent schedule; never hang the device on a bad one
            if (lane == 0) atomicExch(err, 1);
            break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        loss_sum += smp.run(s, a, b, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // every lane's stores are complete before lane 0 publishes the turns
        if (lane == 0) {
            __hip_atomic_fetch_add(doneA + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(doneB + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s = s_next;
    }
    if (lane == 0 && loss_sum != 0.0) atomicAdd(loss_acc, loss_sum);
}

template <typename T, int R, bool PACKED, int OPT>
struct RelmfTicketSample {
    RelDev<T> d;
    const uint32_t *cells;
    __device__ __forceinline__ void rows(int64_t s, int64_t &u, int64_t &i) const {
        const uint32_t cell = cells[s];
        u = (int64_t)(cell / (uint32_t)d.I);
        i = (int64_t)(cell % (uint32_t)d.I);
    }
    __device__ __forceinline__ double run(int64_t, int64_t u, int64_t i, int lane) const { return relmf_sample<T, R, PACKED, OPT, false>(d, u, i, lane); }
};

template <typename T, int R, bool PACKED, int OPT>
__global__ __launch_bounds__(256) void relmf_ticket_kernel(RelDev<T> d, const uint32_t *__restrict__ cells, int64_t n,
                                                          const uint32_t *__restrict__ ku, const uint32_t *__restrict__ ki,
                                                          unsigned int *doneW, unsigned int *doneH, unsigned long long *next,
                                                          double *__restrict__ loss_acc, int *err) {
    ticket2_loop(RelmfTicketSample<T, R, PACKED, OPT>{d, cells}, n, ku, ki, doneW, doneH, next, loss_acc, err);
}

// any K (cymf/relmf.pyx:42 takes any num_components): rows wider than the register layouts (K > 256) are streamed from
// memory in two passes, lanes striding over k -- pass 1 the prediction and the l2 term, pass 2 the element-wise update.
template <typename T, int OPT, bool HOG, typename CellT>
__global__ __launch_bounds__(256) void relmf_wide_kernel(RelDev<T> d, const CellT *__restrict__ cells, int64_t n,
                                                        double *__restrict__ loss_acc) {
    const int lane = lane_id();
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int K = d.K;
    double loss_sum = 0.0;
    for (int64_t s = wave0; s < n; s += n_waves) {
        const CellT cell = cells[s];
        const int64_t u = (int64_t)(cell / (CellT)d.I), i = (int64_t)(cell % (CellT)d.I);
        const T r = d.X[u * d.I + i], p = d.prop[i];
        const int64_t ou = u * K, oi = i * K;
        T py = 0, pl = 0;
        for (int k = lane; k < K; k += 64) {
            const T wv = d.W[ou + k], hv = d.H[oi + k];
            py += wv * hv;
            pl += wv * wv + hv * hv;
        }
        const T y = wave_sum(py), l2 = wave_sum(pl);
        const T qq = r / (p >= d.clip ? p : d.clip);
        loss_sum += (double)(qq * (1 - y) * (1 - y) + (1 - qq) * y * y + d.wd * l2);
        const T c = qq * (1 - y) + (1 - qq) * (0 - y);
        for (int k = lane; k < K; k += 64) {
            T wv = d.W[ou + k], hv = d.H[oi + k];
            const T gw = -(c * hv) + d.wd * wv;
            const T gh = -(c * wv) + d.wd * hv;
            T w0 = 0, w1 = 0, h0 = 0, h1 = 0;
            if constexpr (OPT >= 1) { w0 = d.W0[ou + k]; h0 = d.H0[oi + k]; }
            if constexpr (OPT == 2) { w1 = d.W1[ou + k]; h1 = d.H1[oi + k]; }
            opt_update<T, OPT, HOG>(d.opt, wv, w0, w1, gw);
            opt_update<T, OPT, HOG>(d.opt, hv, h0, h1, gh);
            d.W[ou + k] = wv;
            d.H[oi + k] = hv;
            if constexpr (OPT >= 1) { d.W0[ou + k] = w0; d.H0[oi + k] = h0; }
            if constexpr (OPT == 2) { d.W1[ou + k] = w1; d.H1[oi + k] = h1; }
        }
    }
    if (lane == 0 && loss_sum != 0.0) atomicAdd(loss_acc, loss_sum);
}

// ================================================================== GloVe
template <typename T>
struct GloveDev {
    T *W, *H, *bW, *bH;       // central / context factors and biases
    T *aW, *aH, *abW, *abH;   // AdaGrad accumulators (init ones, optimizer.pyx:96-99)
    int K;
    T lr, x_max, alpha;
};

__device__ __forceinline__ float fpow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double fpow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ float flog(float a) { return logf(a); }
__device__ __forceinline__ double flog(double a) { return log(a); }

template <typename T, int R, bool PACKED>
__device__ __forceinline__ double glove_sample(const GloveDev<T> &d, int64_t c, int64_t x, T cnt, int lane) {
    const int K = d.K;
    const int64_t oc = c * K, ox = x * K;
    Row<T, R, PACKED> w, h, aw, ah;
    w.load(d.W + oc, K, lane);
    h.load(d.H + ox, K, lane);
    aw.load(d.aW + oc, K, lane);
    ah.load(d.aH + ox, K, lane);
    T bw = d.bW[c], bh = d.bH[x], abw = d.abW[c], abh = d.abH[x];
    T pd = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) pd += w.v[q] * h.v[q];
    T diff = wave_sum(pd);                                      // model.pyx:174-175
    diff += bw + bh;                                            // :176
    diff -= flog(cnt);                                          // :177
    const T tmp = diff;
    const T f = fpow(cnt / d.x_max, d.alpha);
    diff *= f < (T)1 ? f : (T)1;                                // :179, weight_func :34-35
    const double loss = (double)((T)0.5 * diff * tmp);          // :180
    const T g2 = diff * diff;
    // biases: AdaGrad-updated K times per sample (model.pyx:195-204):
    //   acc_k = acc_0 + k g^2,  b_K = b_0 - lr g sum_{k=1..K} 1/sqrt(acc_k); lanes share the k's
    T pbw = 0, pbh = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int k = Row<T, R, PACKED>::kof(lane, q);
        if (k < K) {
            pbw += (T)1 / fsqrt(abw + (T)(k + 1) * g2);
            pbh += (T)1 / fsqrt(abh + (T)(k + 1) * g2);
        }
        const T wv = w.v[q], hv = h.v[q];
        const T gw = diff * hv, gh = diff * wv;
        aw.v[q] += gw * gw;
        w.v[q] -= d.lr * gw / fsqrt(aw.v[q]);
        ah.v[q] += gh * gh;
        h.v[q] -= d.lr * gh / fsqrt(ah.v[q]);
    }
    const T sbw = wave_sum(pbw), sbh = wave_sum(pbh);
    w.store(d.W + oc, K, lane);
    h.store(d.H + ox, K, lane);
    aw.store(d.aW + oc, K, lane);
    ah.store(d.aH + ox, K, lane);
    if (lane == 0) {
        d.bW[c] = bw - d.lr * diff * sbw;
        d.bH[x] = bh - d.lr * diff * sbh;
        d.abW[c] = abw + (T)K * g2;
        d.abH[x] = abh + (T)K * g2;
    }
    return loss;
}

template <typename T, int R, bool PACKED>
__global__ __launch_bounds__(256) void glove_kernel(GloveDev<T> d, const int32_t *__restrict__ central,
                                                   const int32_t *__restrict__ context,
                                                   const T *__restrict__ counts, int64_t n,
                                                   double *__restrict__ loss_acc) {
    const int lane = lane_id();
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double loss_sum = 0.0;
    for (int64_t s = wave0; s < n; s += n_waves) loss_sum += glove_sample<T, R, PACKED>(d, central[s], context[s], counts[s], lane);
    if (lane == 0 && loss_sum != 0.0) atomicAdd(loss_acc, loss_sum);
}

// EXACT: the pairs in their given order, executed as a dataflow (ticket2_loop above); a word's bias and accumulator travel with
// its row (central word: bW, abW; context word: bH, abH), so the two row counters order them as well
template <typename T, int R, bool PACKED>
struct GloveTicketSample {
    GloveDev<T> d;
    const int32_t *central, *context;
    const T *counts;
    __device__ __forceinline__ void rows(int64_t s, int64_t &c, int64_t &x) const { c = central[s]; x = context[s]; }
    __device__ __forceinline__ double run(int64_t s, int64_t c, int64_t x, int lane) const { return glove_sample<T, R, PACKED>(d, c, x, counts[s], lane); }
};

template <typename T, int R, bool PACKED>
__global__ __launch_bounds__(256) void glove_ticket_kernel(GloveDev<T> d, const int32_t *__restrict__ central, const int32_t *__restrict__ context,
                                                          const T *__restrict__ counts, int64_t n, const uint32_t *__restrict__ kc,
                                                          const uint32_t *__restrict__ kx, unsigned int *doneC, unsigned int *doneX,
                                                          unsigned long long *next, double *__restrict__ loss_acc, int *err) {
    ticket2_loop(GloveTicketSample<T, R, PACKED>{d, central, context, counts}, n, kc, kx, doneC, doneX, next, loss_acc, err);
}

// any K (cymf/glove.pyx:57): the two-pass form of glove_kernel for rows wider than the register layouts (K > 256)
template <typename T>
__global__ __launch_bounds__(256) void glove_wide_kernel(GloveDev<T> d, const int32_t *__restrict__ central,
                                                        const int32_t *__restrict__ context,
                                                        const T *__restrict__ counts, int64_t n,
                                                        double *__restrict__ loss_acc) {
    const int lane = lane_id();
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int K = d.K;
    double loss_sum = 0.0;
    for (int64_t s = wave0; s < n; s += n_waves) {
        const int64_t c = central[s], x = context[s];
        const T cnt = counts[s];
        const int64_t oc = c * K, ox = x * K;
        const T bw = d.bW[c], bh = d.bH[x], abw = d.abW[c], abh = d.abH[x];
        T pd = 0;
        for (int k = lane; k < K; k += 64) pd += d.W[oc + k] * d.H[ox + k];
        T diff = wave_sum(pd);
        diff += bw + bh;
        diff -= flog(cnt);
        const T tmp = diff;
        const T f = fpow(cnt / d.x_max, d.alpha);
        diff *= f < (T)1 ? f : (T)1;
        loss_sum += (double)((T)0.5 * diff * tmp);
        const T g2 = diff * diff;
        T pbw = 0, pbh = 0;
        for (int k = lane; k < K; k += 64) {
            pbw += (T)1 / fsqrt(abw + (T)(k + 1) * g2);   // the K-fold bias update in closed form (cymf/model.pyx:195-204)
            pbh += (T)1 / fsqrt(abh + (T)(k + 1) * g2);
            T wv = d.W[oc + k], hv = d.H[ox + k], aw = d.aW[oc + k], ah = d.aH[ox + k];
            const T gw = diff * hv, gh = diff * wv;
            aw += gw * gw;
            wv -= d.lr * gw / fsqrt(aw);
            ah += gh * gh;
            hv -= d.lr * gh / fsqrt(ah);
            d.W[oc + k] = wv; d.H[ox + k] = hv; d.aW[oc + k] = aw; d.aH[ox + k] = ah;
        }
        const T sbw = wave_sum(pbw), sbh = wave_sum(pbh);
        if (lane == 0) {
            d.bW[c] = bw - d.lr * diff * sbw;
            d.bH[x] = bh - d.lr * diff * sbh;
            d.abW[c] = abw + (T)K * g2;
            d.abH[x] = abh + (T)K * g2;
        }
    }
    if (lane == 0 && loss_sum != 0.0) atomicAdd(loss_acc, loss_sum);
}

// ================================================================== GloVe THROUGHPUT: central-bucketed step kernel
// Same construction as bpr_step_kernel (bpr.hip), with the central word in the role of the positive item:
//   * pairs sorted by central word; wavefront w walks its own contiguous range of 64-pair chunks;
//   * the central word's row, its AdaGrad accumulator row and its {bias, bias accumulator} pair live
//     in registers across the run; at a chunk boundary the wave exchanges them with the other waves of
//     the same run through RETURNING float atomics of its deltas (AdaGrad accumulators are sums of g^2,
//     so adding deltas is exact for them), at a word switch it adds the deltas and loads the next word;
//   * context rows (parameter + accumulator) and bias pairs are gathered PF pairs ahead in a ring of
//     2*PF entries, updated in place and written back; a HOT context word (bit 30 of the context index)
//     gets atomic deltas instead, so concurrent waves cannot undo each other's updates on such rows.
// Biases are kept interleaved {bias, accumulator} per word (one 8-byte access).
struct GloveStepDev {
    float *W, *H, *aW, *aH;
    float2 *bW2, *bH2;
    int K;
    float lr, x_max, alpha;
};

template <typename RowT, int R>
__device__ __forceinline__ void gl_settle(RowT &a) {
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(a.v[r]));
}
template <typename RowT, int R>
__device__ __forceinline__ void gl_atomic_add_row(float *__restrict__ dst, const RowT &a, const RowT &b, int K, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = RowT::kof(lane, r);
        if (RowT::live(lane, r, K)) atomicAdd(dst + k, a.v[r] - b.v[r]);
    }
}
template <typename RowT, int R>
__device__ __forceinline__ void gl_exchange_row(float *__restrict__ dst, RowT &cur, RowT &base, int K, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = RowT::kof(lane, r);
        if (RowT::live(lane, r, K)) {
            const float dlt = cur.v[r] - base.v[r];
            const float found = atomicAdd(dst + k, dlt);
            cur.v[r] = found + dlt;
            base.v[r] = cur.v[r];
        }
    }
}

template <int R, bool PACKED, int PF>
__global__ __launch_bounds__(256) void glove_step_kernel(GloveStepDev d, const int32_t *__restrict__ central,
                                                        const int32_t *__restrict__ context,
                                                        const float *__restrict__ counts, int64_t n,
                                                        int64_t chunks_per_wave, double *__restrict__ loss_acc) {
    using RowT = Row<float, R, PACKED>;
    constexpr int RING = 2 * PF;
    static_assert(64 % RING == 0, "ring must divide the chunk");
    const int lane = lane_id();
    const int K = d.K;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t s_begin = wave * chunks_per_wave * 64;
    const int64_t s_end = s_begin + chunks_per_wave * 64 < n ? s_begin + chunks_per_wave * 64 : n;
    if (s_begin >= s_end) return;
    const int64_t c_end = (s_end - s_begin + 63) >> 6;

    auto load_meta = [&](int64_t c, int32_t &ce, int32_t &cx, float &lg, float &fw) {
        const int64_t my = s_begin + (c << 6) + lane;
        const bool in = c < c_end && my < s_end;
        ce = in ? central[my] : -1;
        cx = in ? context[my] : 0;
        const float cnt = in ? counts[my] : 1.0f;
        lg = logf(cnt);                                           // model.pyx:177
        const float f = powf(cnt / d.x_max, d.alpha);             // weight_func, model.pyx:34-35
        fw = f < 1.0f ? f : 1.0f;
    };
    int32_t ce_c, cx_c, ce_n, cx_n;
    float lg_c, fw_c, lg_n, fw_n;
    load_meta(0, ce_c, cx_c, lg_c, fw_c);
    load_meta(1, ce_n, cx_n, lg_n, fw_n);

    RowT hq[RING], ahq[RING];
    float2 bhq[RING];
    auto issue = [&](int e, int32_t cx) {
        const int64_t ox = (int64_t)(cx & 0x3fffffff) * K;
        hq[e].load(d.H + ox, K, lane);
        ahq[e].load(d.aH + ox, K, lane);
        bhq[e] = d.bH2[cx & 0x3fffffff];                          // wave-uniform address
    };
#pragma unroll
    for (int e = 0; e < PF; ++e) issue(e, bcast_lane(cx_c, e));

    int cur = -1;
    RowT w, w0, aw, aw0;
    w.fill(0.0f); w0.fill(0.0f); aw.fill(0.0f); aw0.fill(0.0f);
    float2 bw = make_float2(0.0f, 0.0f), bw0 = bw;
    float loss_sum = 0.0f;

    auto flush_central = [&]() {   // add this wave's deltas of the finished run
        gl_atomic_add_row<RowT, R>(d.W + (int64_t)cur * K, w, w0, K, lane);
        gl_atomic_add_row<RowT, R>(d.aW + (int64_t)cur * K, aw, aw0, K, lane);
        if (lane == 0) {
            atomicAdd(&d.bW2[cur].x, bw.x - bw0.x);
            atomicAdd(&d.bW2[cur].y, bw.y - bw0.y);
        }
    };

    for (int64_t c = 0; c < c_end; ++c) {
#pragma unroll 1
        for (int t0 = 0; t0 < 64; t0 += RING) {
#pragma unroll
            for (int e = 0; e < RING; ++e) {
                const int t = t0 + e;
                const int ce = bcast_lane(ce_c, t), cxraw = bcast_lane(cx_c, t);
                const int cx = cxraw & 0x3fffffff;
                const bool hot = (cxraw >> 30) & 1;
                const float lg = __builtin_bit_cast(float, bcast_lane(__builtin_bit_cast(int, lg_c), t));
                const float fw = __builtin_bit_cast(float, bcast_lane(__builtin_bit_cast(int, fw_c), t));
                if (ce >= 0) {                                     // wave-uniform
                    if (ce != cur) {                               // rare: next central word
                        if (cur >= 0) flush_central();
                        cur = ce;
                        w.load(d.W + (int64_t)ce * K, K, lane);
                        aw.load(d.aW + (int64_t)ce * K, K, lane);
                        bw = d.bW2[ce];
                        gl_settle<RowT, R>(w);
                        gl_settle<RowT, R>(aw);
                        asm volatile("" : "+v"(bw.x), "+v"(bw.y));
                        w0 = w; aw0 = aw; bw0 = bw;
                    }
                    const RowT h_old = hq[e], ah_old = ahq[e];
                    const float2 bh_old = bhq[e];
                    float pd = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) pd += w.v[r] * hq[e].v[r];
                    float diff = wave_sum(pd);                     // model.pyx:174-175
                    diff += bw.x + bhq[e].x;                       // :176
                    diff -= lg;                                    // :177
                    const float tmp = diff;
                    diff *= fw;                                    // :179
                    loss_sum += 0.5f * diff * tmp;                 // :180
                    const float g2 = diff * diff;
                    float pbw = 0, pbh = 0;                        // K-fold bias update, closed form (see glove_kernel)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int k = RowT::kof(lane, r);
                        if (RowT::live(lane, r, K)) {   // lanes past K hold zeros (also in the accumulators): keep 0/sqrt(0) out of w
                            pbw += __frsqrt_rn(bw.y + (float)(k + 1) * g2);     // v_rsq_f32: the step kernel is VALU-heavy (4 roots and
                            pbh += __frsqrt_rn(bhq[e].y + (float)(k + 1) * g2); // 4 divisions per element and pair with sqrt + div)
                            const float wv = w.v[r], hv = hq[e].v[r];
                            const float gw = diff * hv, gh = diff * wv;
                            aw.v[r] += gw * gw;
                            w.v[r] -= d.lr * gw * __frsqrt_rn(aw.v[r]);
                            ahq[e].v[r] += gh * gh;
                            hq[e].v[r] -= d.lr * gh * __frsqrt_rn(ahq[e].v[r]);
                        }
                    }
                    const float sbw = wave_sum(pbw), sbh = wave_sum(pbh);
                    bw.x -= d.lr * diff * sbw;
                    bw.y += (float)K * g2;
                    bhq[e].x -= d.lr * diff * sbh;
                    bhq[e].y += (float)K * g2;
                    if (hot) {
                        gl_atomic_add_row<RowT, R>(d.H + (int64_t)cx * K, hq[e], h_old, K, lane);
                        gl_atomic_add_row<RowT, R>(d.aH + (int64_t)cx * K, ahq[e], ah_old, K, lane);
                        if (lane == 0) {
                            atomicAdd(&d.bH2[cx].x, bhq[e].x - bh_old.x);
                            atomicAdd(&d.bH2[cx].y, bhq[e].y - bh_old.y);
                        }
                    } else {
                        hq[e].store(d.H + (int64_t)cx * K, K, lane);
                        ahq[e].store(d.aH + (int64_t)cx * K, K, lane);
                        if (lane == 0) d.bH2[cx] = bhq[e];
                    }
                }
                const int tn = t + PF;
                issue((e + PF) % RING, tn < 64 ? bcast_lane(cx_c, tn & 63) : bcast_lane(cx_n, tn & 63));
            }
        }
        // chunk boundary: exchange the open central word's state with the waves that share its run
        if (cur >= 0 && c + 1 < c_end) {
            gl_exchange_row<RowT, R>(d.W + (int64_t)cur * K, w, w0, K, lane);
            gl_exchange_row<RowT, R>(d.aW + (int64_t)cur * K, aw, aw0, K, lane);
            float fx = 0.0f, fy = 0.0f;
            if (lane == 0) {
                fx = atomicAdd(&d.bW2[cur].x, bw.x - bw0.x);
                fy = atomicAdd(&d.bW2[cur].y, bw.y - bw0.y);
            }
            fx = __builtin_bit_cast(float, bcast_lane(__builtin_bit_cast(int, fx), 0));
            fy = __builtin_bit_cast(float, bcast_lane(__builtin_bit_cast(int, fy), 0));
            bw.x = fx + (bw.x - bw0.x);
            bw.y = fy + (bw.y - bw0.y);
            bw0 = bw;
            gl_settle<RowT, R>(w); gl_settle<RowT, R>(w0); gl_settle<RowT, R>(aw); gl_settle<RowT, R>(aw0);
            asm volatile("" : "+v"(bw.x), "+v"(bw.y), "+v"(bw0.x), "+v"(bw0.y));
        }
        ce_c = ce_n; cx_c = cx_n; lg_c = lg_n; fw_c = fw_n;
        load_meta(c + 2, ce_n, cx_n, lg_n, fw_n);
    }
    if (cur >= 0) flush_central();
    if (lane == 0) atomicAdd(loss_acc, (double)loss_sum);
}

__global__ void pack2_kernel(const float *__restrict__ a, const float *__restrict__ b, float2 *__restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = make_float2(a[i], b[i]);
}
__global__ void unpack2_kernel(const float2 *__restrict__ in, float *__restrict__ a, float *__restrict__ b, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { a[i] = in[i].x; b[i] = in[i].y; }
}

// ---------------------------------------------------------------- RelMF throughput path (f32)
// The epoch's U*I cells are redrawn every epoch (relmf.pyx:142-146), so the bucketing that the BPR and
// GloVe step kernels get once from the host is done on the device per epoch: a counting sort of the
// cells by user (histogram, scan, scatter of the item ids).  One wavefront then owns whole users: W[u]
// (and its optimizer state) stays in registers for the user's ~I draws, the row of X it needs (I floats)
// stays in cache, and the item rows come through a ring of loads issued PF slots ahead.  Every item is
// equally popular here (uniform cells): with a few thousand wavefronts in flight each item row has
// several holders at any time, so a slot ADDS its delta of H[i] (float atomics) instead of storing the
// row back; AdaGrad's accumulator of H[i] is a sum of g^2 and is added to the same way, Adam's moments are
// stored plainly (a lost update there perturbs a running average, a delta-sum of moments is unstable: see bpr.hip).
__global__ void relmf_hist_kernel(const uint32_t *__restrict__ cells, int64_t n, uint32_t I, uint32_t *__restrict__ cnt) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) atomicAdd(cnt + cells[t] / I, 1u);
}

// exclusive scan of U counts -> off[U+1] (int64) and the scatter cursors; one workgroup
__global__ __launch_bounds__(1024) void relmf_scan_kernel(const uint32_t *__restrict__ cnt, int32_t U, int64_t *__restrict__ off,
                                                         unsigned long long *__restrict__ cursor) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int32_t per = (U + 1023) / 1024;
    const int32_t b = tid * per, e = b + per < U ? b + per : U;
    int64_t sum = 0;
    for (int32_t u = b; u < e; ++u) sum += cnt[u];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int q = 0; q < 1024; ++q) { const int64_t v = part[q]; part[q] = run; run += v; }
        off[U] = run;
    }
    __syncthreads();
    int64_t run = part[tid];
    for (int32_t u = b; u < e; ++u) { off[u] = run; cursor[u] = (unsigned long long)run; run += cnt[u]; }
}

__global__ void relmf_scatter_kernel(const uint32_t *__restrict__ cells, int64_t n, uint32_t I,
                                     unsigned long long *__restrict__ cursor, int32_t *__restrict__ items) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const uint32_t c = cells[t], u = c / I;
        items[atomicAdd(cursor + u, 1ull)] = (int32_t)(c - u * I);
    }
}

// Bucketing with workgroup-private counters in LDS (U <= RELMF_LDS_USERS): a workgroup takes a contiguous segment
// of < 65536 cells, counts it into packed 16-bit LDS counters (two users per word), and touches the global
// counters once per (workgroup, user) instead of once per cell -- the global-atomic version spends 18 ms of a
// 160 M-cell epoch on its 320 M contended atomics.  The scatter reserves the workgroup's range of every user
// it holds with one global atomic and ranks its own cells with returning LDS atomics.
constexpr int RELMF_LDS_USERS = 24576;        // 4 U (bases) + 2 U (counters) bytes of LDS <= 144 KB
constexpr int RELMF_SEG = 49152;              // cells per workgroup (< 65536: a 16-bit counter cannot overflow)

__global__ __launch_bounds__(1024) void relmf_hist_lds_kernel(const uint32_t *__restrict__ cells, int64_t n, uint32_t I, int32_t U,
                                                             uint32_t *__restrict__ cnt) {
    extern __shared__ uint32_t sm[];           // [ceil(U/2)] packed counters
    const int tid = threadIdx.x;
    const int words = (U + 1) >> 1;
    for (int k = tid; k < words; k += 1024) sm[k] = 0u;
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * RELMF_SEG, e = b + RELMF_SEG < n ? b + RELMF_SEG : n;
    for (int64_t t = b + tid; t < e; t += 1024) {
        const uint32_t u = cells[t] / I;
        atomicAdd(sm + (u >> 1), 1u << (16 * (u & 1)));
    }
    __syncthreads();
    for (int u = tid; u < U; u += 1024) {
        const uint32_t c = (sm[u >> 1] >> (16 * (u & 1))) & 0xffffu;
        if (c) atomicAdd(cnt + u, c);
    }
}

__global__ __launch_bounds__(1024) void relmf_scatter_lds_kernel(const uint32_t *__restrict__ cells, int64_t n, uint32_t I, int32_t U,
                                                                unsigned long long *__restrict__ cursor,
                                                                int32_t *__restrict__ items) {
    extern __shared__ uint32_t sm[];           // [U] bases, then [ceil(U/2)] packed counters
    uint32_t *base = sm;
    uint32_t *cnt2 = sm + U;
    const int tid = threadIdx.x;
    const int words = (U + 1) >> 1;
    for (int k = tid; k < words; k += 1024) cnt2[k] = 0u;
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * RELMF_SEG, e = b + RELMF_SEG < n ? b + RELMF_SEG : n;
    for (int64_t t = b + tid; t < e; t += 1024) {
        const uint32_t u = cells[t] / I;
        atomicAdd(cnt2 + (u >> 1), 1u << (16 * (u & 1)));
    }
    __syncthreads();
    for (int u = tid; u < U; u += 1024) {
        const uint32_t c = (cnt2[u >> 1] >> (16 * (u & 1))) & 0xffffu;
        base[u] = c ? (uint32_t)atomicAdd(cursor + u, (unsigned long long)c) : 0u;   // (N < 2^32: the stream's range)
    }
    __syncthreads();
    for (int k = tid; k < words; k += 1024) cnt2[k] = 0u;
    __syncthreads();
    for (int64_t t = b + tid; t < e; t += 1024) {
        const uint32_t c = cells[t], u = c / I;
        const uint32_t sh = 16 * (u & 1);
        const uint32_t r = (atomicAdd(cnt2 + (u >> 1), 1u << sh) >> sh) & 0xffffu;
        items[(int64_t)base[u] + r] = (int32_t)(c - u * I);
    }
}

struct RelStepDev {
    float *W, *H, *W0, *W1, *H0, *H1;
    const float *X, *prop;
    int K;
    int64_t I;
    float wd, clip;
    OptParams<float> opt;
    int coherent;   // read the item rows with agent-scope atomic loads (CYMF_RELMF_COHERENT_LOADS, default 1)
    int32_t u_lo, u_hi;         // users of this launch (multi-GPU: the rank's range; else 0 .. U)
    int64_t turn_lo, turn_hi;   // turns of this launch (multi-GPU sub-steps; else 0 .. open end)
};

template <int R, bool PACKED, int OPT, int PF>
__global__ __launch_bounds__(256) void relmf_step_kernel(RelStepDev d, const int64_t *__restrict__ off,
                                                        const int32_t *__restrict__ items, int32_t U,
                                                        int32_t users_per_wave, double *__restrict__ loss_acc, int *err) {
    using RowT = Row<float, R, PACKED>;
    constexpr int NS = opt_num_states(OPT);
    constexpr int NSA = NS ? NS : 1;
    constexpr int RING = 2 * PF;
    constexpr float SFILL = OPT == CYMF_OPT_ADAGRAD ? 1.0f : 0.0f;   // masked lanes of optimizer-state rows (rows.h: Row::load)
    static_assert(64 % RING == 0, "ring must divide the chunk");
    const int lane = lane_id();
    const int K = d.K;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t u_begin = d.u_lo + wave * users_per_wave;
    const int64_t u_end = u_begin + users_per_wave < d.u_hi ? u_begin + users_per_wave : d.u_hi;
    float *const Ws[2] = {d.W0, d.W1};
    float *const Hs[2] = {d.H0, d.H1};
    float loss_sum = 0.0f;
    // The wavefront's users take turns of TURN slots each instead of being finished one after the other: a user's
    // ~I draws in one block would let the item side (and AdaGrad's accumulators most of all) see the users in
    // a few long blocks, far from the random interleaving of the sequential order (measured: norm of H x2.2 with
    // nine users per wavefront back to back, 0.97 with every user on its own wavefront).
    constexpr int64_t TURN = 256;
    for (int64_t turn0 = d.turn_lo * TURN; turn0 < d.turn_hi * TURN; turn0 += TURN) {
    bool any = false;
    for (int64_t u = u_begin; u < u_end; ++u) {
        const int64_t s_begin = off[u] + turn0;
        const int64_t s_end = off[u + 1] < s_begin + TURN ? off[u + 1] : s_begin + TURN;
        if (s_begin >= s_end) continue;
        any = true;
        const int64_t c_end = (s_end - s_begin + 63) >> 6;
        const float *xrow = d.X + u * d.I;
        RowT w, sw[NSA];
        w.load(d.W + u * K, K, lane);
#pragma unroll
        for (int q = 0; q < NS; ++q) sw[q].load(Ws[q] + u * K, K, lane, SFILL);
        auto load_meta = [&](int64_t c) -> int32_t {
            const int64_t my = s_begin + (c << 6) + lane;
            int32_t it = (c < c_end && my < s_end) ? items[my] : -1;
            if (it >= (int32_t)d.I) { atomicExch(err, 1); it = -1; }   // a broken bucketing must not become a wild access
            return it;
        };
        int32_t it_c = load_meta(0), it_n = load_meta(1);
        RowT hq[RING], shq[RING][NSA];
        float xq[RING], pq[RING];
        auto issue = [&](int e, int32_t i) {
            const int64_t oi = (int64_t)(i < 0 ? 0 : i) * K;
            if (d.coherent) hq[e].load_coherent(d.H + oi, K, lane); else hq[e].load(d.H + oi, K, lane);
#pragma unroll
            for (int q = 0; q < NS; ++q) shq[e][q].load(Hs[q] + oi, K, lane, SFILL);
            xq[e] = xrow[i < 0 ? 0 : i];                            // wave-uniform addresses
            pq[e] = d.prop[i < 0 ? 0 : i];
        };
#pragma unroll
        for (int e = 0; e < PF; ++e) issue(e, bcast_lane(it_c, e));
        for (int64_t c = 0; c < c_end; ++c) {
#pragma unroll 1
            for (int t0 = 0; t0 < 64; t0 += RING) {
#pragma unroll
                for (int e = 0; e < RING; ++e) {
                    const int t = t0 + e;
                    const int32_t i = bcast_lane(it_c, t);
                    if (i >= 0) {                                   // wave-uniform
                        const RowT h_old = hq[e];
                        const RowT acc_old = shq[e][0];             // (AdaGrad: the accumulator before this slot)
                        float py = 0, pl = 0;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            py += w.v[r] * hq[e].v[r];
                            pl += w.v[r] * w.v[r] + hq[e].v[r] * hq[e].v[r];
                        }
                        const float y = wave_sum(py), l2 = wave_sum(pl);
                        const float p = pq[e];
                        const float qq = xq[e] / (p >= d.clip ? p : d.clip);                 // r / dmax(p, M)
                        loss_sum += qq * (1 - y) * (1 - y) + (1 - qq) * y * y + d.wd * l2;   // model.pyx:117
                        const float cc = qq * (1 - y) + (1 - qq) * (0 - y);                  // model.pyx:131-139
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float wv = w.v[r], hv = hq[e].v[r];
                            const float gw = -(cc * hv) + d.wd * wv;
                            const float gh = -(cc * wv) + d.wd * hv;
                            float dummy = 0;
                            if (RowT::live(lane, r, K)) {   // masked lanes hold zeros: keep 0/sqrt(0) out
                                opt_update<float, OPT, true>(d.opt, w.v[r], OPT >= 1 ? sw[0].v[r] : dummy, OPT == 2 ? sw[NSA - 1].v[r] : dummy, gw);
                                opt_update<float, OPT, true>(d.opt, hq[e].v[r], OPT >= 1 ? shq[e][0].v[r] : dummy,
                                                             OPT == 2 ? shq[e][NSA - 1].v[r] : dummy, gh);
                            }
                        }
                        gl_atomic_add_row<RowT, R>(d.H + (int64_t)i * K, hq[e], h_old, K, lane);
                        if constexpr (OPT == CYMF_OPT_ADAGRAD) {      // a sum of g^2: adding this slot's share is exact
                            gl_atomic_add_row<RowT, R>(Hs[0] + (int64_t)i * K, shq[e][0], acc_old, K, lane);
                        } else {
#pragma unroll
                            for (int q = 0; q < NS; ++q) shq[e][q].store(Hs[q] + (int64_t)i * K, K, lane);
                        }
                    }
                    const int tn = t + PF;
                    issue((e + PF) % RING, tn < 64 ? bcast_lane(it_c, tn & 63) : bcast_lane(it_n, tn & 63));
                }
            }
            it_c = it_n;
            it_n = load_meta(c + 2);
        }
        w.store(d.W + u * K, K, lane);                              // this wave is the only writer of W[u]
#pragma unroll
        for (int q = 0; q < NS; ++q) sw[q].store(Ws[q] + u * K, K, lane);
    }
    if (!any && d.turn_hi >= ((int64_t)1 << 40)) break;   // open-ended launch: stop when every user of the wave is through
    }
    if (lane == 0) atomicAdd(loss_acc, (double)loss_sum);
}

// Multi-GPU RelMF: users sharded (every rank generates the whole cell stream and buckets it, then works through its own
// users), item table replicated.  After each sub-step (a range of turns) the deltas of H -- and of AdaGrad's accumulators,
// which are plain sums -- are all-reduced; H gets the sequentialisation factor (uniform here: every item is touched about
// U / steps times per sub-step), Adam's moments stay private to the rank.
__global__ void relmf_delta_kernel(const float *__restrict__ H, const float *__restrict__ A, const float *__restrict__ sH,
                                   const float *__restrict__ sA, float *__restrict__ D, int64_t n, int with_acc) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        D[i] = H[i] - sH[i];
        if (with_acc) D[n + i] = A[i] - sA[i];
    }
}
__global__ void relmf_apply_kernel(float *__restrict__ H, float *__restrict__ A, float *__restrict__ sH, float *__restrict__ sA,
                                   const float *__restrict__ D, float scale, int64_t n, int with_acc) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = sH[i] + scale * D[i];
        H[i] = v; sH[i] = v;
        if (with_acc) { const float a = sA[i] + D[n + i]; A[i] = a; sA[i] = a; }
    }
}

template <int R, bool PACKED>
void launch_relmf_step_opt(int opt, const RelStepDev &d, const int64_t *off, const int32_t *items, int32_t U, int32_t upw,
                           double *loss, int *err, int grid, hipStream_t s) {
    switch (opt) {
    case CYMF_OPT_SGD: hipLaunchKernelGGL((relmf_step_kernel<R, PACKED, CYMF_OPT_SGD, 8>), dim3(grid), dim3(256), 0, s, d, off, items, U, upw, loss, err); break;
    case CYMF_OPT_ADAGRAD: hipLaunchKernelGGL((relmf_step_kernel<R, PACKED, CYMF_OPT_ADAGRAD, 4>), dim3(grid), dim3(256), 0, s, d, off, items, U, upw, loss, err); break;
    default: hipLaunchKernelGGL((relmf_step_kernel<R, PACKED, CYMF_OPT_ADAM, 4>), dim3(grid), dim3(256), 0, s, d, off, items, U, upw, loss, err); break;
    }
}

void launch_relmf_step(int K, int opt, const RelStepDev &d, const int64_t *off, const int32_t *items, int32_t U, int32_t upw,
                       double *loss, int *err, int grid, hipStream_t s) {
    if (K <= 64) launch_relmf_step_opt<1, false>(opt, d, off, items, U, upw, loss, err, grid, s);
    else if (K == 128) launch_relmf_step_opt<2, true>(opt, d, off, items, U, upw, loss, err, grid, s);
    else launch_relmf_step_opt<2, false>(opt, d, off, items, U, upw, loss, err, grid, s);
}

// Multi-GPU GloVe: pairs are sharded by CENTRAL word, the context table is replicated.  After a step
// every rank forms the deltas of its replica against the last synchronised state -- context rows, their AdaGrad
// accumulators, {context bias, its accumulator} -- in one buffer that is all-reduced; the accumulators are plain sums
// of g^2, so their summed deltas ARE the sequential accumulators; rows and biases get the sequentialisation factor
// of the item-delta exchange of bpr.hip (build_step_counts / delta_scale_kernel there; rho = lr/5 per touch: AdaGrad, no weight decay).
__global__ void glove_delta_kernel(const float *__restrict__ H, const float *__restrict__ aH, const float2 *__restrict__ b2,
                                   const float *__restrict__ sH, const float *__restrict__ sA, const float2 *__restrict__ sB,
                                   float *__restrict__ D, int64_t VK, int64_t V) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < VK; i += stride) {
        D[i] = H[i] - sH[i];
        D[VK + i] = aH[i] - sA[i];
        if (i < V) {
            D[2 * VK + 2 * i] = b2[i].x - sB[i].x;
            D[2 * VK + 2 * i + 1] = b2[i].y - sB[i].y;
        }
    }
}
__global__ void glove_apply_kernel(float *__restrict__ H, float *__restrict__ aH, float2 *__restrict__ b2, float *__restrict__ sH,
                                   float *__restrict__ sA, float2 *__restrict__ sB, const float *__restrict__ D,
                                   const float *__restrict__ scale, int K, int64_t VK, int64_t V) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < VK; i += stride) {
        const float v = sH[i] + scale[i / K] * D[i];
        H[i] = v; sH[i] = v;
        const float a = sA[i] + D[VK + i];
        aH[i] = a; sA[i] = a;
        if (i < V) {
            const float2 nb = make_float2(sB[i].x + scale[i] * D[2 * VK + 2 * i], sB[i].y + D[2 * VK + 2 * i + 1]);
            b2[i] = nb; sB[i] = nb;
        }
    }
}

void launch_glove_step(int K, const GloveStepDev &d, const int32_t *c, const int32_t *x, const float *cnt, int64_t n,
                       int64_t cpw, double *loss, int grid, hipStream_t s) {
#define CALL_(R_, P_) hipLaunchKernelGGL((glove_step_kernel<R_, P_, 8>), dim3(grid), dim3(256), 0, s, d, c, x, cnt, n, cpw, loss)
    if (K <= 64) { CALL_(1, false); } else if (K == 128) { CALL_(2, true); } else { CALL_(2, false); }
#undef CALL_
}

#define CYMF_DISPATCH_LAYOUT(K, CALL)                                                          \
    do {                                                                                       \
        const int R__ = ((K) + 63) / 64;                                                       \
        const bool P__ = (K) == 128 || (K) == 256;                                             \
        if (R__ == 1) { CALL(1, false); }                                                      \
        else if (R__ == 2) { if (P__) { CALL(2, true); } else { CALL(2, false); } }            \
        else if (R__ == 3) { CALL(3, false); }                                                 \
        else { if (P__) { CALL(4, true); } else { CALL(4, false); } }                          \
    } while (0)

template <typename T, int R, bool PACKED, bool HOG, typename CellT>
void launch_relmf_opt(int opt, const RelDev<T> &d, const CellT *cells, int64_t n, double *loss, int grid,
                      hipStream_t s) {
    switch (opt) {
    case CYMF_OPT_SGD: hipLaunchKernelGGL((relmf_kernel<T, R, PACKED, CYMF_OPT_SGD, HOG, CellT>), dim3(grid), dim3(256), 0, s, d, cells, n, loss); break;
    case CYMF_OPT_ADAGRAD: hipLaunchKernelGGL((relmf_kernel<T, R, PACKED, CYMF_OPT_ADAGRAD, HOG, CellT>), dim3(grid), dim3(256), 0, s, d, cells, n, loss); break;
    default: hipLaunchKernelGGL((relmf_kernel<T, R, PACKED, CYMF_OPT_ADAM, HOG, CellT>), dim3(grid), dim3(256), 0, s, d, cells, n, loss); break;
    }
}

// hog: lock-free launch over all cells (Adam's moment clamp of rows.h on); false: one conflict-free level
template <typename T, typename CellT>
void launch_relmf(int K, int opt, const RelDev<T> &d, const CellT *cells, int64_t n, double *loss, int grid,
                  hipStream_t s, bool hog = false) {
    if (K > 256) {
#define WIDE_(O_)                                                                                                                   \
    do {                                                                                                                            \
        if (hog) hipLaunchKernelGGL((relmf_wide_kernel<T, O_, true, CellT>), dim3(grid), dim3(256), 0, s, d, cells, n, loss);       \
        else hipLaunchKernelGGL((relmf_wide_kernel<T, O_, false, CellT>), dim3(grid), dim3(256), 0, s, d, cells, n, loss);          \
    } while (0)
        if (opt == CYMF_OPT_SGD) WIDE_(CYMF_OPT_SGD); else if (opt == CYMF_OPT_ADAGRAD) WIDE_(CYMF_OPT_ADAGRAD); else WIDE_(CYMF_OPT_ADAM);
#undef WIDE_
        return;
    }
#define CALL_(R_, P_)                                                                 \
    do {                                                                              \
        if (hog) launch_relmf_opt<T, R_, P_, true, CellT>(opt, d, cells, n, loss, grid, s);  \
        else launch_relmf_opt<T, R_, P_, false, CellT>(opt, d, cells, n, loss, grid, s);     \
    } while (0)
    CYMF_DISPATCH_LAYOUT(K, CALL_);
#undef CALL_
}

// dataflow state of an exact-mode epoch: per-row counters of finished accesses (table A rows, then table B rows), the
// dispenser and the error flag
struct TicketState {
    DevBuf<uint32_t> ka, kb, done;
    DevBuf<unsigned long long> next;
    DevBuf<int> err;
    int n_cu = 0;
    int prepare(int64_t nA, int64_t nB, int device, hipStream_t s) {
        if (!n_cu) {
            CYMF_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
            if (n_cu <= 0) n_cu = 256;
        }
        CYMF_TRY(done.alloc((size_t)(nA + nB)));
        CYMF_TRY(next.alloc(1));
        CYMF_TRY(err.alloc(1));
        CYMF_TRY(done.zero(s));
        CYMF_TRY(next.zero(s));
        CYMF_TRY(err.zero(s));
        return 0;
    }
    int blocks(int64_t n) const { return (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)2 * n_cu, (n + 3) / 4)); }   // 8 waves per CU: a latency chain
    int check(hipStream_t s, const char *who) {
        int e = 0;
        CYMF_HIP(hipMemcpyAsync(&e, err.p, sizeof(int), hipMemcpyDeviceToHost, s));
        CYMF_HIP(hipStreamSynchronize(s));
        if (e) return fail(CYMF_ERR_HIP, "%s: a sample waited beyond the spin limit for its turn (inconsistent schedule)", who);
        return 0;
    }
};

// turn numbers: sample s is access ka[s] of row a[s] and access kb[s] of row b[s] in sequential order
template <typename IndexA, typename IndexB>
void number_turns(int64_t n, IndexA a_of, IndexB b_of, int64_t nA, int64_t nB, uint32_t *ka, uint32_t *kb) {
    std::vector<uint32_t> cA((size_t)nA, 0u), cB((size_t)nB, 0u);
    for (int64_t s = 0; s < n; ++s) {
        ka[(size_t)s] = cA[(size_t)a_of(s)]++;
        kb[(size_t)s] = cB[(size_t)b_of(s)]++;
    }
}
template <typename IndexA, typename IndexB>
void number_turns(int64_t n, IndexA a_of, IndexB b_of, int64_t nA, int64_t nB, std::vector<uint32_t> &ka, std::vector<uint32_t> &kb) {
    ka.resize((size_t)n);
    kb.resize((size_t)n);
    number_turns(n, a_of, b_of, nA, nB, ka.data(), kb.data());
}

template <typename T>
void launch_relmf_ticket(int K, int opt, const RelDev<T> &d, const uint32_t *cells, int64_t n, TicketState &t, const uint32_t *ka, const uint32_t *kb,
                         int32_t U, double *loss, hipStream_t s) {
#define OPT_(R_, P_, O_) hipLaunchKernelGGL((relmf_ticket_kernel<T, R_, P_, O_>), dim3(t.blocks(n)), dim3(256), 0, s, d, cells, n, ka, kb, \
                                            t.done.p, t.done.p + U, t.next.p, loss, t.err.p)
#define CALL_(R_, P_)                                                                              \
    do {                                                                                           \
        if (opt == CYMF_OPT_SGD) OPT_(R_, P_, CYMF_OPT_SGD);                                       \
        else if (opt == CYMF_OPT_ADAGRAD) OPT_(R_, P_, CYMF_OPT_ADAGRAD);                          \
        else OPT_(R_, P_, CYMF_OPT_ADAM);                                                          \
    } while (0)
    CYMF_DISPATCH_LAYOUT(K, CALL_);
#undef CALL_
#undef OPT_
}

template <typename T>
void launch_glove_ticket(int K, const GloveDev<T> &d, const int32_t *c, const int32_t *x, const T *cnt, int64_t n, TicketState &t, int32_t V,
                         double *loss, hipStream_t s) {
#define CALL_(R_, P_) hipLaunchKernelGGL((glove_ticket_kernel<T, R_, P_>), dim3(t.blocks(n)), dim3(256), 0, s, d, c, x, cnt, n, t.ka.p, t.kb.p, \
                                         t.done.p, t.done.p + V, t.next.p, loss, t.err.p)
    CYMF_DISPATCH_LAYOUT(K, CALL_);
#undef CALL_
}

template <typename T>
void launch_glove(int K, const GloveDev<T> &d, const int32_t *c, const int32_t *x, const T *cnt, int64_t n,
                  double *loss, int grid, hipStream_t s) {
    if (K > 256) {
        hipLaunchKernelGGL((glove_wide_kernel<T>), dim3(grid), dim3(256), 0, s, d, c, x, cnt, n, loss);
        return;
    }
#define CALL_(R_, P_) hipLaunchKernelGGL((glove_kernel<T, R_, P_>), dim3(grid), dim3(256), 0, s, d, c, x, cnt, n, loss)
    CYMF_DISPATCH_LAYOUT(K, CALL_);
#undef CALL_
}

__global__ void widen_cells_kernel(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// HOGWILD launches: the number of wavefronts bounds the staleness.  With waves striding over the
// samples, about waves * f_max of them hold the hottest row at once (f_max = its share of the
// samples), so waves <= hot_budget / f_max keeps that number at hot_budget; 8 waves per CU at most.
inline int hogwild_grid(int64_t n, double f_max) {
    const double hot_budget = 2.0;
    int64_t waves = f_max > 0 ? (int64_t)(hot_budget / f_max) : n;
    waves = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(waves, n), 256 * 8));
    return (int)((waves + 3) / 4);
}

}  // namespace
}  // namespace cymf

using namespace cymf;

struct cymf_comm;
namespace cymf {
int comm_allreduce_sum_f32(cymf_comm *c, float *d_buf, int64_t n, hipStream_t s);   // comm.hip
int comm_allgatherv(cymf_comm *c, void *d_buf, const int64_t *row_bounds, int64_t row_bytes, hipStream_t s);
int comm_world(cymf_comm *c);
int comm_rank(cymf_comm *c);
}  // namespace cymf


// =====================================================================================
//                                        RelMF
// =====================================================================================
template <typename T>
struct RelStore {
    DevBuf<T> W, H, W0, W1, H0, H1, X, prop;
};

struct cymf_relmf {
    int32_t U = 0, I = 0, K = 0;
    int opt = 0, dtype = 0, mode = 0, device = 0;
    double lr = 0, wd = 0, clip = 0;
    uint32_t seed = 1234;
    hipStream_t stream = nullptr;
    RelStore<float> f32;
    RelStore<double> f64;
    DeviceRng rng;
    DevBuf<uint32_t> d_cells, d_sorted, d_ucnt;
    // U*I >= 2^32 (cymf/relmf.pyx:128 draws on `long`): 64-bit cells, generated and consumed in batches by the generic kernels.
    // force_wide_cells (CYMF_RELMF_FORCE_WIDE_CELLS=1, test hook): run the 64-bit-cell kernels on a small problem's widened cells.
    bool wide = false, force_wide_cells = false;
    DevBuf<uint64_t> d_cells64;
    DevBuf<unsigned long long> d_ucursor;
    // step path (f32 throughput, K <= 128): the cells of epoch e+1 are generated and bucketed by user on
    // side_stream while relmf_step_kernel works through epoch e; buffers double-buffered by epoch parity
    bool step_path = false;
    // tile path (relmf_tiles.hip): the default lock-free mode on one GPU -- B x B stratified tiles, item rows in LDS, no
    // atomics on HBM; the user-bucketed step path above remains for the multi-GPU exchange (and CYMF_RELMF_NO_TILES=1)
    bool tile_ok = false;
    RelTilePlan plan;
    RelTileBufs tb;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_bucketed[2] = {nullptr, nullptr}, ev_step_done[2] = {nullptr, nullptr};
    DevBuf<int64_t> d_uoff[2];
    DevBuf<int32_t> d_items[2];
    int64_t epochs_prepared = 0, epoch_cursor = 0;
    // multi-GPU: user ranges of the ranks, sub-steps per epoch, exchange buffers
    cymf_comm *comm = nullptr;
    std::vector<int64_t> user_bounds;
    int32_t steps_per_epoch = 1;
    DevBuf<float> d_snapH, d_snapA, d_xdelta;
    DevBuf<double> d_loss;
    DevBuf<int> d_err;
    std::vector<uint32_t> h_cells;
    // exact mode: the epoch's draws (device and pinned host copy) and, for the dataflow launch, their turn numbers -- two sets by epoch
    // parity: epoch e + 1 is drawn, brought down, numbered and sent up on side_stream while the kernel of epoch e runs (relmf_exact_prepare)
    DevBuf<uint32_t> x_cells[2], x_ka[2], x_kb[2];
    PinnedBuf<uint32_t> p_cells[2], p_ka[2], p_kb[2];
    int64_t x_epoch[2] = {-1, -1};
    bool x_turns[2] = {false, false};
    hipEvent_t ev_x[2] = {nullptr, nullptr};
    TicketState ticket;                 // exact mode: dataflow execution (CYMF_RELMF_EXACT_LEVELS=1: one launch per level)
    bool have_data = false, have_params = false;
};

// cells of epoch g: generate on the side stream, bucket by user into the buffers of parity g & 1
static int relmf_prepare(cymf_relmf *h, int64_t g_want) {
    const int64_t N = (int64_t)h->U * h->I;
    while (h->epochs_prepared <= g_want) {
        const int64_t g = h->epochs_prepared;
        const int b = (int)(g & 1);
        hipStream_t ss = h->side_stream;
        CYMF_TRY(h->d_cells.alloc((size_t)N));
        CYMF_TRY(h->d_ucnt.alloc((size_t)h->U));
        CYMF_TRY(h->d_ucursor.alloc((size_t)h->U));
        CYMF_TRY(h->d_uoff[b].alloc((size_t)h->U + 1));
        CYMF_TRY(h->d_items[b].alloc((size_t)N));
        if (g >= 2) CYMF_HIP(hipStreamWaitEvent(ss, h->ev_step_done[b], 0));   // buffers b were read by the step kernel of epoch g-2
        CYMF_TRY(h->rng.generate(0, N, h->d_cells.p, ss));
        CYMF_TRY(h->d_ucnt.zero(ss));
        const bool lds = h->U <= RELMF_LDS_USERS && !(getenv("CYMF_RELMF_NO_LDS") && getenv("CYMF_RELMF_NO_LDS")[0] == '1');
        const int segs = (int)((N + RELMF_SEG - 1) / RELMF_SEG);
        if (lds) {
            const size_t sm_h = sizeof(uint32_t) * (size_t)((h->U + 1) / 2), sm_s = sizeof(uint32_t) * ((size_t)h->U + (size_t)((h->U + 1) / 2));
            CYMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(relmf_hist_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_h));
            CYMF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(relmf_scatter_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_s));
            hipLaunchKernelGGL(relmf_hist_lds_kernel, dim3(segs), dim3(1024), sm_h, ss, h->d_cells.p, N, (uint32_t)h->I, h->U, h->d_ucnt.p);
            hipLaunchKernelGGL(relmf_scan_kernel, dim3(1), dim3(1024), 0, ss, h->d_ucnt.p, h->U, h->d_uoff[b].p, h->d_ucursor.p);
            hipLaunchKernelGGL(relmf_scatter_lds_kernel, dim3(segs), dim3(1024), sm_s, ss, h->d_cells.p, N, (uint32_t)h->I, h->U,
                               h->d_ucursor.p, h->d_items[b].p);
        } else {
            hipLaunchKernelGGL(relmf_hist_kernel, dim3(ew_blocks(N)), dim3(256), 0, ss, h->d_cells.p, N, (uint32_t)h->I, h->d_ucnt.p);
            hipLaunchKernelGGL(relmf_scan_kernel, dim3(1), dim3(1024), 0, ss, h->d_ucnt.p, h->U, h->d_uoff[b].p, h->d_ucursor.p);
            hipLaunchKernelGGL(relmf_scatter_kernel, dim3(ew_blocks(N)), dim3(256), 0, ss, h->d_cells.p, N, (uint32_t)h->I, h->d_ucursor.p,
                               h->d_items[b].p);
        }
        CYMF_HIP(hipGetLastError());
        CYMF_HIP(hipEventRecord(h->ev_bucketed[b], ss));
        h->epochs_prepared++;
    }
    return 0;
}

// tile path: cells of epoch g generated on the side stream and grouped by tile into the buffers of parity g & 1
static int relmf_prepare_tiles(cymf_relmf *h, int64_t g_want) {
    const int64_t N = (int64_t)h->U * h->I;
    while (h->epochs_prepared <= g_want) {
        const int64_t g = h->epochs_prepared;
        const int b = (int)(g & 1);
        hipStream_t ss = h->side_stream;
        CYMF_TRY(h->d_cells.alloc((size_t)N));
        if (g >= 2) CYMF_HIP(hipStreamWaitEvent(ss, h->ev_step_done[b], 0));   // buffers b were read by the tile kernels of epoch g-2
        CYMF_TRY(h->rng.generate(0, N, h->d_cells.p, ss));
        CYMF_TRY(relmf_tile_bucket(h->plan, h->d_cells.p, h->tb, b, ss));
        CYMF_HIP(hipEventRecord(h->ev_bucketed[b], ss));
        h->epochs_prepared++;
    }
    return 0;
}

// Exact mode: the draws of epoch e on the device and (pinned) on the host and, for the dataflow launch, their turn numbers on the
// device; everything on side_stream, recorded in ev_x[e & 1].  Called for epoch e + 1 while the kernel of epoch e runs.
static int relmf_exact_prepare(cymf_relmf *h, int64_t e, bool turns) {
    const int b = (int)(e & 1);
    const int64_t N = (int64_t)h->U * h->I;
    hipStream_t ss = h->side_stream;
    if (h->x_epoch[b] != e) {
        CYMF_TRY(h->x_cells[b].alloc((size_t)N));
        CYMF_TRY(h->rng.generate(0, N, h->x_cells[b].p, ss));      // relmf.pyx:128: the epochs' draws in stream order
        CYMF_TRY(h->p_cells[b].reserve((size_t)N));
        CYMF_HIP(hipMemcpyAsync(h->p_cells[b].p, h->x_cells[b].p, (size_t)N * sizeof(uint32_t), hipMemcpyDeviceToHost, ss));
        CYMF_HIP(hipStreamSynchronize(ss));
        h->x_epoch[b] = e;
        h->x_turns[b] = false;
    }
    if (turns && !h->x_turns[b]) {
        CYMF_TRY(h->p_ka[b].reserve((size_t)N)); CYMF_TRY(h->p_kb[b].reserve((size_t)N));
        const uint32_t I32 = (uint32_t)h->I;
        const uint32_t *cells = h->p_cells[b].p;
        number_turns(N, [&](int64_t s) { return cells[(size_t)s] / I32; }, [&](int64_t s) { return cells[(size_t)s] % I32; },
                     h->U, h->I, h->p_ka[b].p, h->p_kb[b].p);
        CYMF_TRY(h->x_ka[b].reserve((size_t)N)); CYMF_TRY(h->x_kb[b].reserve((size_t)N));
        CYMF_HIP(hipMemcpyAsync(h->x_ka[b].p, h->p_ka[b].p, (size_t)N * sizeof(uint32_t), hipMemcpyHostToDevice, ss));
        CYMF_HIP(hipMemcpyAsync(h->x_kb[b].p, h->p_kb[b].p, (size_t)N * sizeof(uint32_t), hipMemcpyHostToDevice, ss));
        h->x_turns[b] = true;
    }
    CYMF_HIP(hipEventRecord(h->ev_x[b], ss));
    return 0;
}

template <typename T>
static int relmf_epoch(cymf_relmf *h, RelStore<T> &st, double *loss_out, bool next = false) {
    const int64_t N = (int64_t)h->U * h->I;   // relmf.pyx:120: one epoch = U*I draws with replacement
    const bool tiled = h->mode == CYMF_MODE_THROUGHPUT && h->tile_ok && !h->comm && sizeof(T) == 4;
    const bool stepped = tiled || (h->mode == CYMF_MODE_THROUGHPUT && h->step_path && sizeof(T) == 4);
    if (h->wide && h->mode != CYMF_MODE_THROUGHPUT)
        return fail(CYMF_ERR_UNSUPPORTED, "cymf_relmf: U*I = %lld draws per epoch: the exact (sequential-order) mode schedules the epoch's draws on the "
                    "host and is limited to U*I < 2^32; the lock-free mode takes any size", (long long)N);
    if (!stepped && !h->wide && h->mode != CYMF_MODE_EXACT) {   // (exact mode: relmf_exact_prepare)
        CYMF_TRY(h->d_cells.alloc((size_t)N));
        CYMF_TRY(h->rng.generate(0, N, h->d_cells.p, h->stream));
    }
    RelDev<T> d;
    d.W = st.W.p; d.H = st.H.p; d.W0 = st.W0.p; d.W1 = st.W1.p; d.H0 = st.H0.p; d.H1 = st.H1.p;
    d.X = st.X.p; d.prop = st.prop.p; d.K = h->K; d.I = h->I; d.wd = (T)h->wd; d.clip = (T)h->clip;
    d.opt = make_opt_params<T>(h->lr);
    CYMF_TRY(h->d_loss.zero(h->stream));
    if (tiled) {
        if constexpr (sizeof(T) == 4) {
            const int64_t e = h->epoch_cursor;
            const int b = (int)(e & 1);
            CYMF_TRY(relmf_prepare_tiles(h, e));
            CYMF_HIP(hipStreamWaitEvent(h->stream, h->ev_bucketed[b], 0));
            CYMF_TRY(relmf_prepare_tiles(h, e + 1));   // next epoch's cells and tiles, concurrently with this epoch's sub-steps
            RelTileParams tp;
            tp.W = st.W.p; tp.H = st.H.p; tp.W0 = st.W0.p; tp.W1 = st.W1.p; tp.H0 = st.H0.p; tp.H1 = st.H1.p;
            tp.X = st.X.p; tp.prop = st.prop.p; tp.wd = (float)h->wd; tp.clip = (float)h->clip;
            tp.opt = make_opt_params<float>(h->lr);
            CYMF_TRY(h->d_err.alloc(1));
            CYMF_TRY(h->d_err.zero(h->stream));
            CYMF_TRY(relmf_tile_epoch(h->plan, tp, h->tb, b, e, h->d_loss.p, h->d_err.p, h->stream));
            CYMF_HIP(hipEventRecord(h->ev_step_done[b], h->stream));
            h->epoch_cursor++;
        }
    } else if (h->mode == CYMF_MODE_THROUGHPUT && h->step_path) {
        if constexpr (sizeof(T) == 4) {
            const int64_t e = h->epoch_cursor;
            const int b = (int)(e & 1);
            CYMF_TRY(relmf_prepare(h, e));
            CYMF_HIP(hipStreamWaitEvent(h->stream, h->ev_bucketed[b], 0));
            CYMF_TRY(relmf_prepare(h, e + 1));   // next epoch's cells, concurrently with this epoch's updates
            RelStepDev sd;
            sd.W = st.W.p; sd.H = st.H.p; sd.W0 = st.W0.p; sd.W1 = st.W1.p; sd.H0 = st.H0.p; sd.H1 = st.H1.p;
            sd.X = st.X.p; sd.prop = st.prop.p; sd.K = h->K; sd.I = h->I; sd.wd = (float)h->wd; sd.clip = (float)h->clip;
            sd.opt = make_opt_params<float>(h->lr);
            sd.coherent = !(getenv("CYMF_RELMF_COHERENT_LOADS") && getenv("CYMF_RELMF_COHERENT_LOADS")[0] == '0');
            sd.u_lo = 0; sd.u_hi = h->U; sd.turn_lo = 0; sd.turn_hi = (int64_t)1 << 40;
            if (h->comm) { sd.u_lo = (int32_t)h->user_bounds[comm_rank(h->comm)]; sd.u_hi = (int32_t)h->user_bounds[comm_rank(h->comm) + 1]; }
            const int32_t my_users = sd.u_hi - sd.u_lo;
            // whole users per wavefront; every user has about I draws, so equal user counts are equal work.
            // Staleness bound, as for the other lock-free launches (hogwild_grid): every wavefront holds up to
            // RING item rows at a time and the cells are uniform over the I items, so about waves * RING / I
            // wavefronts hold any one item row at once; their deltas are all computed from the same stale row and
            // add up.  Four holders measured fastest on 20000 x 8000 (36 ms/epoch; 49 ms with two, 38 ms with eight
            // or more) and track the sequential oracle as closely as two.
            const int ring = h->opt == CYMF_OPT_SGD ? 16 : 8;
            const int holders = getenv("CYMF_RELMF_HOLDERS") ? std::max(1, atoi(getenv("CYMF_RELMF_HOLDERS"))) : 4;
            const int64_t max_waves = std::max<int64_t>(64, std::min<int64_t>(256 * 12, (int64_t)holders * h->I / ring));
            const int32_t upw = (int32_t)std::max<int64_t>(1, ((int64_t)my_users + max_waves - 1) / max_waves);
            const int64_t waves = std::max<int64_t>(1, ((int64_t)my_users + upw - 1) / upw);
            CYMF_TRY(h->d_err.alloc(1));
            CYMF_TRY(h->d_err.zero(h->stream));
            if (!h->comm) {
                launch_relmf_step(h->K, h->opt, sd, h->d_uoff[b].p, h->d_items[b].p, h->U, upw, h->d_loss.p, h->d_err.p, (int)((waves + 3) / 4), h->stream);
                CYMF_HIP(hipGetLastError());
            } else {
                // sub-steps = ranges of turns (a user has about I draws: Poisson, so I + 6 sqrt(I) covers practically all; the
                // last sub-step is open-ended and takes the stragglers), each followed by the exchange of the item deltas
                const int32_t S = std::max(1, h->steps_per_epoch);
                const int64_t turns = ((int64_t)(h->I + 6.0 * std::sqrt((double)h->I)) + 255) / 256 + 1;
                const int world = comm_world(h->comm);
                const int64_t nH = (int64_t)h->I * h->K;
                const int with_acc = h->opt == CYMF_OPT_ADAGRAD ? 1 : 0;
                double rho = 2.0 * h->lr * h->wd + 0.2 * h->lr;
                if (h->opt == CYMF_OPT_ADAM) rho = 5.0 * h->lr;
                const double a = std::pow(1.0 - std::min(0.5, rho), ((double)h->U / S) / world);   // U / S touches of an item per sub-step
                const float scale = (float)(a < 1.0 - 1e-12 ? (1.0 - std::pow(a, world)) / (world * (1.0 - a)) : 1.0);
                for (int32_t q = 0; q < S; ++q) {
                    sd.turn_lo = turns * q / S;
                    sd.turn_hi = q + 1 < S ? turns * (q + 1) / S : (int64_t)1 << 40;
                    if (my_users > 0 && sd.turn_lo < sd.turn_hi) {
                        launch_relmf_step(h->K, h->opt, sd, h->d_uoff[b].p, h->d_items[b].p, h->U, upw, h->d_loss.p, h->d_err.p, (int)((waves + 3) / 4), h->stream);
                        CYMF_HIP(hipGetLastError());
                    }
                    hipLaunchKernelGGL(relmf_delta_kernel, dim3(ew_blocks(nH)), dim3(256), 0, h->stream, st.H.p ? reinterpret_cast<const float *>(st.H.p) : nullptr,
                                       reinterpret_cast<const float *>(st.H0.p), h->d_snapH.p, h->d_snapA.p, h->d_xdelta.p, nH, with_acc);
                    CYMF_HIP(hipGetLastError());
                    CYMF_TRY(comm_allreduce_sum_f32(h->comm, h->d_xdelta.p, nH * (1 + with_acc), h->stream));
                    hipLaunchKernelGGL(relmf_apply_kernel, dim3(ew_blocks(nH)), dim3(256), 0, h->stream, reinterpret_cast<float *>(st.H.p),
                                       reinterpret_cast<float *>(st.H0.p), h->d_snapH.p, h->d_snapA.p, h->d_xdelta.p, scale, nH, with_acc);
                    CYMF_HIP(hipGetLastError());
                }
            }
            CYMF_HIP(hipEventRecord(h->ev_step_done[b], h->stream));
            h->epoch_cursor++;
        }
    } else if (h->mode == CYMF_MODE_THROUGHPUT && h->wide) {
        // U*I >= 2^32: 64-bit draws (the one-lane walker of rng.hip), generated and consumed in batches of 2^26
        const int64_t BATCH = (int64_t)1 << 26;
        CYMF_TRY(h->d_cells64.alloc((size_t)BATCH));
        for (int64_t done = 0; done < N; done += BATCH) {
            const int64_t nb = std::min(BATCH, N - done);
            CYMF_TRY(h->rng.generate64(0, nb, h->d_cells64.p, h->stream));
            launch_relmf<T, uint64_t>(h->K, h->opt, d, h->d_cells64.p, nb, h->d_loss.p, hogwild_grid(nb, 1.0 / std::min(h->U, h->I)), h->stream, /*hog=*/true);
            CYMF_HIP(hipGetLastError());
        }
    } else if (h->mode == CYMF_MODE_THROUGHPUT) {
        if (h->force_wide_cells) {   // test hook: the 64-bit-cell kernels on this problem's (widened) 32-bit draws
            CYMF_TRY(h->d_cells64.alloc((size_t)N));
            hipLaunchKernelGGL(widen_cells_kernel, dim3(ew_blocks(N)), dim3(256), 0, h->stream, h->d_cells.p, h->d_cells64.p, N);
            launch_relmf<T, uint64_t>(h->K, h->opt, d, h->d_cells64.p, N, h->d_loss.p, hogwild_grid(N, 1.0 / std::min(h->U, h->I)), h->stream, /*hog=*/true);
        } else {
            launch_relmf<T, uint32_t>(h->K, h->opt, d, h->d_cells.p, N, h->d_loss.p, hogwild_grid(N, 1.0 / std::min(h->U, h->I)), h->stream, /*hog=*/true);
        }
        CYMF_HIP(hipGetLastError());
    } else {
        const char *lv_env = getenv("CYMF_RELMF_EXACT_LEVELS");   // (read per epoch: the tests switch it inside one process)
        const bool by_levels = lv_env && lv_env[0] == '1';
        const bool flow = !by_levels && h->K <= 256 && !h->force_wide_cells;
        const int64_t e = h->epoch_cursor;
        const int xb = (int)(e & 1);
        CYMF_TRY(relmf_exact_prepare(h, e, flow));     // (made under the previous epoch's kernel, except for a call's first epoch)
        CYMF_HIP(hipStreamWaitEvent(h->stream, h->ev_x[xb], 0));
        h->epoch_cursor++;
        if (flow) {
            // the epoch's draws in their own order, as ONE dataflow launch (relmf_ticket_kernel): the host only numbers the accesses
            CYMF_TRY(h->ticket.prepare(h->U, h->I, h->device, h->stream));
            launch_relmf_ticket<T>(h->K, h->opt, d, h->x_cells[xb].p, N, h->ticket, h->x_ka[xb].p, h->x_kb[xb].p, h->U, h->d_loss.p, h->stream);
            CYMF_HIP(hipGetLastError());
            if (next) CYMF_TRY(relmf_exact_prepare(h, e + 1, true));   // the host's share of epoch e + 1, under the kernel of epoch e
            CYMF_TRY(h->ticket.check(h->stream, "relmf exact mode"));
            double loss_t = 0;
            CYMF_HIP(hipMemcpyAsync(&loss_t, h->d_loss.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
            CYMF_HIP(hipStreamSynchronize(h->stream));
            if (loss_out) *loss_out = loss_t;
            return 0;
        }
        h->h_cells.assign(h->p_cells[xb].p, h->p_cells[xb].p + N);
        std::vector<int32_t> su((size_t)N), si((size_t)N);
        for (int64_t s = 0; s < N; ++s) {
            su[s] = (int32_t)(h->h_cells[s] / (uint32_t)h->I);
            si[s] = (int32_t)(h->h_cells[s] % (uint32_t)h->I);
        }
        std::vector<int64_t> order, off;
        level_schedule(N, su.data(), si.data(), h->U, h->I, order, off);
        std::vector<uint32_t> sorted((size_t)N);
        for (int64_t p = 0; p < N; ++p) sorted[p] = h->h_cells[(size_t)order[p]];
        CYMF_TRY(h->d_sorted.upload(sorted.data(), sorted.size(), h->stream));
        if (h->force_wide_cells) {
            CYMF_TRY(h->d_cells64.alloc((size_t)N));
            hipLaunchKernelGGL(widen_cells_kernel, dim3(ew_blocks(N)), dim3(256), 0, h->stream, h->d_sorted.p, h->d_cells64.p, N);
        }
        for (size_t lv = 1; lv + 1 < off.size(); ++lv) {
            const int64_t b = off[lv], n = off[lv + 1] - off[lv];
            if (n <= 0) continue;
            if (h->force_wide_cells) launch_relmf<T, uint64_t>(h->K, h->opt, d, h->d_cells64.p + b, n, h->d_loss.p, (int)((n + 3) / 4), h->stream);
            else launch_relmf<T, uint32_t>(h->K, h->opt, d, h->d_sorted.p + b, n, h->d_loss.p, (int)((n + 3) / 4), h->stream);
        }
        CYMF_HIP(hipGetLastError());
    }
    double loss = 0;
    int err = 0;
    if (stepped) CYMF_HIP(hipMemcpyAsync(&err, h->d_err.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    CYMF_HIP(hipMemcpyAsync(&loss, h->d_loss.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    if (err == 2) return fail(CYMF_ERR_HIP, "relmf: a cell was drawn more than 255 times inside one tile of one epoch (that draw was not applied)");
    if (err == 3) return fail(CYMF_ERR_HIP, "relmf: an item factor is not finite, or grew more than fourfold inside one tile visit (the tile schedule keeps item rows in block floating point); "
                                            "CYMF_RELMF_NO_TILES=1 runs the float kernels");
    if (err) return fail(CYMF_ERR_HIP, "relmf: the per-epoch bucketing produced an index outside its block");
    if (loss_out) *loss_out = loss;
    return 0;
}

extern "C" int cymf_relmf_create(cymf_relmf **out, int32_t U, int32_t I, int32_t K, int optimizer,
                                 double learning_rate, double weight_decay, double clip_value, uint32_t seed,
                                 int dtype, int mode, int device) {
    if (!out) return fail(CYMF_ERR_INVALID, "cymf_relmf_create: out is NULL");
    *out = nullptr;
    if (U <= 0 || I <= 0 || K <= 0) return fail(CYMF_ERR_INVALID, "cymf_relmf_create: U, I, K must be positive");
    if (optimizer < 0 || optimizer > 2 || (dtype != CYMF_F32 && dtype != CYMF_F64) ||
        (mode != CYMF_MODE_EXACT && mode != CYMF_MODE_THROUGHPUT))
        return fail(CYMF_ERR_INVALID, "cymf_relmf_create: bad optimizer/dtype/mode");
    CYMF_TRY(use_device(device));
    cymf_relmf *h = new cymf_relmf();
    h->U = U; h->I = I; h->K = K; h->opt = optimizer; h->lr = learning_rate; h->wd = weight_decay;
    h->clip = clip_value; h->seed = seed; h->dtype = dtype; h->mode = mode; h->device = device;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(CYMF_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    if (mode == CYMF_MODE_THROUGHPUT) {   // rows are updated from all XCDs inside one kernel: see bpr.hip
        for (DevBuf<float> *b : {&h->f32.W, &h->f32.H, &h->f32.W0, &h->f32.W1, &h->f32.H0, &h->f32.H1}) b->fine = 2;
        for (DevBuf<double> *b : {&h->f64.W, &h->f64.H, &h->f64.W0, &h->f64.W1, &h->f64.H0, &h->f64.H1}) b->fine = 2;
    }
    int rc = h->d_loss.alloc(1);
    h->wide = (uint64_t)U * (uint64_t)I > 0xffffffffull;   // 64-bit draws (cymf/relmf.pyx:128 draws on long)
    h->force_wide_cells = getenv("CYMF_RELMF_FORCE_WIDE_CELLS") && getenv("CYMF_RELMF_FORCE_WIDE_CELLS")[0] == '1';
    h->step_path = !h->wide && mode == CYMF_MODE_THROUGHPUT && dtype == CYMF_F32 && K <= 128 && !(getenv("CYMF_RELMF_NO_STEP") && getenv("CYMF_RELMF_NO_STEP")[0] == '1');
    if (h->force_wide_cells) h->step_path = false;
    h->tile_ok = !h->force_wide_cells && mode == CYMF_MODE_THROUGHPUT && dtype == CYMF_F32 && relmf_tile_plan(U, I, K, optimizer, &h->plan) &&
                 !(getenv("CYMF_RELMF_NO_TILES") && getenv("CYMF_RELMF_NO_TILES")[0] == '1');
    const bool exact_side = mode == CYMF_MODE_EXACT && !h->wide;   // the exact mode draws and schedules the next epoch beside the running one
    if (!rc && exact_side) {
        hipError_t e2 = create_side_stream(&h->side_stream);
        for (int b = 0; b < 2 && e2 == hipSuccess; ++b) e2 = hipEventCreateWithFlags(&h->ev_x[b], hipEventDisableTiming);
        if (e2 != hipSuccess) rc = fail(CYMF_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e2));
    }
    if (!rc && (h->step_path || h->tile_ok)) {
        hipError_t e2 = create_side_stream(&h->side_stream);
        for (int b = 0; b < 2 && e2 == hipSuccess; ++b) {
            e2 = hipEventCreateWithFlags(&h->ev_bucketed[b], hipEventDisableTiming);
            if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&h->ev_step_done[b], hipEventDisableTiming);
        }
        if (e2 != hipSuccess) rc = fail(CYMF_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e2));
    }
    // relmf.pyx:128; >= 2M cells per epoch: chunked jump-ahead generator (rng.hip), else the one-workgroup walker.
    // The generator lives on the stream that consumes it: the side stream on the step path.
    if (!rc) rc = h->rng.init(seed, (uint64_t)U * (uint64_t)I, (h->step_path || h->tile_ok || exact_side) ? h->side_stream : h->stream,
                              /*parallel=*/(int64_t)U * I >= (int64_t)2 << 20);
    if (rc) { (void)cymf_relmf_destroy(h); return rc; }
    *out = h;
    return 0;
}

extern "C" int cymf_relmf_set_data(cymf_relmf *h, const double *X, const double *propensities) {
    if (!h || !X || !propensities) return fail(CYMF_ERR_INVALID, "cymf_relmf_set_data: bad arguments");
    CYMF_TRY(use_device(h->device));
    const size_t n = (size_t)h->U * h->I;
    if (h->dtype == CYMF_F32) { CYMF_TRY(upload_f64(h->f32.X, X, n, h->stream)); CYMF_TRY(upload_f64(h->f32.prop, propensities, (size_t)h->I, h->stream)); }
    else { CYMF_TRY(upload_f64(h->f64.X, X, n, h->stream)); CYMF_TRY(upload_f64(h->f64.prop, propensities, (size_t)h->I, h->stream)); }
    CYMF_HIP(hipStreamSynchronize(h->stream));
    h->have_data = true;
    return 0;
}

template <typename T>
static int relmf_upload(cymf_relmf *h, RelStore<T> &st, const double *W, const double *H) {
    const size_t nW = (size_t)h->U * h->K, nH = (size_t)h->I * h->K;
    CYMF_TRY(upload_f64(st.W, W, nW, h->stream));
    CYMF_TRY(upload_f64(st.H, H, nH, h->stream));
    if (h->opt == CYMF_OPT_ADAGRAD) {
        CYMF_TRY(fill_dev<T>(st.W0, nW, (T)1, h->stream));
        CYMF_TRY(fill_dev<T>(st.H0, nH, (T)1, h->stream));
    } else if (h->opt == CYMF_OPT_ADAM) {
        CYMF_TRY(fill_dev<T>(st.W0, nW, (T)0, h->stream)); CYMF_TRY(fill_dev<T>(st.W1, nW, (T)0, h->stream));
        CYMF_TRY(fill_dev<T>(st.H0, nH, (T)0, h->stream)); CYMF_TRY(fill_dev<T>(st.H1, nH, (T)0, h->stream));
    }
    CYMF_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cymf_relmf_upload(cymf_relmf *h, const double *W, const double *H) {
    if (!h || !W || !H) return fail(CYMF_ERR_INVALID, "cymf_relmf_upload: bad arguments");
    CYMF_TRY(use_device(h->device));
    if (h->dtype == CYMF_F32) CYMF_TRY(relmf_upload(h, h->f32, W, H)); else CYMF_TRY(relmf_upload(h, h->f64, W, H));
    if (h->comm) {
        if (!h->step_path) return fail(CYMF_ERR_UNSUPPORTED, "cymf_relmf: a communicator needs the float32 throughput step path (K <= 128)");
        const size_t nH = (size_t)h->I * h->K;
        const bool acc = h->opt == CYMF_OPT_ADAGRAD;
        CYMF_TRY(h->d_snapH.alloc(nH));
        CYMF_TRY(h->d_xdelta.alloc(nH * (acc ? 2 : 1)));
        CYMF_HIP(hipMemcpyAsync(h->d_snapH.p, h->f32.H.p, nH * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        if (acc) {
            CYMF_TRY(h->d_snapA.alloc(nH));
            CYMF_HIP(hipMemcpyAsync(h->d_snapA.p, h->f32.H0.p, nH * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        }
        CYMF_HIP(hipStreamSynchronize(h->stream));
    }
    h->have_params = true;
    return 0;
}

extern "C" int cymf_relmf_download(cymf_relmf *h, double *W, double *H) {
    if (!h || !W || !H || !h->have_params) return fail(CYMF_ERR_INVALID, "cymf_relmf_download: bad arguments / no params");
    CYMF_TRY(use_device(h->device));
    const size_t nW = (size_t)h->U * h->K, nH = (size_t)h->I * h->K;
    if (h->comm && h->dtype == CYMF_F32)   // every rank returns all user rows
        CYMF_TRY(comm_allgatherv(h->comm, h->f32.W.p, h->user_bounds.data(), (int64_t)h->K * (int64_t)sizeof(float), h->stream));
    if (h->dtype == CYMF_F32) { CYMF_TRY(download_f64(h->f32.W, W, nW, h->stream)); CYMF_TRY(download_f64(h->f32.H, H, nH, h->stream)); }
    else { CYMF_TRY(download_f64(h->f64.W, W, nW, h->stream)); CYMF_TRY(download_f64(h->f64.H, H, nH, h->stream)); }
    return 0;
}

extern "C" int cymf_relmf_set_steps_per_epoch(cymf_relmf *h, int32_t steps) {
    if (!h || steps < 1) return fail(CYMF_ERR_INVALID, "cymf_relmf_set_steps_per_epoch: bad arguments");
    h->steps_per_epoch = steps;
    return 0;
}

extern "C" int cymf_relmf_attach_comm(cymf_relmf *h, cymf_comm *c, const int64_t *user_bounds) {
    if (!h || !c || !user_bounds) return fail(CYMF_ERR_INVALID, "cymf_relmf_attach_comm: bad arguments");
    if (h->have_params) return fail(CYMF_ERR_INVALID, "cymf_relmf_attach_comm must precede cymf_relmf_upload");
    const int world = comm_world(c);
    if (user_bounds[0] != 0 || user_bounds[world] != h->U) return fail(CYMF_ERR_INVALID, "cymf_relmf_attach_comm: bounds must run from 0 to U");
    for (int r = 0; r < world; ++r)
        if (user_bounds[r] > user_bounds[r + 1]) return fail(CYMF_ERR_INVALID, "cymf_relmf_attach_comm: bounds not monotone");
    h->comm = c;
    h->user_bounds.assign(user_bounds, user_bounds + world + 1);
    return 0;
}

extern "C" int cymf_relmf_epochs(cymf_relmf *h, int32_t n_epochs, double *loss_out) {
    if (!h || n_epochs < 0) return fail(CYMF_ERR_INVALID, "cymf_relmf_epochs: bad arguments");
    if (!h->have_data || !h->have_params) return fail(CYMF_ERR_INVALID, "cymf_relmf_epochs before set_data/upload");
    CYMF_TRY(use_device(h->device));
    for (int32_t e = 0; e < n_epochs; ++e) {
        double *lo = loss_out ? loss_out + e : nullptr;
        const bool next = e + 1 < n_epochs;   // (exact mode: the following epoch is prepared under this epoch's kernel)
        if (h->dtype == CYMF_F32) CYMF_TRY(relmf_epoch<float>(h, h->f32, lo, next)); else CYMF_TRY(relmf_epoch<double>(h, h->f64, lo, next));
    }
    return 0;
}

extern "C" int cymf_relmf_destroy(cymf_relmf *h) {
    if (!h) return 0;
    if (!cymf::runtime_alive(h->device)) return 0;   // process exit / runtime already torn down: leak quietly
    if (h->side_stream) (void)hipStreamSynchronize(h->side_stream);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    for (int b = 0; b < 2; ++b) {
        if (h->ev_bucketed[b]) (void)hipEventDestroy(h->ev_bucketed[b]);
        if (h->ev_step_done[b]) (void)hipEventDestroy(h->ev_step_done[b]);
        if (h->ev_x[b]) (void)hipEventDestroy(h->ev_x[b]);
    }
    delete h;
    return 0;
}

// =====================================================================================
//                                        GloVe
// =====================================================================================
template <typename T>
struct GloveStore {
    DevBuf<T> W, H, bW, bH, aW, aH, abW, abH, counts;
};

struct cymf_glove {
    DevBuf<float2> bW2, bH2;        // throughput f32: {bias, accumulator} interleaved per word
    DevBuf<float> d_counts_f32;     // throughput f32: counts in central-sorted order
    bool step_path = false;         // throughput f32, K <= 128: central-bucketed step kernel
    int64_t step_waves = 1;
    int32_t V = 0, Vc = 0, K = 0;
    int dtype = 0, mode = 0, device = 0;
    double lr = 0, x_max = 0, alpha = 0;
    hipStream_t stream = nullptr;
    GloveStore<float> f32;
    GloveStore<double> f64;
    int64_t N = 0;
    DevBuf<int32_t> d_central, d_context;   // EXACT by levels: level order; otherwise the given order
    std::vector<int64_t> level_off;
    TicketState ticket;                     // exact mode: dataflow execution of the given order (CYMF_GLOVE_EXACT_LEVELS=1: by levels)
    bool use_tickets = false;
    double f_max = 1.0;                     // share of the pairs that touch the most frequent word
    DevBuf<double> d_loss;
    // multi-GPU: central-word ranges per rank, steps (local windows of the pair order), exchange buffers
    cymf_comm *comm = nullptr;
    std::vector<int64_t> central_bounds;
    int32_t steps_per_epoch = 1;
    std::vector<int64_t> step_off;
    DevBuf<float> d_snapH, d_snapA, d_delta, d_scale;
    DevBuf<float2> d_snapB;
    bool have_data = false, have_params = false;
};

extern "C" int cymf_glove_create(cymf_glove **out, int32_t V, int32_t Vc, int32_t K, double learning_rate, double x_max,
                                 double alpha, int dtype, int mode, int device) {
    if (!out) return fail(CYMF_ERR_INVALID, "cymf_glove_create: out is NULL");
    *out = nullptr;
    if (V <= 0 || Vc <= 0 || K <= 0) return fail(CYMF_ERR_INVALID, "cymf_glove_create: V, Vc, K must be positive");
    if ((dtype != CYMF_F32 && dtype != CYMF_F64) || (mode != CYMF_MODE_EXACT && mode != CYMF_MODE_THROUGHPUT))
        return fail(CYMF_ERR_INVALID, "cymf_glove_create: bad dtype/mode");
    CYMF_TRY(use_device(device));
    cymf_glove *h = new cymf_glove();
    h->V = V; h->Vc = Vc; h->K = K; h->lr = learning_rate; h->x_max = x_max; h->alpha = alpha;
    h->dtype = dtype; h->mode = mode; h->device = device;
    if (mode == CYMF_MODE_THROUGHPUT) {   // rows are updated from all XCDs inside one kernel: see bpr.hip
        for (DevBuf<float> *b : {&h->f32.W, &h->f32.H, &h->f32.bW, &h->f32.bH, &h->f32.aW, &h->f32.aH, &h->f32.abW, &h->f32.abH}) b->fine = 2;
        for (DevBuf<double> *b : {&h->f64.W, &h->f64.H, &h->f64.bW, &h->f64.bH, &h->f64.aW, &h->f64.aH, &h->f64.abW, &h->f64.abH}) b->fine = 2;
        h->bW2.fine = h->bH2.fine = 2;
        h->step_path = dtype == CYMF_F32 && K <= 128 && V < (1 << 30) && !(getenv("CYMF_GLOVE_NO_STEP") && getenv("CYMF_GLOVE_NO_STEP")[0] == '1');
    }
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(CYMF_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    int rc = h->d_loss.alloc(1);
    if (rc) { (void)hipStreamDestroy(h->stream); delete h; return rc; }
    *out = h;
    return 0;
}

extern "C" int cymf_glove_set_steps_per_epoch(cymf_glove *h, int32_t steps) {
    if (!h || steps < 1) return fail(CYMF_ERR_INVALID, "cymf_glove_set_steps_per_epoch: bad arguments");
    if (h->have_data) return fail(CYMF_ERR_INVALID, "cymf_glove_set_steps_per_epoch must precede cymf_glove_set_data");
    h->steps_per_epoch = steps;
    return 0;
}

extern "C" int cymf_glove_attach_comm(cymf_glove *h, cymf_comm *c, const int64_t *central_bounds) {
    if (!h || !c || !central_bounds) return fail(CYMF_ERR_INVALID, "cymf_glove_attach_comm: bad arguments");
    if (h->have_data || h->have_params) return fail(CYMF_ERR_INVALID, "cymf_glove_attach_comm must precede cymf_glove_set_data and cymf_glove_upload");
    const int world = comm_world(c);
    if (central_bounds[0] != 0 || central_bounds[world] != h->V) return fail(CYMF_ERR_INVALID, "cymf_glove_attach_comm: bounds must run from 0 to V");
    for (int r = 0; r < world; ++r)
        if (central_bounds[r] > central_bounds[r + 1]) return fail(CYMF_ERR_INVALID, "cymf_glove_attach_comm: bounds not monotone");
    h->comm = c;
    h->central_bounds.assign(central_bounds, central_bounds + world + 1);
    return 0;
}

extern "C" int cymf_glove_set_data(cymf_glove *h, const int32_t *central, const int32_t *context, const double *counts,
                                   int64_t N) {
    if (!h || N < 0 || (N > 0 && (!central || !context || !counts))) return fail(CYMF_ERR_INVALID, "cymf_glove_set_data: bad arguments");
    CYMF_TRY(use_device(h->device));
    for (int64_t s = 0; s < N; ++s) {
        // the context bias is sized like the central table (glove.pyx:94), so context < V as well
        if (central[s] < 0 || central[s] >= h->V || context[s] < 0 || context[s] >= h->Vc || context[s] >= h->V)
            return fail(CYMF_ERR_INVALID, "cymf_glove_set_data: pair %lld out of range", (long long)s);
        if (!(counts[s] > 0)) return fail(CYMF_ERR_INVALID, "cymf_glove_set_data: count[%lld] must be > 0", (long long)s);
    }
    h->N = N;
    {
        std::vector<int64_t> nc((size_t)h->V, 0), nx((size_t)h->Vc, 0);
        int64_t mx = 0;
        for (int64_t s = 0; s < N; ++s) { mx = std::max(mx, ++nc[central[s]]); mx = std::max(mx, ++nx[context[s]]); }
        h->f_max = N > 0 ? (double)mx / (double)N : 1.0;
    }
    std::vector<int32_t> c(central, central + N), x(context, context + N);
    std::vector<double> cnt(counts, counts + N);
    h->level_off.clear();
    h->use_tickets = false;
    if (h->mode == CYMF_MODE_EXACT && N > 0) {   // the order is fixed for all epochs: schedule once
        const char *lv = getenv("CYMF_GLOVE_EXACT_LEVELS");
        if (!(lv && lv[0] == '1') && h->K <= 256) {   // dataflow: the pairs stay in their given order, the accesses of every row are numbered
            std::vector<uint32_t> kc, kx;
            number_turns(N, [&](int64_t s) { return central[s]; }, [&](int64_t s) { return context[s]; }, h->V, h->Vc, kc, kx);
            CYMF_TRY(h->ticket.ka.upload(kc.data(), kc.size(), h->stream));
            CYMF_TRY(h->ticket.kb.upload(kx.data(), kx.size(), h->stream));
            h->use_tickets = true;
        } else {
            std::vector<int64_t> order;
            level_schedule(N, central, context, h->V, h->Vc, order, h->level_off);
            for (int64_t p = 0; p < N; ++p) { c[p] = central[order[p]]; x[p] = context[order[p]]; cnt[p] = counts[order[p]]; }
        }
    }
    const int32_t S = h->step_path ? std::max(1, h->steps_per_epoch) : 1;
    h->step_off.assign((size_t)S + 1, 0);
    h->step_off[S] = N;
    if (h->comm) {
        if (!h->step_path) return fail(CYMF_ERR_UNSUPPORTED, "cymf_glove: a communicator needs the float32 throughput step path (K <= 128)");
        const int64_t lo = h->central_bounds[comm_rank(h->comm)], hi = h->central_bounds[comm_rank(h->comm) + 1];
        for (int64_t s = 0; s < N; ++s)
            if (central[s] < lo || central[s] >= hi)
                return fail(CYMF_ERR_INVALID, "cymf_glove_set_data: pair %lld has central word %d outside this rank's range [%lld, %lld)",
                            (long long)s, central[s], (long long)lo, (long long)hi);
    }
    if (h->step_path && N > 0) {
        // (step, central)-bucketed order (stable): step = window of the given order; hot context words: a word that
        // is the context of at least HOT pairs is updated with atomic deltas by the step kernel; the number of
        // wavefronts keeps the expected number of waves inside a COLD row's read-modify-write window at <= 2
        auto step_of = [&](int64_t l) -> int32_t { return (int32_t)(((__int128)l * S) / N); };
        std::fill(h->step_off.begin(), h->step_off.end(), 0);
        std::vector<int64_t> nx((size_t)h->Vc, 0);
        for (int64_t s = 0; s < N; ++s) nx[context[s]]++;
        for (int64_t l = 0; l < N; ++l) h->step_off[(size_t)step_of(l) + 1]++;
        h->step_off[0] = 0;
        for (int32_t q = 0; q < S; ++q) h->step_off[q + 1] += h->step_off[q];
        const int64_t HOT = 4096;
        std::vector<int64_t> nc((size_t)h->V + 1);
        for (int32_t q = 0; q < S; ++q) {
            const int64_t b = h->step_off[q], e = h->step_off[q + 1];   // the window is [b, e) of the given order too
            std::fill(nc.begin(), nc.end(), 0);
            for (int64_t l = b; l < e; ++l) nc[(size_t)central[l] + 1]++;
            for (int32_t v = 0; v < h->V; ++v) nc[v + 1] += nc[v];
            for (int64_t l = b; l < e; ++l) {
                const int64_t p = b + nc[central[l]]++;
                c[p] = central[l];
                x[p] = context[l] | (nx[context[l]] >= HOT ? (1 << 30) : 0);
                cnt[p] = counts[l];
            }
        }
        const double f_cold = (double)std::min<int64_t>(HOT, N) / (double)N;
        h->step_waves = std::max<int64_t>(1, std::min<int64_t>((int64_t)(2.0 / (f_cold * 8)), 256 * 8));
    }
    if (h->comm) {   // sequentialisation factors per step and context word (see build_step_counts in bpr.hip)
        const int world = comm_world(h->comm);
        std::vector<float> cntx((size_t)S * h->V, 0.0f);
        for (int32_t q = 0; q < S; ++q)
            for (int64_t l = h->step_off[q]; l < h->step_off[q + 1]; ++l) cntx[(size_t)q * h->V + context[l]] += 1.0f;
        DevBuf<float> dtmp;
        CYMF_TRY(dtmp.upload(cntx.data(), cntx.size(), h->stream));
        CYMF_TRY(comm_allreduce_sum_f32(h->comm, dtmp.p, (int64_t)cntx.size(), h->stream));
        CYMF_HIP(hipMemcpyAsync(cntx.data(), dtmp.p, cntx.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        CYMF_HIP(hipStreamSynchronize(h->stream));
        // per-touch contraction assumed for a context row under AdaGrad (no weight decay): lr/16.  Measured with eight
        // ranks on one GPU (V = 20000, 3 M pairs, K = 64, lr 0.05, 6 epochs, single rank 0.230): plain sums (rho -> 0) spike to
        // 14.6 in the second epoch; rho = 0.01 is calm but slow (0.256 at 32 steps per epoch, 0.283 at 2); 0.003 reaches 0.2335
        // at 32 steps per epoch, 0.259 at 8 -- many small steps matter more here than for BPR (few epochs, large early steps).
        double rho = h->lr / 16.0;
        if (const char *er = getenv("CYMF_GLOVE_DELTA_RHO")) rho = atof(er);   // experiments
        const double base = 1.0 - std::min(0.5, rho);
        for (float &v : cntx) {
            const double a = std::pow(base, (double)v / world);
            v = (float)(a < 1.0 - 1e-12 ? (1.0 - std::pow(a, world)) / (world * (1.0 - a)) : 1.0);
        }
        CYMF_TRY(h->d_scale.upload(cntx.data(), cntx.size(), h->stream));
    }
    CYMF_TRY(h->d_central.upload(c.data(), c.size(), h->stream));
    CYMF_TRY(h->d_context.upload(x.data(), x.size(), h->stream));
    if (h->dtype == CYMF_F32) CYMF_TRY(upload_f64(h->f32.counts, cnt.data(), cnt.size(), h->stream));
    else CYMF_TRY(upload_f64(h->f64.counts, cnt.data(), cnt.size(), h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    h->have_data = true;
    return 0;
}

template <typename T>
static int glove_upload(cymf_glove *h, GloveStore<T> &st, const double *W, const double *b, const double *Wc, const double *bc) {
    const size_t nW = (size_t)h->V * h->K, nH = (size_t)h->Vc * h->K;
    CYMF_TRY(upload_f64(st.W, W, nW, h->stream));
    CYMF_TRY(upload_f64(st.bW, b, (size_t)h->V, h->stream));
    CYMF_TRY(upload_f64(st.H, Wc, nH, h->stream));
    CYMF_TRY(upload_f64(st.bH, bc, (size_t)h->V, h->stream));
    CYMF_TRY(fill_dev<T>(st.aW, nW, (T)1, h->stream));
    CYMF_TRY(fill_dev<T>(st.aH, nH, (T)1, h->stream));
    CYMF_TRY(fill_dev<T>(st.abW, (size_t)h->V, (T)1, h->stream));
    CYMF_TRY(fill_dev<T>(st.abH, (size_t)h->V, (T)1, h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cymf_glove_upload(cymf_glove *h, const double *W, const double *bias, const double *Wc, const double *bias_c) {
    if (!h || !W || !bias || !Wc || !bias_c) return fail(CYMF_ERR_INVALID, "cymf_glove_upload: bad arguments");
    CYMF_TRY(use_device(h->device));
    if (h->dtype == CYMF_F32) CYMF_TRY(glove_upload(h, h->f32, W, bias, Wc, bias_c));
    else CYMF_TRY(glove_upload(h, h->f64, W, bias, Wc, bias_c));
    if (h->step_path) {
        CYMF_TRY(h->bW2.alloc((size_t)h->V));
        CYMF_TRY(h->bH2.alloc((size_t)h->V));
        hipLaunchKernelGGL(pack2_kernel, dim3(ew_blocks(h->V)), dim3(256), 0, h->stream, h->f32.bW.p, h->f32.abW.p, h->bW2.p, (int64_t)h->V);
        hipLaunchKernelGGL(pack2_kernel, dim3(ew_blocks(h->V)), dim3(256), 0, h->stream, h->f32.bH.p, h->f32.abH.p, h->bH2.p, (int64_t)h->V);
        CYMF_HIP(hipGetLastError());
        CYMF_HIP(hipStreamSynchronize(h->stream));
    }
    if (h->comm) {   // the state every rank starts from is the synchronised one
        const size_t VK = (size_t)h->V * h->K;
        if ((size_t)h->Vc != (size_t)h->V) return fail(CYMF_ERR_UNSUPPORTED, "cymf_glove: multi-GPU needs a square co-occurrence matrix (V == Vc)");
        CYMF_TRY(h->d_snapH.alloc(VK)); CYMF_TRY(h->d_snapA.alloc(VK)); CYMF_TRY(h->d_snapB.alloc((size_t)h->V));
        CYMF_TRY(h->d_delta.alloc(2 * VK + 2 * (size_t)h->V));
        CYMF_HIP(hipMemcpyAsync(h->d_snapH.p, h->f32.H.p, VK * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        CYMF_HIP(hipMemcpyAsync(h->d_snapA.p, h->f32.aH.p, VK * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        CYMF_HIP(hipMemcpyAsync(h->d_snapB.p, h->bH2.p, (size_t)h->V * sizeof(float2), hipMemcpyDeviceToDevice, h->stream));
        CYMF_HIP(hipStreamSynchronize(h->stream));
    }
    h->have_params = true;
    return 0;
}

extern "C" int cymf_glove_download(cymf_glove *h, double *W, double *bias, double *Wc, double *bias_c) {
    if (!h || !W || !bias || !Wc || !bias_c || !h->have_params) return fail(CYMF_ERR_INVALID, "cymf_glove_download: bad arguments / no params");
    CYMF_TRY(use_device(h->device));
    const size_t nW = (size_t)h->V * h->K, nH = (size_t)h->Vc * h->K;
    if (h->comm) {   // every rank ends with all central rows and biases (its own range was trained here)
        CYMF_TRY(comm_allgatherv(h->comm, h->f32.W.p, h->central_bounds.data(), (int64_t)h->K * (int64_t)sizeof(float), h->stream));
        CYMF_TRY(comm_allgatherv(h->comm, h->bW2.p, h->central_bounds.data(), (int64_t)sizeof(float2), h->stream));
    }
    if (h->step_path) {
        hipLaunchKernelGGL(unpack2_kernel, dim3(ew_blocks(h->V)), dim3(256), 0, h->stream, h->bW2.p, h->f32.bW.p, h->f32.abW.p, (int64_t)h->V);
        hipLaunchKernelGGL(unpack2_kernel, dim3(ew_blocks(h->V)), dim3(256), 0, h->stream, h->bH2.p, h->f32.bH.p, h->f32.abH.p, (int64_t)h->V);
        CYMF_HIP(hipGetLastError());
    }
    if (h->dtype == CYMF_F32) {
        CYMF_TRY(download_f64(h->f32.W, W, nW, h->stream)); CYMF_TRY(download_f64(h->f32.bW, bias, (size_t)h->V, h->stream));
        CYMF_TRY(download_f64(h->f32.H, Wc, nH, h->stream)); CYMF_TRY(download_f64(h->f32.bH, bias_c, (size_t)h->V, h->stream));
    } else {
        CYMF_TRY(download_f64(h->f64.W, W, nW, h->stream)); CYMF_TRY(download_f64(h->f64.bW, bias, (size_t)h->V, h->stream));
        CYMF_TRY(download_f64(h->f64.H, Wc, nH, h->stream)); CYMF_TRY(download_f64(h->f64.bH, bias_c, (size_t)h->V, h->stream));
    }
    return 0;
}

template <typename T>
static int glove_epoch(cymf_glove *h, GloveStore<T> &st, double *loss_out) {
    GloveDev<T> d;
    d.W = st.W.p; d.H = st.H.p; d.bW = st.bW.p; d.bH = st.bH.p;
    d.aW = st.aW.p; d.aH = st.aH.p; d.abW = st.abW.p; d.abH = st.abH.p;
    d.K = h->K; d.lr = (T)h->lr; d.x_max = (T)h->x_max; d.alpha = (T)h->alpha;
    CYMF_TRY(h->d_loss.zero(h->stream));
    if (h->N > 0) {
        if (h->mode == CYMF_MODE_THROUGHPUT && h->step_path) {
            if constexpr (sizeof(T) == 4) {
                GloveStepDev r;
                r.W = reinterpret_cast<float *>(st.W.p); r.H = reinterpret_cast<float *>(st.H.p);
                r.aW = reinterpret_cast<float *>(st.aW.p); r.aH = reinterpret_cast<float *>(st.aH.p);
                r.bW2 = h->bW2.p; r.bH2 = h->bH2.p;
                r.K = h->K; r.lr = (float)h->lr; r.x_max = (float)h->x_max; r.alpha = (float)h->alpha;
                const int32_t S = (int32_t)h->step_off.size() - 1;
                for (int32_t q = 0; q < S; ++q) {
                    const int64_t b = h->step_off[q], n = h->step_off[q + 1] - b;
                    if (n > 0) {
                        const int64_t chunks = (n + 63) / 64;
                        int64_t waves = std::max<int64_t>(1, std::min(chunks, h->step_waves));
                        const int64_t cpw = (chunks + waves - 1) / waves;
                        waves = (chunks + cpw - 1) / cpw;
                        launch_glove_step(h->K, r, h->d_central.p + b, h->d_context.p + b, reinterpret_cast<const float *>(st.counts.p) + b, n,
                                          cpw, h->d_loss.p, (int)((waves + 3) / 4), h->stream);
                        CYMF_HIP(hipGetLastError());
                    }
                    if (h->comm) {   // every rank takes part in every step's exchange, with or without pairs of its own
                        const int64_t VK = (int64_t)h->V * h->K, V = h->V;
                        hipLaunchKernelGGL(glove_delta_kernel, dim3(ew_blocks(VK)), dim3(256), 0, h->stream, r.H, r.aH, r.bH2, h->d_snapH.p,
                                           h->d_snapA.p, h->d_snapB.p, h->d_delta.p, VK, V);
                        CYMF_HIP(hipGetLastError());
                        CYMF_TRY(comm_allreduce_sum_f32(h->comm, h->d_delta.p, 2 * VK + 2 * V, h->stream));
                        hipLaunchKernelGGL(glove_apply_kernel, dim3(ew_blocks(VK)), dim3(256), 0, h->stream, r.H, r.aH, r.bH2, h->d_snapH.p,
                                           h->d_snapA.p, h->d_snapB.p, h->d_delta.p, h->d_scale.p + (size_t)q * h->V, h->K, VK, V);
                        CYMF_HIP(hipGetLastError());
                    }
                }
            }
        } else if (h->mode == CYMF_MODE_THROUGHPUT) {
            launch_glove<T>(h->K, d, h->d_central.p, h->d_context.p, st.counts.p, h->N, h->d_loss.p, hogwild_grid(h->N, h->f_max), h->stream);
        } else if (h->use_tickets) {
            CYMF_TRY(h->ticket.prepare(h->V, h->Vc, h->device, h->stream));
            launch_glove_ticket<T>(h->K, d, h->d_central.p, h->d_context.p, st.counts.p, h->N, h->ticket, h->V, h->d_loss.p, h->stream);
            CYMF_HIP(hipGetLastError());
            CYMF_TRY(h->ticket.check(h->stream, "glove exact mode"));
        } else {
            for (size_t lv = 1; lv + 1 < h->level_off.size(); ++lv) {
                const int64_t b = h->level_off[lv], n = h->level_off[lv + 1] - b;
                if (n > 0) launch_glove<T>(h->K, d, h->d_central.p + b, h->d_context.p + b, st.counts.p + b, n, h->d_loss.p, (int)((n + 3) / 4), h->stream);
            }
        }
        CYMF_HIP(hipGetLastError());
    }
    double loss = 0;
    CYMF_HIP(hipMemcpyAsync(&loss, h->d_loss.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    if (loss_out) *loss_out = loss;
    return 0;
}

extern "C" int cymf_glove_epochs(cymf_glove *h, int32_t n_epochs, double *loss_out) {
    if (!h || n_epochs < 0) return fail(CYMF_ERR_INVALID, "cymf_glove_epochs: bad arguments");
    if (!h->have_data || !h->have_params) return fail(CYMF_ERR_INVALID, "cymf_glove_epochs before set_data/upload");
    CYMF_TRY(use_device(h->device));
    for (int32_t e = 0; e < n_epochs; ++e) {
        double *lo = loss_out ? loss_out + e : nullptr;
        if (h->dtype == CYMF_F32) CYMF_TRY(glove_epoch<float>(h, h->f32, lo)); else CYMF_TRY(glove_epoch<double>(h, h->f64, lo));
    }
    return 0;
}

extern "C" int cymf_glove_destroy(cymf_glove *h) {
    if (!h) return 0;
    if (!cymf::runtime_alive(h->device)) return 0;   // process exit / runtime already torn down: leak quietly
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
    return 0;
}
