// bpr.hip -- BPR negative-sampling SGD on gfx950.
// Replaces BPR._fit_bpr (cymf/bpr.pyx:117-190): epoch loop :160-171, BprModel.forward/backward
// (cymf/model.pyx:47-87) and the optimizers (cymf/optimizer.pyx:40-160), fused per triplet.
//
// One wavefront owns one triplet (u, i, j): the K factors of a row are spread over the 64 lanes
// (rows.h), the dot product is a DPP reduction, everything else is element-wise.
//
//   EXACT mode      : the reference's sequential order.  Two triplets commute iff their row sets
//                     {W[u], H[i], H[j]} are disjoint, so level(l) = 1 + max level of the last
//                     earlier triplet touching u, i or j gives conflict-free waves of work that are
//                     equivalent to the serial chain; one launch per level.
//   THROUGHPUT mode : HOGWILD (cymf/bpr.pyx:75).  Triplets are bucketed by positive item; a
//                     wavefront walks 64 consecutive slots of the item-sorted order with the
//                     positive item's row held in registers and flushed once per item run with a
//                     float atomic add of its delta; W[u] and H[j] rows are gathered ahead and
//                     written back in place.
#include "store.h"
#include "bpr_groups.h"

#include <algorithm>
#include <chrono>
#include <cmath>

namespace cymf {

// =====================================================================================
//                                        kernels
// =====================================================================================
namespace {

template <typename T>
struct BprDev {
    T *W, *H;          // (U,K), (I,K)
    T *W0, *W1;        // optimizer state for W (AdaGrad: acc; Adam: m, v)
    T *H0, *H1;        // optimizer state for H
    int K;
    T wd;
    OptParams<T> opt;
};

__device__ __forceinline__ float softplus_neg(float x) {   // -log(sigmoid(x)), overflow-free
    return fmaxf(-x, 0.0f) + log1pf(__expf(-fabsf(x)));
}
__device__ __forceinline__ double softplus_neg(double x) { return fmax(-x, 0.0) + log1p(exp(-fabs(x))); }
__device__ __forceinline__ float inv1pexp(float x) { return 1.0f / (1.0f + expf(x)); }
__device__ __forceinline__ double inv1pexp(double x) { return 1.0 / (1.0 + exp(x)); }

// forward (model.pyx:47-62) + backward (model.pyx:66-87) on rows already in registers.
// Gradients of every component use the pre-update values (model.pyx:81-87).
// HOG (lock-free float32 kernels): the per-triplet chain is what one wavefront runs back to back, so it is kept short --
// s = 1 / (1 + e^x) and the softplus through v_exp_f32 / v_log_f32 (relative error ~1e-6: far inside the lock-free mode's
// statistical bar; the exact mode keeps expf / log1pf), and the weight-decay term of the loss is accumulated per lane in
// `l2_acc` and reduced once per wavefront at the end instead of once per triplet (one wave-wide reduction less on the chain).
template <typename T, int R, bool PACKED, int OPT, bool HOG = false>
__device__ __forceinline__ T bpr_update_rows(const BprDev<T> &d, Row<T, R, PACKED> &w, Row<T, R, PACKED> &hi,
                                             Row<T, R, PACKED> &hj, Row<T, R, PACKED> *sw, Row<T, R, PACKED> *shi,
                                             Row<T, R, PACKED> *shj, T *l2_acc = nullptr) {
    T px = 0, pl = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        px += w.v[r] * (hi.v[r] - hj.v[r]);
        pl += w.v[r] * w.v[r] + hi.v[r] * hi.v[r] + hj.v[r] * hj.v[r];
    }
    const T x = wave_sum(px);
    T loss, s;
    if constexpr (HOG && sizeof(T) == 4) {
        *l2_acc += pl;
        // raw v_exp_f32 / v_log_f32 (base 2, no denormal rescue: e below 2^-126 may flush to 0, 1 + e lies in [1, 2])
        const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));          // e^{-|x|} in (0, 1]
        loss = fmaxf(-x, 0.0f) + 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + e);   // -log(sigmoid(x))
        const float r1 = __builtin_amdgcn_rcpf(1.0f + e);        // sigmoid(|x|)
        s = x >= 0.0f ? e * r1 : r1;                             // 1 / (1 + e^x)
    } else {
        const T l2 = wave_sum(pl);
        loss = softplus_neg(x) + d.wd * l2;
        s = inv1pexp(x);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const T wv = w.v[r], iv = hi.v[r], jv = hj.v[r];
        const T gw = -(s * (iv - jv) - d.wd * wv);
        const T gi = -(s * wv - d.wd * iv);
        const T gj = -(s * (-wv) - d.wd * jv);
        T dummy = 0;
        opt_update<T, OPT, HOG>(d.opt, w.v[r], OPT >= 1 ? sw[0].v[r] : dummy, OPT == 2 ? sw[1].v[r] : dummy, gw);
        opt_update<T, OPT, HOG>(d.opt, hi.v[r], OPT >= 1 ? shi[0].v[r] : dummy, OPT == 2 ? shi[1].v[r] : dummy, gi);
        opt_update<T, OPT, HOG>(d.opt, hj.v[r], OPT >= 1 ? shj[0].v[r] : dummy, OPT == 2 ? shj[1].v[r] : dummy, gj);
    }
    return loss;
}

// ---------------------------------------------------------------- EXACT: one level = independent triplets
template <typename T, int R, bool PACKED, int OPT>
__global__ __launch_bounds__(256) void bpr_level_kernel(BprDev<T> d, const int32_t *__restrict__ tu,
                                                       const int32_t *__restrict__ ti,
                                                       const int32_t *__restrict__ tj, int n,
                                                       double *__restrict__ loss_acc) {
    const int lane = lane_id();
    const int t = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (t >= n) return;
    const int K = d.K;
    const int64_t ou = (int64_t)tu[t] * K, oi = (int64_t)ti[t] * K, oj = (int64_t)tj[t] * K;
    constexpr int NS = opt_num_states(OPT);
    Row<T, R, PACKED> w, hi, hj, sw[NS ? NS : 1], shi[NS ? NS : 1], shj[NS ? NS : 1];
    w.load(d.W + ou, K, lane);
    hi.load(d.H + oi, K, lane);
    hj.load(d.H + oj, K, lane);
    if constexpr (NS >= 1) { sw[0].load(d.W0 + ou, K, lane); shi[0].load(d.H0 + oi, K, lane); shj[0].load(d.H0 + oj, K, lane); }
    if constexpr (NS >= 2) { sw[1].load(d.W1 + ou, K, lane); shi[1].load(d.H1 + oi, K, lane); shj[1].load(d.H1 + oj, K, lane); }
    const T loss = bpr_update_rows<T, R, PACKED, OPT>(d, w, hi, hj, sw, shi, shj);
    w.store(d.W + ou, K, lane);
    hi.store(d.H + oi, K, lane);
    hj.store(d.H + oj, K, lane);
    if constexpr (NS >= 1) { sw[0].store(d.W0 + ou, K, lane); shi[0].store(d.H0 + oi, K, lane); shj[0].store(d.H0 + oj, K, lane); }
    if constexpr (NS >= 2) { sw[1].store(d.W1 + ou, K, lane); shi[1].store(d.H1 + oi, K, lane); shj[1].store(d.H1 + oj, K, lane); }
    if (lane == 0) atomicAdd(loss_acc, (double)loss);
}

// ---------------------------------------------------------------- any K: rows streamed from memory in two passes
// The register layouts above hold a row in at most 4 values per lane (K <= 256).  The reference takes any
// num_components (cymf/bpr.pyx:50): wider rows run here, one wavefront per triplet, lanes striding over k --
// pass 1 forms x and the l2 term (cymf/model.pyx:52-59), pass 2 updates component by component from the
// pre-update values (cymf/model.pyx:78-87).  HOG = false: one conflict-free level of the exact order (`tj` never
// negative); HOG = true: a step of the lock-free mode over the item-sorted slots (`tj` = slot_neg: -1 skipped,
// bit 30 the hot flag), the item rows receive their deltas as float atomics so that no concurrent update is lost.
template <typename T, int OPT, bool HOG>
__global__ __launch_bounds__(256) void bpr_wide_kernel(BprDev<T> d, const int32_t *__restrict__ tu,
                                                      const int32_t *__restrict__ ti, const int32_t *__restrict__ tj,
                                                      int64_t n, double *__restrict__ loss_acc,
                                                      unsigned long long *__restrict__ performed_acc) {
    const int lane = lane_id();
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int K = d.K;
    double loss_sum = 0.0;
    unsigned long long n_done = 0;
    for (int64_t t = wave0; t < n; t += n_waves) {
        const int32_t jraw = tj[t];
        if (jraw < 0) continue;                                     // skipped draw (cymf/bpr.pyx:166-167)
        const int64_t ou = (int64_t)tu[t] * K, oi = (int64_t)ti[t] * K, oj = (int64_t)(jraw & 0x3fffffff) * K;
        T px = 0, pl = 0;
        for (int k = lane; k < K; k += 64) {
            const T wv = d.W[ou + k], iv = d.H[oi + k], jv = d.H[oj + k];
            px += wv * (iv - jv);
            pl += wv * wv + iv * iv + jv * jv;
        }
        const T x = wave_sum(px), l2 = wave_sum(pl);
        loss_sum += (double)(softplus_neg(x) + d.wd * l2);
        const T s = inv1pexp(x);
        for (int k = lane; k < K; k += 64) {
            T wv = d.W[ou + k], iv = d.H[oi + k], jv = d.H[oj + k];
            const T gw = -(s * (iv - jv) - d.wd * wv);
            const T gi = -(s * wv - d.wd * iv);
            const T gj = -(s * (-wv) - d.wd * jv);
            T w0 = 0, w1 = 0, i0 = 0, i1 = 0, j0 = 0, j1 = 0;
            if constexpr (OPT >= 1) { w0 = d.W0[ou + k]; i0 = d.H0[oi + k]; j0 = d.H0[oj + k]; }
            if constexpr (OPT == 2) { w1 = d.W1[ou + k]; i1 = d.H1[oi + k]; j1 = d.H1[oj + k]; }
            const T iv_old = iv, jv_old = jv;
            opt_update<T, OPT, HOG>(d.opt, wv, w0, w1, gw);
            opt_update<T, OPT, HOG>(d.opt, iv, i0, i1, gi);
            opt_update<T, OPT, HOG>(d.opt, jv, j0, j1, gj);
            d.W[ou + k] = wv;
            if constexpr (HOG && sizeof(T) == 4) {
                atomicAdd(reinterpret_cast<float *>(d.H) + oi + k, (float)(iv - iv_old));
                atomicAdd(reinterpret_cast<float *>(d.H) + oj + k, (float)(jv - jv_old));
            } else {
                d.H[oi + k] = iv;
                d.H[oj + k] = jv;
            }
            if constexpr (OPT >= 1) { d.W0[ou + k] = w0; d.H0[oi + k] = i0; d.H0[oj + k] = j0; }
            if constexpr (OPT == 2) { d.W1[ou + k] = w1; d.H1[oi + k] = i1; d.H1[oj + k] = j1; }
        }
        ++n_done;
    }
    if (lane == 0 && n_done) {
        atomicAdd(loss_acc, loss_sum);
        if (performed_acc) atomicAdd(performed_acc, n_done);
    }
}

// ---------------------------------------------------------------- EXACT: dataflow execution of the sequential order
// One launch per level costs a launch gap per level (~6 000 per epoch on ml-1m-shaped data).  The same
// dependence structure can be executed by ONE launch: triplet l may run as soon as the earlier triplets that
// touch its three rows are done.  The host numbers the accesses of every row in sequential order ("turns":
// triplet l is access number ku[l] of W[u], ki[l] of H[i], kj[l] of H[j]); a per-row counter on the device
// counts finished accesses; a wavefront takes the next triplet of the order from a dispenser, waits until the
// three counters show its turn numbers, updates the rows and bumps the counters.  Triplets are handed out in
// order to whichever wavefront asks next, so the smallest unfinished triplet is always held by a RUNNING
// wavefront and never waits on anything unfinished: the schedule cannot deadlock, whatever part of the grid is
// resident (no cooperative launch needed: that costs ~40 ms of queue draining per call here); a spin limit
// turns a broken schedule into an error instead of a hang.  Rows travel between wavefronts on different
// XCDs: the tables are uncached memory in this mode, the counters are bumped after an agent-scope release
// fence and the rows are loaded after an acquire fence (which also drops the CU's L1 lines).
constexpr unsigned int TICKET_SPIN_LIMIT = 1u << 20;   // polls of >= 0.2 us each; a healthy schedule waits < 10 ms for a turn

// all three counters are read in one round trip per poll (three sequential waits cost three round trips per triplet
// even when every turn has already come)
__device__ __forceinline__ bool ticket_wait3(const unsigned int *c0, unsigned int t0, const unsigned int *c1, unsigned int t1,
                                             const unsigned int *c2, unsigned int t2) {
    unsigned int spins = 0;
    while (true) {
        const unsigned int v0 = __hip_atomic_load(c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int v1 = __hip_atomic_load(c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int v2 = __hip_atomic_load(c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v0 == t0 && v1 == t1 && v2 == t2) return true;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > TICKET_SPIN_LIMIT) return false;
    }
}

template <typename T, int R, bool PACKED, int OPT>
__global__ __launch_bounds__(256) void bpr_ticket_kernel(BprDev<T> d, const int32_t *__restrict__ tu,
                                                        const int32_t *__restrict__ ti, const int32_t *__restrict__ tj,
                                                        const uint32_t *__restrict__ ku, const uint32_t *__restrict__ ki,
                                                        const uint32_t *__restrict__ kj, int64_t n,
                                                        unsigned int *doneW, unsigned int *doneH,
                                                        unsigned long long *next, double *__restrict__ loss_acc, int *err) {
    const int lane = lane_id();
    const int K = d.K;
    constexpr int NS = opt_num_states(OPT);
    double loss_sum = 0.0;
    auto grab = [&]() -> int64_t {
        unsigned long long v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(next, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
        return (int64_t)(((unsigned long long)hi << 32) | lo);
    };
    int64_t l = grab();
    while (l < n) {
        const int64_t l_next = grab();   // asked for early: its latency hides behind this triplet's waits and loads
        const int32_t u = tu[l], i = ti[l], j = tj[l];
        const bool ok = ticket_wait3(doneW + u, ku[l], doneH + i, ki[l], doneH + j, kj[l]);
        if (!ok) {   // cannot happen with a consistent schedule; never hang the device on a bad one
            if (lane == 0) atomicExch(err, 1);
            break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int64_t ou = (int64_t)u * K, oi = (int64_t)i * K, oj = (int64_t)j * K;
        Row<T, R, PACKED> w, hi, hj, sw[NS ? NS : 1], shi[NS ? NS : 1], shj[NS ? NS : 1];
        w.load(d.W + ou, K, lane);
        hi.load(d.H + oi, K, lane);
        hj.load(d.H + oj, K, lane);
        if constexpr (NS >= 1) { sw[0].load(d.W0 + ou, K, lane); shi[0].load(d.H0 + oi, K, lane); shj[0].load(d.H0 + oj, K, lane); }
        if constexpr (NS >= 2) { sw[1].load(d.W1 + ou, K, lane); shi[1].load(d.H1 + oi, K, lane); shj[1].load(d.H1 + oj, K, lane); }
        loss_sum += (double)bpr_update_rows<T, R, PACKED, OPT>(d, w, hi, hj, sw, shi, shj);
        w.store(d.W + ou, K, lane);
        hi.store(d.H + oi, K, lane);
        hj.store(d.H + oj, K, lane);
        if constexpr (NS >= 1) { sw[0].store(d.W0 + ou, K, lane); shi[0].store(d.H0 + oi, K, lane); shj[0].store(d.H0 + oj, K, lane); }
        if constexpr (NS >= 2) { sw[1].store(d.W1 + ou, K, lane); shi[1].store(d.H1 + oi, K, lane); shj[1].store(d.H1 + oj, K, lane); }
        // every lane's stores are complete (the fence waits for them wave-wide) before lane 0 publishes the turn
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) {
            __hip_atomic_fetch_add(doneW + u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(doneH + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(doneH + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        l = l_next;
    }
    if (lane == 0 && loss_sum != 0.0) atomicAdd(loss_acc, loss_sum);
}

// ---------------------------------------------------------------- THROUGHPUT: resolve negatives of a slot range
// The reference asks `negative in user_positives[user]` of a std::set per user (bpr.pyx:140,166).
// Device form: ONE open-addressing table over all (user, item) pairs of X (64-bit keys, load <= 1/2,
// linear probing): a membership test is a single 64-byte sector read almost always, where a binary
// search of the user's CSR row costs log2(n_u) dependent sector reads (measured: 93 GB of fetches
// per 100M-slot epoch, more than a third of what all step kernels of the epoch move).
__global__ __launch_bounds__(256) void pair_table_build_kernel(const int32_t *__restrict__ indptr,
                                                              const int32_t *__restrict__ indices, int32_t U,
                                                              unsigned long long *__restrict__ table,
                                                              unsigned long long mask) {
    // one wavefront per user row, lanes stride over the row's items
    const int lane = lane_id();
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave0; u < U; u += n_waves) {
        const int32_t p0 = indptr[u], p1 = indptr[u + 1];
        for (int32_t p = p0 + lane; p < p1; p += 64) {
            const unsigned long long key = ((unsigned long long)u << 32) | (unsigned int)indices[p];
            unsigned long long slot = pair_hash(key) & mask;
            while (true) {
                const unsigned long long prev = atomicCAS(table + slot, PAIR_EMPTY, key);
                if (prev == PAIR_EMPTY || prev == key) break;
                slot = (slot + 1) & mask;
            }
        }
    }
}

// slot t: triplet (slot_user[t], slot_item[t]) at global stream position slot_pos[t];
// negative = draws[pos]; skipped (-1) when it is one of the user's positives (bpr.pyx:165-167).
__global__ __launch_bounds__(256) void bpr_sample_kernel(const int32_t *__restrict__ slot_user,
                                                        const uint32_t *__restrict__ slot_pos,
                                                        const uint32_t *__restrict__ draws,
                                                        const unsigned long long *__restrict__ table,
                                                        unsigned long long mask,
                                                        const uint32_t *__restrict__ hot_bits,
                                                        int32_t *__restrict__ slot_neg, int64_t n,
                                                        unsigned long long *__restrict__ skipped) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned int my_skips = 0;
    for (; t < n; t += stride) {
        const int32_t u = slot_user[t];
        const int32_t j = (int32_t)draws[slot_pos[t]];
        const bool found = pair_table_has(table, mask, u, j);
        // bit 30 marks a negative that is a HOT item (many positive-side exchanges per step): the step
        // kernel adds its delta atomically instead of storing the row back
        const int32_t hot = (int32_t)((hot_bits[j >> 5] >> (j & 31)) & 1u) << 30;
        slot_neg[t] = found ? -1 : (j | hot);
        my_skips += found ? 1u : 0u;
    }
    // wave-level count, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) my_skips += __shfl_xor(my_skips, off, 64);
    if (lane_id() == 0 && my_skips) atomicAdd(skipped, (unsigned long long)my_skips);
}

// original-order view of the resolved negatives (test hook)
__global__ void bpr_unsort_neg_kernel(const int32_t *__restrict__ slot_neg, const uint32_t *__restrict__ slot_local,
                                      int32_t *__restrict__ out, int64_t n) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const int32_t j = slot_neg[t];
        out[slot_local[t]] = j < 0 ? j : (j & 0x3fffffff);
    }
}

// ---------------------------------------------------------------- THROUGHPUT: the step kernel
// HOGWILD with bounded staleness.  The step's slots are sorted by positive item; wavefront w walks
// its own CONTIGUOUS range of 64-slot chunks (blocked distribution), so the waves that are live at
// one moment work on items spread over the whole popularity range and only ~n_waves * f_i of them
// share a hot item i (f_i = its share of the triplets).
//   H[i] (positive item) : held in registers across the item run; every 64 slots the wave adds its
//        delta with a RETURNING float atomic and continues from (value found + delta): one coherent
//        exchange point per chunk, no lost update however many waves share the item.
//   H[j] (negative item) : gathered PF triplets ahead, its delta added with a float atomic: H is only
//        ever modified at the memory side, so a stale L2/L1 copy can delay an update but never undo one.
//   W[u] (user)          : gathered PF triplets ahead, written back in place (a user's triplets are
//        spread over the item-sorted order; two waves rarely hold the same user at once).
// Lane t of a chunk holds slot t's (user, item, negative); the walk itself is wave-uniform.
template <typename RowT, int R>
__device__ __forceinline__ void atomic_add_row(float *__restrict__ dst, const RowT &a, const RowT &b, int K, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = RowT::kof(lane, r);
        if (RowT::packed || k < K) atomicAdd(dst + k, a.v[r] - b.v[r]);      // no-return global_atomic_add_f32
    }
}

// dst += (cur - base); cur = base = value found at the memory side + own delta
template <typename RowT, int R>
__device__ __forceinline__ void exchange_row(float *__restrict__ dst, RowT &cur, RowT &base, int K, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = RowT::kof(lane, r);
        if (RowT::packed || k < K) {
            const float dlt = cur.v[r] - base.v[r];
            const float found = atomicAdd(dst + k, dlt);    // returning form
            cur.v[r] = found + dlt;
            base.v[r] = cur.v[r];
        }
    }
}

// Pins the s_waitcnt of loads/atomic returns into the (rare) branch that issued them.  Without
// it hipcc sinks the wait to the next use after the branch merge, i.e. a vmcnt(0) on the common
// path that drains the whole prefetch ring at every triplet.
template <typename RowT, int R>
__device__ __forceinline__ void settle(RowT &a) {
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(a.v[r]));
}

template <int R, bool PACKED, int OPT, int PF, int PFJ = PF>
__global__ __launch_bounds__(256) void bpr_step_kernel(BprDev<float> d, const int32_t *__restrict__ slot_user,
                                                      const int32_t *__restrict__ slot_item,
                                                      const int32_t *__restrict__ slot_neg, int64_t slot_begin,
                                                      int64_t slot_end, int64_t chunks_per_wave, int xcd_stride,
                                                      const int64_t *__restrict__ wave_ranges, int64_t n_waves,
                                                      double *__restrict__ loss_acc,
                                                      unsigned long long *__restrict__ performed_acc) {
    using RowT = Row<float, R, PACKED>;
    constexpr int NS = opt_num_states(OPT);
    constexpr int NSA = NS ? NS : 1;
    constexpr float SFILL = OPT == CYMF_OPT_ADAGRAD ? 1.0f : 0.0f;   // masked lanes of optimizer-state rows (rows.h: Row::load)
    static_assert(64 % PF == 0, "ring depth must divide the chunk");
    const int diag = xcd_stride >> 8;      // developer diagnostics (tools/sweep_step.py): bit0 no H[j] atomics, bit1 no W store, bit2 all H[j] atomic
    xcd_stride &= 255;
    if (blockIdx.x % xcd_stride) return;   // diagnostic: xcd_stride 8 keeps every active block on one XCD
    const int lane = lane_id();
    const int K = PACKED ? 64 * R : d.K;   // packed layouts: a compile-time row length (row offsets become shifts, no 64-bit scalar multiplies per row)
    const int64_t wave = ((int64_t)(blockIdx.x / xcd_stride) * blockDim.x + threadIdx.x) >> 6;
    // this wave's slots [slot_begin, slot_end): either an equal share of the step (SGD: an item run may be
    // split between waves, which then exchange at chunk boundaries) or an explicit ITEM-ALIGNED range
    // (AdaGrad/Adam: a run belongs to one wave, see the host side)
    const bool shared_runs = wave_ranges == nullptr;
    if (wave >= n_waves) return;
    const int64_t step_begin = slot_begin, step_end = slot_end;
    if (shared_runs) {
        const int64_t b0 = slot_begin + wave * chunks_per_wave * 64;
        const int64_t e0 = b0 + chunks_per_wave * 64;
        slot_begin = b0;
        slot_end = e0 < slot_end ? e0 : slot_end;
    } else {
        slot_begin = wave_ranges[wave];
        slot_end = wave_ranges[wave + 1];
    }
    const int64_t c_begin = 0;
    const int64_t c_end = (slot_end - slot_begin + 63) >> 6;
    if (c_end <= 0) return;
    // only the first and the last item run of the range can continue in another wave's range: runs
    // strictly inside are owned by this wave alone and need no exchange at chunk boundaries
    int shared_first = -1, shared_last = -1;
    if (shared_runs) {
        const int32_t fi = slot_item[slot_begin], li = slot_item[slot_end - 1];
        if (slot_begin > step_begin && slot_item[slot_begin - 1] == fi) shared_first = fi;
        if (slot_end < step_end && slot_item[slot_end] == li) shared_last = li;
    }
    float loss_sum = 0.0f, l2_acc = 0.0f;
    unsigned int n_done = 0;
    float *const Ws[2] = {d.W0, d.W1};
    float *const Hs[2] = {d.H0, d.H1};

    // slot metadata of the current and the next chunk, one slot per lane; j < 0 = nothing to do
    // (skipped draw, or past the end of the step); such slots gather row 0 and are predicated off.
    auto load_meta = [&](int64_t c, int32_t &u, int32_t &i, int32_t &j) {
        const int64_t my = slot_begin + (c << 6) + lane;
        const bool in = c < c_end && my < slot_end;
        u = in ? slot_user[my] : 0;
        i = in ? slot_item[my] : 0;
        j = in ? slot_neg[my] : -1;
        if (j < 0) u = 0;
    };
    int32_t u_c, i_c, j_c, u_n, i_n, j_n;
    load_meta(c_begin, u_c, i_c, j_c);
    load_meta(c_begin + 1, u_n, i_n, j_n);

    // Ring of RING = 2*PF entries of gathered (W[u], H[j]) row pairs (+ optimizer state rows): entry e
    // holds slot t with t % RING == e.  Slot t is updated IN PLACE in its entry and stored from there;
    // the entry refilled right after is (e + PF) % RING, whose stores were issued PF triplets ago.  So PF
    // triplets' loads are in flight (across chunk boundaries too) and PF triplets' stores are draining,
    // and no register that a pending store still reads is overwritten early -- hipcc guards such a
    // reuse with s_waitcnt vmcnt(N) (N = operations issued since), which at N ~ 0 drains the ring.
    // The two rings may have different depths: user rows come from HBM (W does not fit the Infinity
    // Cache) and want a long lead (PF), negative rows come from the cache-resident H and get a short one
    // (PFJ), which also shortens their read-modify-write window (fewer HOGWILD collisions per wave).
    constexpr int RING = 2 * PF;
    constexpr int RINGJ = 2 * PFJ;
    constexpr int XCHG = RING > 32 ? RING : 32;   // slots between exchanges of a shared item run
    static_assert(64 % RING == 0 && RING % RINGJ == 0, "rings must divide the chunk and each other");
    RowT wq[RING], jq[RINGJ], swq[RING][NSA], sjq[RINGJ][NSA];
    auto issue_w = [&](int p, int32_t u) {
        const int64_t ou = (int64_t)u * K;
        wq[p].load(d.W + ou, K, lane);
#pragma unroll
        for (int q = 0; q < NS; ++q) swq[p][q].load(Ws[q] + ou, K, lane, SFILL);
    };
    auto issue_j = [&](int p, int32_t j) {
        const int64_t oj = (int64_t)(j < 0 ? 0 : (j & 0x3fffffff)) * K;
        jq[p].load(d.H + oj, K, lane);
#pragma unroll
        for (int q = 0; q < NS; ++q) sjq[p][q].load(Hs[q] + oj, K, lane, SFILL);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) issue_w(p, bcast_lane(u_c, p));
#pragma unroll
    for (int p = 0; p < PFJ; ++p) issue_j(p, bcast_lane(j_c, p));

    int cur_item = -1;
    RowT hi, hi0, shi[NSA];
    hi.fill(0.0f); hi0.fill(0.0f);
#pragma unroll
    for (int q = 0; q < NSA; ++q) shi[q].fill(0.0f);

    for (int64_t c = c_begin; c < c_end; ++c) {
        n_done += (unsigned int)__popcll(__ballot(j_c >= 0));
#pragma unroll 1
        for (int t0 = 0; t0 < 64; t0 += RING) {
#pragma unroll
            for (int p = 0; p < RING; ++p) {
                const int t = t0 + p;
                const int u = bcast_lane(u_c, t), jraw = bcast_lane(j_c, t), item = bcast_lane(i_c, t);
                const int j = jraw & 0x3fffffff;
                const bool hot = (jraw >> 30) & 1;
                if (jraw >= 0) {                               // wave-uniform
                    if (item != cur_item) {                    // rare: next item run
                        if (cur_item >= 0) {                   // H[i] += (hi - hi0) of the finished run
                            atomic_add_row<RowT, R>(d.H + (int64_t)cur_item * K, hi, hi0, K, lane);
                            // the run's optimizer state is written back as the consistent set this wave holds.
                            // It is never delta-summed: a fast EMA (Adam's m decays ~completely within 64
                            // updates) summed over c waves has gain -(c-1) per exchange and diverges for c >= 3.
#pragma unroll
                            for (int q = 0; q < NS; ++q) shi[q].store(Hs[q] + (int64_t)cur_item * K, K, lane);
                        }
                        cur_item = item;
                        hi.load(d.H + (int64_t)item * K, K, lane);
#pragma unroll
                        for (int q = 0; q < NS; ++q) shi[q].load(Hs[q] + (int64_t)item * K, K, lane, SFILL);
                        settle<RowT, R>(hi);
                        hi0 = hi;
#pragma unroll
                        for (int q = 0; q < NS; ++q) settle<RowT, R>(shi[q]);
                    }
                    const int pj = p % RINGJ;
                    const RowT hj_old = jq[pj];
                    loss_sum += bpr_update_rows<float, R, PACKED, OPT, true>(d, wq[p], hi, jq[pj], swq[p], shi, sjq[pj], &l2_acc);
                    if (!(diag & 2)) wq[p].store(d.W + (int64_t)u * K, K, lane);
#pragma unroll
                    for (int q = 0; q < NS; ++q) swq[p][q].store(Ws[q] + (int64_t)u * K, K, lane);
                    // H[j]: a cold negative is written back in place (HOGWILD); a hot one -- an item whose
                    // positive-side deltas land every few microseconds -- gets its delta added atomically,
                    // so that this write cannot undo them
                    if (hot || (diag & 4)) {
                        if (!(diag & 1)) atomic_add_row<RowT, R>(d.H + (int64_t)j * K, jq[pj], hj_old, K, lane);
                    } else {
                        jq[pj].store(d.H + (int64_t)j * K, K, lane);
                    }
#pragma unroll
                    for (int q = 0; q < NS; ++q) sjq[pj][q].store(Hs[q] + (int64_t)j * K, K, lane);
                }
                // refill the entries PF / PFJ ahead with the rows of slots t + PF / t + PFJ (this chunk or the next)
                const int tw = t + PF, tj = t + PFJ;
                issue_w((p + PF) % RING, tw < 64 ? bcast_lane(u_c, tw & 63) : bcast_lane(u_n, tw & 63));
                issue_j((p + PFJ) % RINGJ, tj < 64 ? bcast_lane(j_c, tj & 63) : bcast_lane(j_n, tj & 63));
            }
            // every XCHG slots: exchange the open item's progress with the other waves that share its run.
            // The delta-sum of c concurrent waves is stable while c * (1 - (1 - lr*wd)^XCHG) < 1, so a shorter
            // interval buys a proportionally larger safe number of wavefronts.
            if ((t0 + RING) % XCHG == 0 && cur_item >= 0 && (cur_item == shared_first || cur_item == shared_last)) {
                exchange_row<RowT, R>(d.H + (int64_t)cur_item * K, hi, hi0, K, lane);
                settle<RowT, R>(hi);
                settle<RowT, R>(hi0);
            }
        }
        u_c = u_n; i_c = i_n; j_c = j_n;
        load_meta(c + 2, u_n, i_n, j_n);
    }
    if (cur_item >= 0) {
        atomic_add_row<RowT, R>(d.H + (int64_t)cur_item * K, hi, hi0, K, lane);
#pragma unroll
        for (int q = 0; q < NS; ++q) shi[q].store(Hs[q] + (int64_t)cur_item * K, K, lane);
    }
    const float l2_all = wave_sum(l2_acc);   // weight-decay term of the loss, all triplets of this wavefront
    if (lane == 0 && n_done) {
        atomicAdd(loss_acc, (double)(loss_sum + d.wd * l2_all));
        atomicAdd(performed_acc, (unsigned long long)n_done);
    }
}

// ---------------------------------------------------------------- small element-wise helpers
template <typename T>
__global__ void fill_kernel(T *p, T v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

template <typename T>
__global__ void cast_from_f64_kernel(const double *__restrict__ in, T *__restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (T)in[i];
}

template <typename T>
__global__ void cast_to_f64_kernel(const T *__restrict__ in, double *__restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (double)in[i];
}

// multi-GPU: delta = H - snap   /   H = snap = snap + summed delta
__global__ void delta_kernel(const float *__restrict__ H, const float *__restrict__ snap, float *__restrict__ delta, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) delta[i] = H[i] - snap[i];
}
constexpr int DELTA_TAIL = 16;   // floats behind the I x K item deltas of an exchange message (the first two: the measured data term)

// Multi-GPU: the data term of the per-touch contraction, MEASURED.  A touch of item row h by user row w moves it by
// lr s w with s = sigma(-x), x = w . (h - h_j): its Jacobian is -lr sigma'(x) w w^T, a contraction by lr sigma'(x) |w|^2 along w.
// One wavefront per sampled slot of the step (a stride through the item-sorted slots) forms sigma'(x) |w|^2 from the factors
// as they stand when the step starts; out[0] += that, out[1] += 1.  The two floats ride at the tail of the item-delta message,
// so every rank ends with the job-wide sums and computes the same factors (delta_scale_kernel) without a host round trip.
__global__ __launch_bounds__(256) void bpr_curvature_kernel(const float *__restrict__ W, const float *__restrict__ H, int K,
                                                           const int32_t *__restrict__ slot_user, const int32_t *__restrict__ slot_item,
                                                           const int32_t *__restrict__ slot_neg, int64_t b, int64_t e, int64_t n_samples,
                                                           float *__restrict__ out) {
    // a wavefront works CURV_PER_WAVE samples and adds once: thousands of atomics on ONE address serialise at ~12 ns each
    // (8 192 single-sample wavefronts: 210 us per step, a third of an eight-rank job's step kernel)
    constexpr int CURV_PER_WAVE = 8;
    const int lane = lane_id();
    const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (e <= b) return;
    float acc = 0.0f, cnt = 0.0f;
    for (int q = 0; q < CURV_PER_WAVE; ++q) {
        const int64_t smp = w0 * CURV_PER_WAVE + q;
        if (smp >= n_samples) break;
        const int64_t t = b + (smp * (e - b)) / n_samples;
        const int32_t jraw = slot_neg[t];
        if (jraw < 0) continue;
        const int64_t ou = (int64_t)slot_user[t] * K, oi = (int64_t)slot_item[t] * K, oj = (int64_t)(jraw & 0x3fffffff) * K;
        float px = 0.0f, pw = 0.0f;
        for (int k = lane; k < K; k += 64) {
            const float wv = W[ou + k];
            px += wv * (H[oi + k] - H[oj + k]);
            pw += wv * wv;
        }
        const float x = wave_sum(px), w2 = wave_sum(pw);
        const float sg = 1.0f / (1.0f + __expf(x));
        acc += sg * (1.0f - sg) * w2;
        cnt += 1.0f;
    }
    if (lane == 0 && cnt > 0.0f) {
        atomicAdd(out, acc);
        atomicAdd(out + 1, cnt);
    }
}

// scale[i] = (1 - a^N) / (N (1 - a)), a = (1 - rho)^(n_i / N): the sequentialisation factor of item i's summed deltas for this
// step (build_step_counts), rho = rho_fixed + lr_data * (curv[0] / curv[1]) -- the measured data term, job-wide (curv = the two
// floats at the tail of the reduced message; curv == nullptr or lr_data == 0: rho_fixed alone).
__global__ void delta_scale_kernel(const float *__restrict__ counts, int I, int world, float rho_fixed, float lr_data,
                                   const float *__restrict__ curv, float *__restrict__ scale) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= I) return;
    float rho = rho_fixed;
    if (curv && lr_data > 0.0f && curv[1] > 0.5f) rho += lr_data * curv[0] / curv[1];
    rho = fminf(rho, 0.5f);
    const float lb = log1pf(-rho);
    const float a = __expf(lb * counts[i] / (float)world);
    const float aN = __expf(lb * counts[i]);
    scale[i] = a < 1.0f - 1e-6f ? (1.0f - aN) / ((float)world * (1.0f - a)) : 1.0f;
}

// scale[row]: the sequentialisation factor of that item for this step (delta_scale_kernel)
// overlapped exchange: `base` is the state every rank agrees on bit for bit (all damped sums applied so far); the
// live table is base + this rank's delta of the step whose exchange is still in flight.  When the sum of the
// previous step arrives: base += s * sum, H = base + (delta of the step just computed), reference point = H.
__global__ void correct_delta_kernel(float *__restrict__ H, float *__restrict__ snap, float *__restrict__ base,
                                     const float *__restrict__ glob, const float *__restrict__ local_cur,
                                     const float *__restrict__ scale, int K, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float bnew = base[i] + scale[i / K] * glob[i];
        const float v = local_cur ? bnew + local_cur[i] : bnew;
        base[i] = bnew;
        H[i] = v;
        snap[i] = v;
    }
}
__global__ void snapshot_kernel(const float *__restrict__ H, float *__restrict__ snap, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) snap[i] = H[i];
}
__global__ void apply_delta_kernel(float *__restrict__ H, float *__restrict__ snap, const float *__restrict__ delta,
                                   const float *__restrict__ scale, int K, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = snap[i] + scale[i / K] * delta[i];
        H[i] = v;
        snap[i] = v;
    }
}

inline int ew_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

// =====================================================================================
//                                  typed device storage
// =====================================================================================
template <typename T>
int upload_f64(DevBuf<T> &dst, const double *src, size_t n, hipStream_t s) {
    CYMF_TRY(dst.alloc(n));
    if constexpr (sizeof(T) == sizeof(double)) {
        CYMF_HIP(hipMemcpyAsync(dst.p, src, n * sizeof(double), hipMemcpyHostToDevice, s));
        CYMF_HIP(hipStreamSynchronize(s));   // the caller's array (possibly a temporary of the wrapper) is consumed on return
    } else {
        DevBuf<double> tmp;
        tmp.fine = staging_memtype();
        CYMF_TRY(tmp.alloc(n));
        CYMF_HIP(hipMemcpyAsync(tmp.p, src, n * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(cast_from_f64_kernel<T>, dim3(ew_blocks((int64_t)n)), dim3(256), 0, s, tmp.p, dst.p, (int64_t)n);
        CYMF_HIP(hipGetLastError());
        CYMF_HIP(hipStreamSynchronize(s));
    }
    return 0;
}

template <typename T>
int download_f64(const DevBuf<T> &src, double *dst, size_t n, hipStream_t s) {
    if constexpr (sizeof(T) == sizeof(double)) {
        CYMF_HIP(hipMemcpyAsync(dst, src.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
        CYMF_HIP(hipStreamSynchronize(s));
    } else {
        DevBuf<double> tmp;
        tmp.fine = staging_memtype();
        CYMF_TRY(tmp.alloc(n));
        hipLaunchKernelGGL(cast_to_f64_kernel<T>, dim3(ew_blocks((int64_t)n)), dim3(256), 0, s, src.p, tmp.p, (int64_t)n);
        CYMF_HIP(hipGetLastError());
        CYMF_HIP(hipMemcpyAsync(dst, tmp.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
        CYMF_HIP(hipStreamSynchronize(s));
    }
    return 0;
}

template int upload_f64<float>(DevBuf<float> &, const double *, size_t, hipStream_t);
template int upload_f64<double>(DevBuf<double> &, const double *, size_t, hipStream_t);
template int download_f64<float>(const DevBuf<float> &, double *, size_t, hipStream_t);
template int download_f64<double>(const DevBuf<double> &, double *, size_t, hipStream_t);

template <typename T>
int fill_dev(DevBuf<T> &b, size_t n, T v, hipStream_t s) {
    CYMF_TRY(b.alloc(n));
    hipLaunchKernelGGL(fill_kernel<T>, dim3(ew_blocks((int64_t)n)), dim3(256), 0, s, b.p, v, (int64_t)n);
    CYMF_HIP(hipGetLastError());
    return 0;
}
template int fill_dev<float>(DevBuf<float> &, size_t, float, hipStream_t);
template int fill_dev<double>(DevBuf<double> &, size_t, double, hipStream_t);

// =====================================================================================
//                                       the trainer
// =====================================================================================
template <typename T>
struct BprStore {
    DevBuf<T> W, H, W0, W1, H0, H1;
    BprDev<T> view(int K, double wd, double lr) {
        BprDev<T> d;
        d.W = W.p; d.H = H.p; d.W0 = W0.p; d.W1 = W1.p; d.H0 = H0.p; d.H1 = H1.p;
        d.K = K; d.wd = (T)wd; d.opt = make_opt_params<T>(lr);
        return d;
    }
};

}  // namespace cymf

using namespace cymf;

struct cymf_bpr {
    int32_t U = 0, I = 0, K = 0;
    int opt = 0, dtype = 0, mode = 0, device = 0;
    double lr = 0, wd = 0;
    uint32_t seed = 1234;
    hipStream_t stream = nullptr, rng_stream = nullptr;
    int prio_side = 0, prio_comm = 0;                          // stream priorities (see cymf_bpr_create)
    hipStream_t gen_stream = nullptr;                          // batched index-stream generation (draw_batch > 1), beside the sampling
    hipEvent_t ev_batch_sampled[2] = {nullptr, nullptr};       // the last epoch of a batch of draws has been sampled (buffer reusable)
    BprStore<float> f32;
    BprStore<double> f64;
    bool have_params = false, have_data = false;

    // data
    int64_t N = 0, N_global = 0;
    std::vector<int32_t> h_users, h_pos_items, h_indptr, h_indices;
    bool gpos_identity = true;            // h_gpos[l] == l (no global_pos given: one rank)
    std::vector<uint32_t> h_gpos;
    DevBuf<int32_t> d_indptr, d_indices;
    DevBuf<unsigned long long> d_pair_table;   // throughput: open-addressing set of the (user, item) pairs of X
    unsigned long long pair_mask = 0;

    // stream of negatives
    DeviceRng rng;
    bool rng_ready = false;
    DevBuf<uint32_t> d_draws[2];
    int64_t draw_batch = 1;         // epochs generated per call of the index-stream generator (see request_epoch_draws)
    hipEvent_t ev_gen[2] = {nullptr, nullptr}, ev_sampled[2] = {nullptr, nullptr};
    int64_t epochs_generated = 0;   // draws of epochs [0, epochs_generated) have been requested
    int64_t epoch_cursor = 0;       // next epoch to train
    int32_t step_cursor = 0;        // next step inside epoch_cursor (throughput)
    bool epoch_sampled = false;     // slot_neg holds epoch_cursor's negatives

    // exact mode scratch
    uint32_t *h_draws2[2] = {nullptr, nullptr};   // pinned (hipHostMalloc): the exact mode reads the epoch's draws on the host
    // ... and writes the epoch's schedule (performed triplets and their turns) for the device: two sets, by epoch parity -- the
    // schedule of epoch e + 1 is numbered and sent up while the dataflow kernel of epoch e runs (exact_prepare)
    PinnedBuf<int32_t> p_tu[2], p_ti[2], p_tj[2];
    PinnedBuf<uint32_t> p_ku[2], p_ki[2], p_kj[2];
    DevBuf<int32_t> d_xu[2], d_xi[2], d_xj[2];
    DevBuf<uint32_t> d_xku[2], d_xki[2], d_xkj[2];
    struct ExactPrep { int64_t epoch = -1, n_perf = 0, n_skipped = 0; double t_draws = 0, t_turns = 0; } prep[2];
    hipStream_t up_stream = nullptr;              // the schedule's uploads (they must not queue behind the running kernel)
    hipEvent_t ev_up[2] = {nullptr, nullptr};
    int64_t h_draws_cap[2] = {0, 0};
    int64_t exact_fetched = 0;
    hipEvent_t ev_draws_host[2] = {nullptr, nullptr};
    std::vector<int32_t> h_last_neg;
    DevBuf<int32_t> d_tu, d_ti, d_tj;
    // exact mode, dataflow execution (bpr_ticket_kernel): turn numbers per triplet, finished-access counters per row
    bool exact_tickets = true;   // CYMF_BPR_EXACT_LEVELS=1 selects one launch per level instead
    std::vector<uint64_t> h_pos_bits;   // exact mode: U x I membership bitmap when it fits 256 MB (else binary search of the CSR row)
    DevBuf<uint32_t> d_ku, d_ki, d_kj, d_done;
    DevBuf<unsigned long long> d_next;
    DevBuf<int> d_err;
    int n_cu = 0;

    // throughput mode
    int32_t steps_per_epoch = 1;
    bool steps_auto = false;              // steps_per_epoch = 0 was asked for: chosen from the data (choose_steps_per_epoch)
    int32_t max_waves = 256 * 12;         // hardware side: 12 wavefronts per CU (measured plateau on C3)
    double f_item_max = 0.0;              // share of the triplets that carry the most popular positive item
    int32_t rows_per_inflight = 8;        // staleness bound: table rows per row in flight
    bool item_aligned = false;            // experiment: AdaGrad/Adam item runs owned by one wave (CYMF_BPR_ITEM_ALIGNED=1)
    int32_t adaptive_rpi_factor = 2;      // AdaGrad/Adam: stricter rows-in-flight bound (state RMW is not atomic)
    int32_t step_pf = 84;                 // prefetch ring depth of the step kernel (8 or 16)
    int32_t group_mode = -1;              // small tables: the group kernel (bpr_groups.hip).  -1 = when the bound on the rows in flight,
                                          // not the chip, would size the step kernel's launch; 0 / 1 = CYMF_BPR_GROUPS
    int32_t group_waves = 0;              // its wavefronts (0 = default by optimizer; CYMF_BPR_GROUP_WAVES)
    int32_t xcd_stride = 1;               // diagnostic (CYMF_BPR_XCD_STRIDE=8: all active blocks on one XCD)
    std::vector<int64_t> step_off;           // slot offsets, steps_per_epoch+1
    DevBuf<int32_t> d_slot_user, d_slot_item, d_slot_neg[2];   // slot_neg double-buffered by epoch parity
    hipEvent_t ev_epoch_done[2] = {nullptr, nullptr};          // last step of an epoch finished reading slot_neg[b]
    int64_t epochs_sampled = 0;                                // throughput: epochs [0, epochs_sampled) have been sampled
    DevBuf<uint32_t> d_slot_pos, d_slot_local;
    DevBuf<unsigned long long> d_skipped, d_performed;
    DevBuf<int64_t> d_wave_ranges;       // AdaGrad/Adam: item-aligned slot ranges of the waves, all steps
    std::vector<int64_t> wave_range_off; // per step: offset into d_wave_ranges (n_ranges + 1 entries each)
    std::vector<int32_t> h_slot_item;    // host copy of slot_item (range construction)
    int64_t wave_ranges_for = -1;
    DevBuf<uint32_t> d_hot_bits;         // bit i set: item i is hot (see bpr_sample_kernel)
    int32_t hot_threshold = 256;         // positives per step from which an item counts as hot
    int64_t slots_done = 0;              // slots walked by the step kernels since create
    DevBuf<int32_t> d_unsorted_neg;

    DevBuf<double> d_loss;
    int64_t performed = 0, skipped = 0;

    // multi-GPU
    cymf_comm *comm = nullptr;
    DevBuf<float> d_snap, d_delta;
    // overlapped exchange (default): the all-reduce of step s runs on comm_stream under the step kernel of step s+1
    bool overlap_exchange = true;   // CYMF_BPR_SYNC_EXCHANGE=1: exchange and apply before the next step instead
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_delta_ready = nullptr, ev_reduced[2] = {nullptr, nullptr};
    DevBuf<float> d_local[2], d_glob[2], d_base;
    bool exch_pending = false;
    std::vector<int64_t> user_bounds;   // [world + 1] user ranges of the ranks: download() gathers the W rows (optional)
    int exch_parity = 0;
    int64_t exch_count = 0;
    int32_t exch_step = 0;              // step whose exchange is in flight (its touch counts size the factors)
    DevBuf<float> d_step_counts;        // [steps_per_epoch][I] job-wide touches of every item per step (build_step_counts)
    DevBuf<float> d_delta_scale;        // [I] sequentialisation factors of the summed deltas of the step being applied
    double rho_fixed = 0.0, rho_lr_data = 0.0;   // per-touch contraction: fixed part, and the learning rate of the measured data term

    // profiling of the dominant kernel
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms = 0;
    int64_t prof_launches = 0, prof_units = 0;
};

namespace cymf {
int comm_allreduce_sum_f32(cymf_comm *c, float *d_buf, int64_t n, hipStream_t s);   // comm.hip
int comm_allreduce_sum_f32_to(cymf_comm *c, const float *d_in, float *d_out, int64_t n, int64_t cap, hipStream_t s);
int64_t comm_padded_count(cymf_comm *c, int64_t n);
int comm_allgatherv(cymf_comm *c, void *d_buf, const int64_t *row_bounds, int64_t row_bytes, hipStream_t s);
int comm_rank(cymf_comm *c);
int comm_world(cymf_comm *c);
}


namespace {

int layout_R(int K) { return (K + 63) / 64; }
bool layout_packed(int K) { return K == 128 || K == 256; }

#define CYMF_DISPATCH_LAYOUT(K, CALL)                                      \
    do {                                                                   \
        const int R__ = layout_R(K);                                       \
        if (R__ == 1) { CALL(1, false); }                                  \
        else if (R__ == 2) { if (layout_packed(K)) { CALL(2, true); } else { CALL(2, false); } } \
        else if (R__ == 3) { CALL(3, false); }                             \
        else { if (layout_packed(K)) { CALL(4, true); } else { CALL(4, false); } } \
    } while (0)

template <typename T, int R, bool PACKED>
void launch_level_opt(int opt, const BprDev<T> &d, const int32_t *tu, const int32_t *ti, const int32_t *tj, int n,
                      double *loss, hipStream_t s) {
    dim3 grid((n + 3) / 4), block(256);
    switch (opt) {
    case CYMF_OPT_SGD: hipLaunchKernelGGL((bpr_level_kernel<T, R, PACKED, CYMF_OPT_SGD>), grid, block, 0, s, d, tu, ti, tj, n, loss); break;
    case CYMF_OPT_ADAGRAD: hipLaunchKernelGGL((bpr_level_kernel<T, R, PACKED, CYMF_OPT_ADAGRAD>), grid, block, 0, s, d, tu, ti, tj, n, loss); break;
    default: hipLaunchKernelGGL((bpr_level_kernel<T, R, PACKED, CYMF_OPT_ADAM>), grid, block, 0, s, d, tu, ti, tj, n, loss); break;
    }
}

template <typename T, bool HOG>
void launch_wide(int opt, const BprDev<T> &d, const int32_t *tu, const int32_t *ti, const int32_t *tj, int64_t n, double *loss,
                 unsigned long long *perf, int64_t waves, hipStream_t s) {
    dim3 grid((unsigned)std::max<int64_t>(1, (waves + 3) / 4)), block(256);
    switch (opt) {
    case CYMF_OPT_SGD: hipLaunchKernelGGL((bpr_wide_kernel<T, CYMF_OPT_SGD, HOG>), grid, block, 0, s, d, tu, ti, tj, n, loss, perf); break;
    case CYMF_OPT_ADAGRAD: hipLaunchKernelGGL((bpr_wide_kernel<T, CYMF_OPT_ADAGRAD, HOG>), grid, block, 0, s, d, tu, ti, tj, n, loss, perf); break;
    default: hipLaunchKernelGGL((bpr_wide_kernel<T, CYMF_OPT_ADAM, HOG>), grid, block, 0, s, d, tu, ti, tj, n, loss, perf); break;
    }
}

template <typename T>
void launch_level(int K, int opt, const BprDev<T> &d, const int32_t *tu, const int32_t *ti, const int32_t *tj, int n,
                  double *loss, hipStream_t s) {
    if (K > 256) {   // rows wider than the register layouts: one wavefront per triplet of the level, two passes over k
        launch_wide<T, false>(opt, d, tu, ti, tj, n, loss, nullptr, n, s);
        return;
    }
#define CALL_(R_, P_) launch_level_opt<T, R_, P_>(opt, d, tu, ti, tj, n, loss, s)
    CYMF_DISPATCH_LAYOUT(K, CALL_);
#undef CALL_
}



// one launch, about as many workgroups as fit the device at once (more would only queue up behind them)
template <typename T, int R, bool PACKED, int OPT>
int launch_ticket_inst(const BprDev<T> &d, const int32_t *tu, const int32_t *ti, const int32_t *tj, const uint32_t *ku,
                       const uint32_t *ki, const uint32_t *kj, int64_t n, unsigned int *doneW, unsigned int *doneH,
                       unsigned long long *next, double *loss, int *err, int n_cu, hipStream_t s) {
    int64_t blocks = (int64_t)2 * n_cu;            // 8 waves per CU are plenty: the schedule is a latency chain
    blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, (n + 3) / 4));
    hipLaunchKernelGGL((bpr_ticket_kernel<T, R, PACKED, OPT>), dim3((unsigned)blocks), dim3(256), 0, s, d, tu, ti, tj, ku, ki, kj, n,
                       doneW, doneH, next, loss, err);
    CYMF_HIP(hipGetLastError());
    return 0;
}

template <typename T, int R, bool PACKED>
int launch_ticket_opt(int opt, const BprDev<T> &d, const int32_t *tu, const int32_t *ti, const int32_t *tj, const uint32_t *ku,
                      const uint32_t *ki, const uint32_t *kj, int64_t n, unsigned int *doneW, unsigned int *doneH,
                      unsigned long long *next, double *loss, int *err, int n_cu, hipStream_t s) {
    switch (opt) {
    case CYMF_OPT_SGD: return launch_ticket_inst<T, R, PACKED, CYMF_OPT_SGD>(d, tu, ti, tj, ku, ki, kj, n, doneW, doneH, next, loss, err, n_cu, s);
    case CYMF_OPT_ADAGRAD: return launch_ticket_inst<T, R, PACKED, CYMF_OPT_ADAGRAD>(d, tu, ti, tj, ku, ki, kj, n, doneW, doneH, next, loss, err, n_cu, s);
    default: return launch_ticket_inst<T, R, PACKED, CYMF_OPT_ADAM>(d, tu, ti, tj, ku, ki, kj, n, doneW, doneH, next, loss, err, n_cu, s);
    }
}

template <typename T>
int launch_ticket(int K, int opt, const BprDev<T> &d, const int32_t *tu, const int32_t *ti, const int32_t *tj, const uint32_t *ku,
                  const uint32_t *ki, const uint32_t *kj, int64_t n, unsigned int *doneW, unsigned int *doneH,
                  unsigned long long *next, double *loss, int *err, int n_cu, hipStream_t s) {
    int rc = 0;
#define CALL_(R_, P_) rc = launch_ticket_opt<T, R_, P_>(opt, d, tu, ti, tj, ku, ki, kj, n, doneW, doneH, next, loss, err, n_cu, s)
    CYMF_DISPATCH_LAYOUT(K, CALL_);
#undef CALL_
    return rc;
}

template <int R, bool PACKED, int STEP_PF, int STEP_PFJ = STEP_PF>
void launch_step_opt(int opt, const BprDev<float> &d, const int32_t *su, const int32_t *si, const int32_t *sn,
                     int64_t b, int64_t e, int64_t cpw, int xs, const int64_t *wr, int64_t nw, double *loss, unsigned long long *perf, int grid_blocks, hipStream_t s) {
    dim3 grid(grid_blocks * (xs & 255)), block(256);
    switch (opt) {
    case CYMF_OPT_SGD: hipLaunchKernelGGL((bpr_step_kernel<R, PACKED, CYMF_OPT_SGD, STEP_PF, STEP_PFJ>), grid, block, 0, s, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf); break;
    case CYMF_OPT_ADAGRAD: hipLaunchKernelGGL((bpr_step_kernel<R, PACKED, CYMF_OPT_ADAGRAD, STEP_PF, STEP_PFJ>), grid, block, 0, s, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf); break;
    default: hipLaunchKernelGGL((bpr_step_kernel<R, PACKED, CYMF_OPT_ADAM, STEP_PF, STEP_PFJ>), grid, block, 0, s, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf); break;
    }
}

template <int R, bool PACKED>
void launch_step_pf(int opt, const BprDev<float> &d, const int32_t *su, const int32_t *si, const int32_t *sn, int64_t b,
                    int64_t e, int64_t cpw, int xs, const int64_t *wr, int64_t nw, double *loss, unsigned long long *perf,
                    int grid_blocks, int pf, hipStream_t s) {
    if constexpr (R >= 3) {   // K > 128: 4 rows ahead (ring of 8) is what fits the register file without spilling
        launch_step_opt<R, PACKED, 4>(opt, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf, grid_blocks, s);
    } else if (opt == CYMF_OPT_ADAM) {   // 6 rows per entry
        launch_step_opt<R, PACKED, 4>(opt, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf, grid_blocks, s);
    } else {
        if (opt == CYMF_OPT_SGD && pf == 164) launch_step_opt<R, PACKED, 16, 4>(opt, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf, grid_blocks, s);
        else if (opt == CYMF_OPT_SGD && pf == 168) launch_step_opt<R, PACKED, 16, 8>(opt, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf, grid_blocks, s);
        else if (opt == CYMF_OPT_SGD && pf == 84) launch_step_opt<R, PACKED, 8, 4>(opt, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf, grid_blocks, s);
        else launch_step_opt<R, PACKED, 8>(opt, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf, grid_blocks, s);
    }
}

void launch_step(int K, int opt, const BprDev<float> &d, const int32_t *su, const int32_t *si, const int32_t *sn,
                 int64_t b, int64_t e, int64_t cpw, int xs, const int64_t *wr, int64_t nw, double *loss, unsigned long long *perf,
                 int grid_blocks, int pf, hipStream_t s) {
#define CALL_(R_, P_) launch_step_pf<R_, P_>(opt, d, su, si, sn, b, e, cpw, xs, wr, nw, loss, perf, grid_blocks, pf, s)
    CYMF_DISPATCH_LAYOUT(K, CALL_);
#undef CALL_
}

bool csr_has(const std::vector<int32_t> &indptr, const std::vector<int32_t> &indices, int32_t u, int32_t item) {
    const int32_t *b = indices.data() + indptr[u], *e = indices.data() + indptr[u + 1];
    return std::binary_search(b, e, item);
}

// overlapped exchange: wait for the all-reduce in flight (if any) and apply it (correct_delta_kernel); local_cur is
// the delta of the step just computed (nullptr when flushing: then H ends equal to the common base on every rank).
// With nothing in flight the reference point is simply refreshed (snapshot_if_idle).
int finish_exchange(cymf_bpr *h, bool snapshot_if_idle, const float *local_cur = nullptr) {
    const int64_t n = (int64_t)h->I * h->K;
    if (h->exch_pending) {
        const int pb = h->exch_parity;
        CYMF_HIP(hipStreamWaitEvent(h->stream, h->ev_reduced[pb], 0));
        hipLaunchKernelGGL(delta_scale_kernel, dim3((h->I + 255) / 256), dim3(256), 0, h->stream, h->d_step_counts.p + (size_t)h->exch_step * h->I,
                           h->I, comm_world(h->comm), (float)h->rho_fixed, (float)h->rho_lr_data, h->d_glob[pb].p + n, h->d_delta_scale.p);
        hipLaunchKernelGGL(correct_delta_kernel, dim3(ew_blocks(n)), dim3(256), 0, h->stream, h->f32.H.p, h->d_snap.p, h->d_base.p,
                           h->d_glob[pb].p, local_cur, h->d_delta_scale.p, h->K, n);
        CYMF_HIP(hipGetLastError());
        h->exch_pending = false;
    } else if (snapshot_if_idle) {
        hipLaunchKernelGGL(snapshot_kernel, dim3(ew_blocks(n)), dim3(256), 0, h->stream, h->f32.H.p, h->d_snap.p, n);
        CYMF_HIP(hipGetLastError());
    }
    return 0;
}

// ---- the draws of epoch `e`, generated in stream order on rng_stream.  The stream is ONE generator that is never reseeded
// (cymf/bpr.pyx:141): epoch e owns draws [e N, (e + 1) N).  Large epochs are generated one at a time into d_draws[e & 1].  A small
// epoch (ml-1m-shaped: 466 k draws) would be walked by ONE workgroup in ~0.8 ms -- three times what the group kernel needs for
// the epoch's triplets -- so the lock-free mode generates `draw_batch` epochs per call with the chunked jump-ahead generator
// (>= 4 M draws: enough chunks to spread over the chip) into d_draws[(e / draw_batch) & 1]; an epoch is a slice of its batch.
const uint32_t *epoch_draws(const cymf_bpr *h, int64_t e) {
    return h->d_draws[(int)((e / h->draw_batch) & 1)].p + (size_t)(e % h->draw_batch) * (size_t)h->N_global;
}

int request_epoch_draws(cymf_bpr *h, int64_t e) {
    while (h->epochs_generated <= e) {
        const int64_t g = h->epochs_generated, E = h->draw_batch;
        const int64_t q = g / E;
        const int b = (int)(q & 1);
        CYMF_TRY(h->d_draws[b].alloc((size_t)h->N_global * (size_t)E));
        hipStream_t gs = E > 1 ? h->gen_stream : h->rng_stream;
        // buffer b was last read by the sampling of epoch g - 2 (E == 1) / of the last epoch of batch q - 2 (batches: their
        // generation runs on its own stream, a whole batch ahead of its first use, so that it never stands in front of a sampling
        // kernel the step stream is waiting for)
        if (E == 1 && g >= 2) CYMF_HIP(hipStreamWaitEvent(gs, h->ev_sampled[b], 0));
        if (E > 1 && q >= 2) CYMF_HIP(hipStreamWaitEvent(gs, h->ev_batch_sampled[b], 0));
        CYMF_TRY(h->rng.generate(0, h->N_global * E, h->d_draws[b].p, gs));
        CYMF_HIP(hipEventRecord(h->ev_gen[b], gs));
        h->epochs_generated += E;
    }
    return 0;
}

int fetch_loss(cymf_bpr *h, double *out) {
    CYMF_HIP(hipMemcpyAsync(out, h->d_loss.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

// =============================== EXACT epoch ===============================
// draws of epoch g: generated on rng_stream and copied to pinned host memory of parity g & 1, all asynchronously
int exact_fetch_draws(cymf_bpr *h, int64_t g) {
    while (h->exact_fetched <= g) {
        const int64_t q = h->exact_fetched;
        const int b = (int)(q & 1);
        CYMF_TRY(request_epoch_draws(h, q));
        if (h->h_draws_cap[b] < h->N_global) {
            if (h->h_draws2[b]) (void)hipHostFree(h->h_draws2[b]);
            h->h_draws2[b] = nullptr;
            CYMF_HIP(hipHostMalloc((void **)&h->h_draws2[b], (size_t)std::max<int64_t>(h->N_global, 1) * sizeof(uint32_t)));
            h->h_draws_cap[b] = h->N_global;
        }
        CYMF_HIP(hipMemcpyAsync(h->h_draws2[b], epoch_draws(h, q), (size_t)h->N_global * sizeof(uint32_t), hipMemcpyDeviceToHost, h->rng_stream));
        CYMF_HIP(hipEventRecord(h->ev_sampled[b], h->rng_stream));     // d_draws[b] may be regenerated after this copy
        CYMF_HIP(hipEventRecord(h->ev_draws_host[b], h->rng_stream));
        h->exact_fetched++;
    }
    return 0;
}

// Exact mode, dataflow launch, first half (host): wait for the draws of epoch e, number the turns -- triplet l is access number ku
// of W[u], ki of H[i], kj of H[j] in sequential order -- into the pinned set of e's parity and send them up on up_stream.
// Runs while the kernel of epoch e - 1 works (cymf_bpr_epochs): that kernel reads the OTHER set.
int exact_prepare(cymf_bpr *h, int64_t e) {
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int b = (int)(e & 1);
    if (h->prep[b].epoch == e) return 0;
    const double t_entry = now();
    CYMF_TRY(exact_fetch_draws(h, e));                  // (already on its way since the previous epoch, except the first time)
    CYMF_HIP(hipEventSynchronize(h->ev_draws_host[b]));
    const uint32_t *h_draws = h->h_draws2[b];
    const int64_t N = h->N;
    const double t_draws = now();
    std::vector<uint32_t> cntW((size_t)h->U, 0u), cntH((size_t)h->I, 0u);
    h->h_last_neg.assign((size_t)N, -1);
    // (pinned staging -- common.h: PinnedBuf -- reused every second epoch: the copies of epoch e - 2 were consumed before its kernel ran)
    CYMF_TRY(h->p_tu[b].reserve((size_t)N)); CYMF_TRY(h->p_ti[b].reserve((size_t)N)); CYMF_TRY(h->p_tj[b].reserve((size_t)N));
    CYMF_TRY(h->p_ku[b].reserve((size_t)N)); CYMF_TRY(h->p_ki[b].reserve((size_t)N)); CYMF_TRY(h->p_kj[b].reserve((size_t)N));
    int32_t *const tu = h->p_tu[b].p, *const ti = h->p_ti[b].p, *const tj = h->p_tj[b].p;
    uint32_t *const ku = h->p_ku[b].p, *const ki = h->p_ki[b].p, *const kj = h->p_kj[b].p;
    int64_t n_perf = 0;
    int T = host_threads(N);
    if ((int64_t)T * ((int64_t)h->U + h->I) > ((int64_t)64 << 20)) T = (int)std::max<int64_t>(1, ((int64_t)64 << 20) / ((int64_t)h->U + h->I));
    int32_t *const last_neg = h->h_last_neg.data();
    const int32_t *const users_l = h->h_users.data(), *const items_l = h->h_pos_items.data();
    const uint32_t *const gpos = h->h_gpos.data();
    // the skip test of every triplet (bpr.pyx:166-167: a binary search in the user's positives where U x I is too large for a bitmap) has
    // no order in it: on a few host threads for large problems -- at 10^7 triplets it was a second per epoch, more than the kernel
    parallel_chunks(N, T, [&](int, int64_t lb, int64_t le) {
        for (int64_t l = lb; l < le; ++l) {
            const int32_t u = users_l[l];
            const int32_t j = (int32_t)h_draws[gpos[l]];
            const bool positive = h->h_pos_bits.empty() ? csr_has(h->h_indptr, h->h_indices, u, j)
                                                        : (h->h_pos_bits[((size_t)u * h->I + j) >> 6] >> (((size_t)u * h->I + j) & 63)) & 1;
            last_neg[l] = positive ? -1 : j;
        }
    });
    if (T <= 1) {
        for (int64_t l = 0; l < N; ++l) {
            const int32_t j = last_neg[l];
            if (j < 0) continue;
            const int32_t u = users_l[l], i = items_l[l];
            tu[n_perf] = u; ti[n_perf] = i; tj[n_perf] = j;
            ku[n_perf] = cntW[u]++; ki[n_perf] = cntH[i]++; kj[n_perf] = cntH[j]++;
            ++n_perf;
        }
    } else {
        // the turn numbers are ranks inside each row's accesses in sequential order: per-thread counts of every row over contiguous
        // chunks of the order, offsets by (row, chunk), then every chunk numbers its own triplets -- the serial numbers exactly
        const size_t Us = (size_t)h->U, Is = (size_t)h->I;
        std::vector<uint32_t> cw((size_t)T * Us, 0u), ch((size_t)T * Is, 0u);
        std::vector<int64_t> first((size_t)T + 1, 0);
        parallel_chunks(N, T, [&](int t, int64_t lb, int64_t le) {
            uint32_t *w = cw.data() + (size_t)t * Us, *hh = ch.data() + (size_t)t * Is;
            int64_t np = 0;
            for (int64_t l = lb; l < le; ++l) {
                const int32_t j = last_neg[l];
                if (j < 0) continue;
                w[users_l[l]]++; hh[items_l[l]]++; hh[j]++;
                ++np;
            }
            first[(size_t)t + 1] = np;
        });
        for (int t = 0; t < T; ++t) first[(size_t)t + 1] += first[(size_t)t];
        parallel_chunks((int64_t)Us, T, [&](int, int64_t b0, int64_t e0) {
            for (int64_t r = b0; r < e0; ++r) { uint32_t run = 0; for (int t = 0; t < T; ++t) { uint32_t &c = cw[(size_t)t * Us + (size_t)r]; const uint32_t v = c; c = run; run += v; } }
        });
        parallel_chunks((int64_t)Is, T, [&](int, int64_t b0, int64_t e0) {
            for (int64_t r = b0; r < e0; ++r) { uint32_t run = 0; for (int t = 0; t < T; ++t) { uint32_t &c = ch[(size_t)t * Is + (size_t)r]; const uint32_t v = c; c = run; run += v; } }
        });
        parallel_chunks(N, T, [&](int t, int64_t lb, int64_t le) {
            uint32_t *w = cw.data() + (size_t)t * Us, *hh = ch.data() + (size_t)t * Is;
            int64_t p = first[(size_t)t];
            for (int64_t l = lb; l < le; ++l) {
                const int32_t j = last_neg[l];
                if (j < 0) continue;
                const int32_t u = users_l[l], i = items_l[l];
                tu[p] = u; ti[p] = i; tj[p] = j;
                ku[p] = w[u]++; ki[p] = hh[i]++; kj[p] = hh[j]++;
                ++p;
            }
        });
        n_perf = first[(size_t)T];
    }
    const double t_turns = now();
    CYMF_TRY(h->d_xu[b].reserve((size_t)N)); CYMF_TRY(h->d_xi[b].reserve((size_t)N)); CYMF_TRY(h->d_xj[b].reserve((size_t)N));
    CYMF_TRY(h->d_xku[b].reserve((size_t)N)); CYMF_TRY(h->d_xki[b].reserve((size_t)N)); CYMF_TRY(h->d_xkj[b].reserve((size_t)N));
    if (n_perf > 0) {
        const size_t bytes = (size_t)n_perf * sizeof(int32_t);
        CYMF_HIP(hipMemcpyAsync(h->d_xu[b].p, tu, bytes, hipMemcpyHostToDevice, h->up_stream));
        CYMF_HIP(hipMemcpyAsync(h->d_xi[b].p, ti, bytes, hipMemcpyHostToDevice, h->up_stream));
        CYMF_HIP(hipMemcpyAsync(h->d_xj[b].p, tj, bytes, hipMemcpyHostToDevice, h->up_stream));
        CYMF_HIP(hipMemcpyAsync(h->d_xku[b].p, ku, bytes, hipMemcpyHostToDevice, h->up_stream));
        CYMF_HIP(hipMemcpyAsync(h->d_xki[b].p, ki, bytes, hipMemcpyHostToDevice, h->up_stream));
        CYMF_HIP(hipMemcpyAsync(h->d_xkj[b].p, kj, bytes, hipMemcpyHostToDevice, h->up_stream));
    }
    CYMF_HIP(hipEventRecord(h->ev_up[b], h->up_stream));
    h->prep[b].epoch = e; h->prep[b].n_perf = n_perf; h->prep[b].n_skipped = N - n_perf;
    h->prep[b].t_draws = t_draws - t_entry; h->prep[b].t_turns = t_turns - t_draws;
    return 0;
}

// ... second half: the prepared epoch as ONE dataflow launch, its loss read back (synchronises).  `next`: prepare epoch e + 1 under
// this epoch's kernel.
template <typename T>
int exact_run(cymf_bpr *h, BprStore<T> &st, bool next, double *loss_out) {
    static const bool dbg_t = getenv("CYMF_DEBUG_TIMING") != nullptr;   // host-side phases of an exact epoch to stderr
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int64_t e = h->epoch_cursor;
    const int b = (int)(e & 1);
    const double t_entry = now();
    CYMF_TRY(exact_prepare(h, e));                      // (done under the previous epoch's kernel, except for a call's first epoch)
    const double t_prep = now();
    const int64_t N = h->N, n_perf = h->prep[b].n_perf, n_skipped = h->prep[b].n_skipped;
    struct SyncOnExit { hipStream_t s, u; ~SyncOnExit() { (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(u); } } settle{h->stream, h->up_stream};   // also on error returns
    CYMF_HIP(hipStreamWaitEvent(h->stream, h->ev_up[b], 0));
    CYMF_TRY(h->d_done.alloc((size_t)h->U + (size_t)h->I));
    CYMF_TRY(h->d_done.zero(h->stream));
    CYMF_TRY(h->d_err.alloc(1));
    CYMF_TRY(h->d_err.zero(h->stream));
    CYMF_TRY(h->d_next.alloc(1));
    CYMF_TRY(h->d_next.zero(h->stream));
    CYMF_TRY(h->d_loss.zero(h->stream));
    BprDev<T> d = st.view(h->K, h->wd, h->lr);
    hipEvent_t p0 = nullptr, p1 = nullptr;
    if (h->profiling) {
        CYMF_HIP(hipEventCreate(&p0)); CYMF_HIP(hipEventCreate(&p1));
        CYMF_HIP(hipEventRecord(p0, h->stream));
    }
    if (n_perf > 0)
        CYMF_TRY(launch_ticket<T>(h->K, h->opt, d, h->d_xu[b].p, h->d_xi[b].p, h->d_xj[b].p, h->d_xku[b].p, h->d_xki[b].p, h->d_xkj[b].p, n_perf,
                                  h->d_done.p, h->d_done.p + h->U, h->d_next.p, h->d_loss.p, h->d_err.p, h->n_cu, h->stream));
    if (h->profiling) {
        CYMF_HIP(hipEventRecord(p1, h->stream));
        h->prof_events.emplace_back(p0, p1);
        h->prof_launches += 1;
        h->prof_units += n_perf;
    }
    // the next epoch's draws are generated and copied to the host now, under this epoch's kernel: a one-workgroup
    // generator on an otherwise idle GPU runs at idle clocks (measured 20-28 ms instead of 0.8 ms on some boxes)
    CYMF_TRY(exact_fetch_draws(h, e + 1));
    const double t_launched = now();
    if (next) CYMF_TRY(exact_prepare(h, e + 1));        // the host's share of epoch e + 1, under the kernel of epoch e
    const double t_next = now();
    int err = 0;
    CYMF_HIP(hipMemcpyAsync(&err, h->d_err.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    double loss = 0;
    CYMF_TRY(fetch_loss(h, &loss));
    if (dbg_t) fprintf(stderr, "[exact] epoch %lld: own schedule %.2f ms (draws %.2f, turns %.2f when it was made), memsets+launch %.2f ms, next epoch's schedule %.2f ms, "
                               "wait for kernel+readback %.2f ms\n", (long long)e, t_prep - t_entry, h->prep[b].t_draws, h->prep[b].t_turns, t_launched - t_prep,
                       t_next - t_launched, now() - t_next);
    if (err) return fail(CYMF_ERR_HIP, "exact mode: a wavefront waited past the spin limit for its turn (inconsistent schedule)");
    if (loss_out) *loss_out = N ? loss / (double)N : 0.0;   // bpr.pyx:171
    h->performed += n_perf;
    h->skipped += n_skipped;
    h->epoch_cursor++;
    return 0;
}

template <typename T>
int epoch_exact(cymf_bpr *h, BprStore<T> &st, double *loss_out, bool next = false) {
    if (h->exact_tickets) return exact_run<T>(h, st, next, loss_out);
    const int64_t e = h->epoch_cursor;
    const int b = (int)(e & 1);
    CYMF_TRY(exact_fetch_draws(h, e));                  // (already on its way since the previous epoch, except the first time)
    CYMF_HIP(hipEventSynchronize(h->ev_draws_host[b]));
    const uint32_t *h_draws = h->h_draws2[b];
    const int64_t N = h->N;
    // level scheduling (host): level(l) = 1 + max(level of the last earlier triplet touching u, i or j)
    std::vector<int32_t> lastW((size_t)h->U, 0), lastH((size_t)h->I, 0), level((size_t)N, 0);
    h->h_last_neg.assign((size_t)N, -1);
    int32_t n_levels = 0;
    int64_t n_skipped = 0;
    for (int64_t l = 0; l < N; ++l) {
        const int32_t u = h->h_users[l], i = h->h_pos_items[l];
        const int32_t j = (int32_t)h_draws[h->h_gpos[l]];
        if (csr_has(h->h_indptr, h->h_indices, u, j)) { ++n_skipped; continue; }   // bpr.pyx:166-167
        h->h_last_neg[l] = j;
        int32_t lv = std::max(lastW[u], std::max(lastH[i], lastH[j])) + 1;
        lastW[u] = lastH[i] = lastH[j] = lv;
        level[l] = lv;
        n_levels = std::max(n_levels, lv);
    }
    std::vector<int64_t> off((size_t)n_levels + 2, 0);
    for (int64_t l = 0; l < N; ++l) if (level[l]) off[level[l] + 1]++;
    for (int32_t v = 1; v <= n_levels + 1; ++v) off[v] += off[v - 1];
    const int64_t n_perf = N - n_skipped;
    std::vector<int32_t> tu((size_t)n_perf), ti((size_t)n_perf), tj((size_t)n_perf);
    {
        std::vector<int64_t> cur(off.begin(), off.end());
        for (int64_t l = 0; l < N; ++l) {
            if (!level[l]) continue;
            const int64_t p = cur[level[l]]++;
            tu[p] = h->h_users[l]; ti[p] = h->h_pos_items[l]; tj[p] = h->h_last_neg[l];
        }
    }
    CYMF_TRY(h->d_tu.upload(tu.data(), tu.size(), h->stream));
    CYMF_TRY(h->d_ti.upload(ti.data(), ti.size(), h->stream));
    CYMF_TRY(h->d_tj.upload(tj.data(), tj.size(), h->stream));
    CYMF_TRY(h->d_loss.zero(h->stream));
    BprDev<T> d = st.view(h->K, h->wd, h->lr);
    hipEvent_t p0 = nullptr, p1 = nullptr;
    if (h->profiling) {
        CYMF_HIP(hipEventCreate(&p0)); CYMF_HIP(hipEventCreate(&p1));
        CYMF_HIP(hipEventRecord(p0, h->stream));
    }
    for (int32_t lv = 1; lv <= n_levels; ++lv) {
        const int64_t b0 = off[lv], n = off[lv + 1] - off[lv];
        if (n <= 0) continue;
        launch_level<T>(h->K, h->opt, d, h->d_tu.p + b0, h->d_ti.p + b0, h->d_tj.p + b0, (int)n, h->d_loss.p, h->stream);
    }
    CYMF_HIP(hipGetLastError());
    if (h->profiling) {
        CYMF_HIP(hipEventRecord(p1, h->stream));
        h->prof_events.emplace_back(p0, p1);
        h->prof_launches += n_levels;
        h->prof_units += n_perf;
    }
    double loss = 0;
    CYMF_TRY(fetch_loss(h, &loss));
    if (loss_out) *loss_out = N ? loss / (double)N : 0.0;   // bpr.pyx:171
    h->performed += n_perf;
    h->skipped += n_skipped;
    h->epoch_cursor++;
    return 0;
}

// =============================== THROUGHPUT steps ===============================
// Negatives of epoch e: generate (rng.hip) and resolve against the users' positives, both on the
// rng stream, into the buffers of parity e & 1 -- one epoch ahead of the step kernels, so that the
// whole sampling pipeline of epoch e+1 runs concurrently with the steps of epoch e.
bool use_group_kernel(const cymf_bpr *h);
// small epochs on the group kernel: the negatives are resolved inside it (BprGroupSample), no sampling kernel, no second stream
// between the generator and the steps
bool fused_sampling(const cymf_bpr *h) { return h->draw_batch > 1 && use_group_kernel(h); }

int prepare_epoch(cymf_bpr *h, int64_t e) {
    while (h->epochs_sampled <= e) {
        const int64_t g = h->epochs_sampled;
        const int b = (int)(g & 1);
        CYMF_TRY(request_epoch_draws(h, g));
        const int64_t E = h->draw_batch;
        if (fused_sampling(h)) {   // the group kernel resolves the negatives itself: nothing to do on the side stream
            h->epochs_sampled++;     // (the NEXT batch of draws is requested by run_one_step, once the buffer it goes to is free)
            continue;
        }
        if (E > 1) CYMF_HIP(hipStreamWaitEvent(h->rng_stream, h->ev_gen[(int)((g / E) & 1)], 0));
        // slot_neg[b] was last read by the steps of epoch g-2
        if (g >= 2) CYMF_HIP(hipStreamWaitEvent(h->rng_stream, h->ev_epoch_done[b], 0));
        if (h->N > 0) {
            int blocks = (int)std::min<int64_t>((h->N + 255) / 256, 256 * 16);
            hipLaunchKernelGGL(bpr_sample_kernel, dim3(blocks), dim3(256), 0, h->rng_stream, h->d_slot_user.p, h->d_slot_pos.p,
                               epoch_draws(h, g), h->d_pair_table.p, h->pair_mask, h->d_hot_bits.p, h->d_slot_neg[b].p, h->N,
                               h->d_skipped.p);
            CYMF_HIP(hipGetLastError());
        }
        CYMF_HIP(hipEventRecord(h->ev_sampled[b], h->rng_stream));
        if (E > 1 && g % E == E - 1) CYMF_HIP(hipEventRecord(h->ev_batch_sampled[(int)((g / E) & 1)], h->rng_stream));
        if (E > 1 && g % E == 0) CYMF_TRY(request_epoch_draws(h, g + E));   // the NEXT batch, beside this batch's epochs
        h->epochs_sampled++;
    }
    return 0;
}

int ensure_epoch_sampled(cymf_bpr *h) {
    if (h->epoch_sampled) return 0;
    const int64_t e = h->epoch_cursor;
    CYMF_TRY(prepare_epoch(h, e));
    if (fused_sampling(h)) CYMF_HIP(hipStreamWaitEvent(h->stream, h->ev_gen[(int)((e / h->draw_batch) & 1)], 0));   // the epoch's draws exist
    else CYMF_HIP(hipStreamWaitEvent(h->stream, h->ev_sampled[(int)(e & 1)], 0));
    h->epoch_sampled = true;
    CYMF_TRY(prepare_epoch(h, e + 1));   // next epoch's negatives, concurrently with this epoch's steps
    return 0;
}

// AdaGrad / Adam: an item run is owned by ONE wavefront (its optimizer state cannot be merged, see the
// kernel).  Greedy cut of every step's item-sorted slots into ranges of about slots/waves, moved to
// the next item boundary; a run longer than the target becomes a range of its own (the step's tail).
int ensure_wave_ranges(cymf_bpr *h, int64_t waves_target) {
    if (h->wave_ranges_for == waves_target && h->d_wave_ranges.p) return 0;
    std::vector<int64_t> all;
    h->wave_range_off.assign((size_t)h->steps_per_epoch + 1, 0);
    for (int32_t s = 0; s < h->steps_per_epoch; ++s) {
        const int64_t b = h->step_off[s], e = h->step_off[s + 1];
        h->wave_range_off[s] = (int64_t)all.size();
        const int64_t target = std::max<int64_t>(64, (e - b + waves_target - 1) / std::max<int64_t>(waves_target, 1));
        int64_t start = b;
        all.push_back(b);
        while (start < e) {
            int64_t cut = std::min(e, start + target);
            if (cut < e) {   // advance to the end of the item run that contains slot cut-1
                const int32_t item = h->h_slot_item[(size_t)cut - 1];
                while (cut < e && h->h_slot_item[(size_t)cut] == item) ++cut;
            }
            all.push_back(cut);
            start = cut;
        }
    }
    h->wave_range_off[(size_t)h->steps_per_epoch] = (int64_t)all.size();
    CYMF_TRY(h->d_wave_ranges.upload(all.data(), all.size(), h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    h->wave_ranges_for = waves_target;
    return 0;
}

// Small tables run the group kernel (bpr_groups.hip: four triplets per wavefront, every write-back an atomic delta).  Chosen
// when the step kernel's bound on the rows in flight would leave it a fraction of the wavefronts the chip holds (C2: 94 of
// 3 072) and the atomics' bytes are small (both tables inside 64 MB); never under a communicator (the multi-GPU exchange is
// built around the step kernel's item-sorted steps).
bool group_kernel_fits(const cymf_bpr *h) {   // by shape alone (what the choice of steps_per_epoch is based on)
    if (h->comm || h->dtype != CYMF_F32 || !bpr_group_supported(h->K)) return false;
    if (h->group_mode >= 0) return h->group_mode == 1;
    const int64_t rpi = (int64_t)h->rows_per_inflight * (h->opt == CYMF_OPT_SGD ? 1 : h->adaptive_rpi_factor);
    const int64_t by_rows = std::max<int64_t>(1, std::min<int64_t>(h->I / (rpi * 4), h->U / (rpi * 4)));
    return by_rows * 2 < h->max_waves && ((int64_t)h->U + h->I) * h->K * (int64_t)sizeof(float) <= ((int64_t)64 << 20);
}
// ... and by the windows the caller chose: the groups work a window's run of one item concurrently, from about the same value of
// its row; a long run is a mini-batch large enough to overshoot (C2 SGD, hottest item's run per window 5 858 slots: the loss
// diverges; 1 464: norm of H +7 %; 366: -4.5 %; Adam, whose step does not shrink with the gradient, loses its way at a few
// hundred) -- such a layout stays on the step kernel, whose few wavefronts walk a run sequentially.  The automatic
// steps_per_epoch keeps the run at 128 slots or fewer.
bool use_group_kernel(const cymf_bpr *h) {
    if (!group_kernel_fits(h)) return false;
    if (h->group_mode == 1) return true;
    return h->f_item_max * (double)h->N <= (h->opt == CYMF_OPT_ADAM ? 128.0 : 512.0) * (double)h->steps_per_epoch;
}

// one step -- or, with the group kernel, `fuse` consecutive steps of the epoch in ONE launch: its groups walk the (step, item)-
// sorted slots interleaved, so the launch moves through the steps' windows in order by itself
int run_one_step(cymf_bpr *h, int32_t fuse = 1) {
    CYMF_TRY(ensure_epoch_sampled(h));
    const int32_t s = h->step_cursor;
    const bool groups = use_group_kernel(h);
    if (!groups) fuse = 1;
    fuse = std::max<int32_t>(1, std::min<int32_t>(fuse, h->steps_per_epoch - s));
    const int64_t b = h->step_off[s], e = h->step_off[s + fuse];
    BprDev<float> d = h->f32.view(h->K, h->wd, h->lr);
    if (e > b && groups) {
        hipEvent_t p0 = nullptr, p1 = nullptr;
        if (h->profiling) {
            if (h->prof_pool.size() >= 2) {
                p0 = h->prof_pool.back(); h->prof_pool.pop_back();
                p1 = h->prof_pool.back(); h->prof_pool.pop_back();
            } else {
                CYMF_HIP(hipEventCreate(&p0)); CYMF_HIP(hipEventCreate(&p1));
            }
            CYMF_HIP(hipEventRecord(p0, h->stream));
        }
        BprGroupDev gd{d.W, d.H, d.W0, d.W1, d.H0, d.H1, d.K, d.wd, d.opt};
        // wavefronts = triplets in flight (4 groups x 16-slot blocks each): one or two per CU reach the atomics' rate for SGD / AdaGrad
        // (C2: 128 / 256 / 512 / 1 024 wavefronts 0.47 / 0.27 / 0.24 / 0.25 ms per epoch);
        // small data sets get fewer (the front should stay a few percent of an epoch), and Adam -- whose first moment remembers
        // ten updates -- about one percent of an epoch (measurements: bpr_groups.hip header)
        const int64_t slots_per_wave = h->opt == CYMF_OPT_ADAM ? 6400 : 1024;   // (C2 Adam: 73 wavefronts, ~1 % of an epoch in flight)
        const int n_waves = h->group_waves > 0 ? h->group_waves
                                               : (int)std::max<int64_t>(1, std::min<int64_t>(512, h->N / slots_per_wave));
        BprGroupSample smp;
        if (fused_sampling(h)) {
            smp.slot_pos = h->d_slot_pos.p; smp.draws = epoch_draws(h, h->epoch_cursor); smp.table = h->d_pair_table.p;
            smp.mask = h->pair_mask; smp.slot_neg_out = h->d_slot_neg[(int)(h->epoch_cursor & 1)].p;
        }
        CYMF_TRY(bpr_group_launch(h->opt, gd, h->d_slot_user.p, h->d_slot_item.p, h->d_slot_neg[(int)(h->epoch_cursor & 1)].p, b, e, n_waves,
                                  h->d_loss.p, h->d_performed.p, h->stream, smp));
        if (h->profiling) {
            CYMF_HIP(hipEventRecord(p1, h->stream));
            h->prof_events.emplace_back(p0, p1);
            h->prof_launches += 1;
            h->prof_units += e - b;
        }
        h->slots_done += e - b;
    } else if (e > b) {
        if (h->comm) {   // the measured data term of this step's factors: two floats at the tail of the message that leaves after the step
            const int64_t n = (int64_t)h->I * h->K;
            float *tail = (h->overlap_exchange ? h->d_local[(int)(h->exch_count & 1)].p : h->d_delta.p) + n;
            CYMF_HIP(hipMemsetAsync(tail, 0, DELTA_TAIL * sizeof(float), h->stream));
            if (h->rho_lr_data > 0.0) {
                const int64_t n_samples = std::min<int64_t>(e - b, 2048);   // a mean over 2 048 triplets: a few per cent of noise on a damping factor
                hipLaunchKernelGGL(bpr_curvature_kernel, dim3((unsigned)((n_samples + 31) / 32)), dim3(256), 0, h->stream, h->f32.W.p, h->f32.H.p, h->K,
                                   h->d_slot_user.p, h->d_slot_item.p, h->d_slot_neg[(int)(h->epoch_cursor & 1)].p, b, e, n_samples, tail);
                CYMF_HIP(hipGetLastError());
            }
        }
        const int64_t chunks = (e - b + 63) / 64;
        // Number of wavefronts = bounded staleness (see the kernel header): at most `max_waves` from
        // the hardware side, and few enough that the rows in flight (waves * PF) stay a small
        // fraction of the smaller table, so that two waves rarely hold the same row at once.
        int64_t waves = std::min<int64_t>(chunks, h->max_waves);
        // adaptive optimizers carry per-row state whose read-modify-write is not atomic: 4x stricter
        const int64_t rpi = (int64_t)h->rows_per_inflight * (h->opt == CYMF_OPT_SGD ? 1 : h->adaptive_rpi_factor);
        // Small tables (the staleness bound, not the chip, limits the wavefronts): the instantiation with rings of 4 instead of 8
        // rows admits twice the wavefronts at the same number of rows in flight, and four triplets of lead still cover a
        // memory latency (C2, 6040 x 3706: 1.61 -> 1.36 ms per epoch, same loss trajectory).  CYMF_BPR_NARROW=0/1 overrides.
        bool narrow = h->K > 128 || h->opt == CYMF_OPT_ADAM;
        if (!narrow) {
            const int64_t by_rows8 = std::max<int64_t>(1, std::min<int64_t>(h->I / (rpi * 8), h->U / (rpi * 8)));
            narrow = by_rows8 < waves;
            if (const char *en = getenv("CYMF_BPR_NARROW")) narrow = en[0] == '1';
        }
        const bool sgd = h->opt == CYMF_OPT_SGD;   // only the SGD instantiations use asymmetric leads (launch_step_pf)
        const int pf_w = narrow ? 4 : (sgd && h->step_pf > 100 ? 16 : 8);                          // lead of the user-row ring
        const int pf_j = narrow ? 4 : (sgd && h->step_pf > 10 ? h->step_pf % 10 : 8);              // lead of the negative-row ring
        const int64_t by_rows = std::max<int64_t>(1, std::min<int64_t>(h->I / (rpi * pf_j), h->U / (rpi * pf_w)));
        waves = std::max<int64_t>(1, std::min(waves, by_rows));
        // delta-sum exchange of a run shared by c = waves * f_max wavefronts contracts by (1 - lr*wd)^32 per wave and
        // interval: keep c * (1 - (1 - lr*wd)^32) <= 1/2 (measured: 0.49 trains like the sequential run, 0.65 degrades)
        if (h->opt == CYMF_OPT_SGD && h->f_item_max > 0 && h->lr * h->wd > 0) {
            const double shrink = 1.0 - std::pow(1.0 - std::min(h->lr * h->wd, 0.5), 32.0);
            waves = std::max<int64_t>(1, std::min<int64_t>(waves, (int64_t)(0.5 / (h->f_item_max * shrink))));
        }
        int64_t cpw = (chunks + waves - 1) / waves;
        waves = (chunks + cpw - 1) / cpw;
        const int64_t *wave_ranges = nullptr;
        if (h->item_aligned && h->opt != CYMF_OPT_SGD) {   // item-aligned ranges (experiment: single-wave runs are tail-bound)
            CYMF_TRY(ensure_wave_ranges(h, waves));
            wave_ranges = h->d_wave_ranges.p + h->wave_range_off[s];
            waves = h->wave_range_off[s + 1] - h->wave_range_off[s] - 1;
        }
        const int grid = (int)((waves + 3) / 4);
        hipEvent_t p0 = nullptr, p1 = nullptr;
        if (h->profiling) {
            if (h->prof_pool.size() >= 2) {
                p0 = h->prof_pool.back(); h->prof_pool.pop_back();
                p1 = h->prof_pool.back(); h->prof_pool.pop_back();
            } else {
                CYMF_HIP(hipEventCreate(&p0)); CYMF_HIP(hipEventCreate(&p1));
            }
            CYMF_HIP(hipEventRecord(p0, h->stream));
        }
        if (h->K > 256)   // any num_components (cymf/bpr.pyx:50): the two-pass kernel, item deltas as atomics
            launch_wide<float, true>(h->opt, d, h->d_slot_user.p + b, h->d_slot_item.p + b, h->d_slot_neg[(int)(h->epoch_cursor & 1)].p + b,
                                     e - b, h->d_loss.p, h->d_performed.p, std::min<int64_t>(waves, e - b), h->stream);
        else
            launch_step(h->K, h->opt, d, h->d_slot_user.p, h->d_slot_item.p, h->d_slot_neg[(int)(h->epoch_cursor & 1)].p, b, e, cpw, h->xcd_stride, wave_ranges, waves, h->d_loss.p, h->d_performed.p, grid, h->step_pf, h->stream);
        CYMF_HIP(hipGetLastError());
        if (h->profiling) {
            CYMF_HIP(hipEventRecord(p1, h->stream));
            h->prof_events.emplace_back(p0, p1);
            h->prof_launches += 1;
            h->prof_units += e - b;
        }
        h->slots_done += e - b;
    }
    if (h->comm && h->overlap_exchange) {
        // delta of this step against the reference point; the exchange of the PREVIOUS step has had this step's
        // kernel to finish: its damped sum replaces the local delta that H already carries; then this step's
        // delta goes out on the communication stream while the next step computes
        const int64_t n = (int64_t)h->I * h->K;
        const int b = (int)(h->exch_count & 1);
        hipLaunchKernelGGL(delta_kernel, dim3(ew_blocks(n)), dim3(256), 0, h->stream, h->f32.H.p, h->d_snap.p, h->d_local[b].p, n);
        CYMF_HIP(hipGetLastError());
        CYMF_TRY(finish_exchange(h, /*snapshot_if_idle=*/true, h->d_local[b].p));
        CYMF_HIP(hipEventRecord(h->ev_delta_ready, h->stream));
        CYMF_HIP(hipStreamWaitEvent(h->comm_stream, h->ev_delta_ready, 0));
        CYMF_TRY(comm_allreduce_sum_f32_to(h->comm, h->d_local[b].p, h->d_glob[b].p, n + DELTA_TAIL, (int64_t)std::min(h->d_local[b].n, h->d_glob[b].n), h->comm_stream));
        CYMF_HIP(hipEventRecord(h->ev_reduced[b], h->comm_stream));
        h->exch_pending = true;
        h->exch_parity = b;
        h->exch_step = s;
        h->exch_count++;
    } else if (h->comm) {   // sum of the ranks' item-factor deltas
        const int64_t n = (int64_t)h->I * h->K;
        hipLaunchKernelGGL(delta_kernel, dim3(ew_blocks(n)), dim3(256), 0, h->stream, h->f32.H.p, h->d_snap.p, h->d_delta.p, n);
        CYMF_HIP(hipGetLastError());
        CYMF_TRY(comm_allreduce_sum_f32(h->comm, h->d_delta.p, n + DELTA_TAIL, h->stream));
        hipLaunchKernelGGL(delta_scale_kernel, dim3((h->I + 255) / 256), dim3(256), 0, h->stream, h->d_step_counts.p + (size_t)s * h->I, h->I,
                           comm_world(h->comm), (float)h->rho_fixed, (float)h->rho_lr_data, h->d_delta.p + n, h->d_delta_scale.p);
        hipLaunchKernelGGL(apply_delta_kernel, dim3(ew_blocks(n)), dim3(256), 0, h->stream, h->f32.H.p, h->d_snap.p, h->d_delta.p,
                           h->d_delta_scale.p, h->K, n);
        CYMF_HIP(hipGetLastError());
    }
    h->step_cursor += fuse;
    if (h->step_cursor >= h->steps_per_epoch) {
        CYMF_HIP(hipEventRecord(h->ev_epoch_done[(int)(h->epoch_cursor & 1)], h->stream));
        if (fused_sampling(h)) {
            const int64_t ec = h->epoch_cursor, E = h->draw_batch;
            if (ec % E == E - 1)   // this batch of draws has been consumed: its buffer may be regenerated behind this event
                CYMF_HIP(hipEventRecord(h->ev_batch_sampled[(int)((ec / E) & 1)], h->stream));
            // first epoch of a batch done: generate the NEXT batch beside this batch's remaining epochs.  Its buffer was last read by
            // the previous batch, whose event was recorded (above) before this epoch was launched.
            if (ec % E == 0) CYMF_TRY(request_epoch_draws(h, ec + E));
        }
        h->step_cursor = 0;
        h->epoch_cursor++;
        h->epoch_sampled = false;
    }
    return 0;
}

int collect_skips(cymf_bpr *h) {   // after a stream sync: performed comes from the step kernels
    unsigned long long p = 0;
    CYMF_HIP(hipMemcpyAsync(&p, h->d_performed.p, sizeof p, hipMemcpyDeviceToHost, h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    h->performed = (int64_t)p;
    h->skipped = h->slots_done - h->performed;
    return 0;
}

// Multi-GPU: every rank trains its replica of H for one step and the ranks' deltas are summed.  A plain
// sum is only right while an item's row moves little within the step; a popular item's replica runs
// through n/N updates per rank, each contracting it by (1 - lr*wd) towards a local equilibrium, and N
// such deltas summed overshoot by up to N (gain -(N-1) per step: divergence, reproduced on the CPU in
// tests/test_dist_gloo.py).  Modelling a rank's step on row i as the affine map h -> a h + b with
// a = (1 - lr*wd')^(n_i/N), composing the N maps sequentially instead of adding them gives
//     h_new = snap + s_i * sum_r delta_r,   s_i = (1 - a^N) / (N (1 - a))
// (s -> 1 for rarely touched rows = plain sum; s -> 1/N for hot rows = average of the replicas).
// n_i = positives of item i in the step over all ranks (all-reduced once here) + its expected share
// of the uniform negatives; wd' = 2 wd leaves room for the curvature of the data term.  The optimizer state
// (AdaGrad accumulators, Adam moments) of the item rows stays private to the rank, like W: only H is exchanged.
int build_step_counts(cymf_bpr *h, const std::vector<int32_t> &slot_item) {
    const int32_t S = h->steps_per_epoch;
    std::vector<float> cnt((size_t)S * h->I, 0.0f);
    for (int32_t s = 0; s < S; ++s)
        for (int64_t t = h->step_off[s]; t < h->step_off[s + 1]; ++t) cnt[(size_t)s * h->I + slot_item[(size_t)t]] += 1.0f;
    std::vector<float> slots((size_t)S, 0.0f);
    for (int32_t s = 0; s < S; ++s) slots[s] = (float)(h->step_off[s + 1] - h->step_off[s]);
    {   // global counts: exact in float32 below 2^24 per item and step
        DevBuf<float> d;
        CYMF_TRY(d.upload(cnt.data(), cnt.size(), h->stream));
        CYMF_HIP(hipStreamSynchronize(h->stream));
        CYMF_TRY(comm_allreduce_sum_f32(h->comm, d.p, (int64_t)cnt.size(), h->stream));
        CYMF_HIP(hipMemcpyAsync(cnt.data(), d.p, cnt.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        DevBuf<float> d2;
        CYMF_TRY(d2.upload(slots.data(), slots.size(), h->stream));
        CYMF_TRY(comm_allreduce_sum_f32(h->comm, d2.p, (int64_t)slots.size(), h->stream));
        CYMF_HIP(hipMemcpyAsync(slots.data(), d2.p, slots.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        CYMF_HIP(hipStreamSynchronize(h->stream));
    }
    // n_i = positives of item i in the step over all ranks + its expected share of the step's uniform negatives
    for (int32_t s = 0; s < S; ++s)
        for (int32_t i = 0; i < h->I; ++i) cnt[(size_t)s * h->I + i] += slots[s] / (float)h->I;
    CYMF_TRY(h->d_step_counts.upload(cnt.data(), cnt.size(), h->stream));
    CYMF_TRY(h->d_delta_scale.alloc((size_t)h->I));
    // Per-touch contraction rho of a replica towards its local equilibrium.  SGD / AdaGrad: the weight decay contracts by lr wd per
    // touch (x2: head-room), the data term by lr sigma'(x) |w|^2 along the user row it meets -- MEASURED at the start of every step
    // (bpr_curvature_kernel) and summed over the ranks with the deltas.  Round 2 used the constant lr / 5 fitted on one
    // 200k x 20k problem: with the tiny initial factors (|w|^2 ~ 1e-5) nothing contracts, and damping the first steps' sums as
    // if it did is what left eight ranks 24 % behind after the first epoch at C3's full size; as the factors grow the measured
    // term takes over (C3 after three epochs: ~0.3 lr).  Adam moves every element by about lr per touch whatever the gradient:
    // 5 lr (CPU emulation, tests/test_dist_gloo.py), no data term.  CYMF_BPR_DELTA_RHO fixes rho (experiments).
    h->rho_fixed = 2.0 * h->lr * h->wd;
    h->rho_lr_data = h->lr;
    if (h->opt == CYMF_OPT_ADAM) { h->rho_fixed = 5.0 * h->lr; h->rho_lr_data = 0.0; }
    if (const char *er = getenv("CYMF_BPR_DELTA_RHO")) { h->rho_fixed = atof(er); h->rho_lr_data = 0.0; }
    if (const char *ed = getenv("CYMF_BPR_DELTA_DATA")) h->rho_lr_data = h->lr * atof(ed);   // multiple of the measured term (experiments)
    CYMF_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

bool group_kernel_fits(const cymf_bpr *h);

// steps_per_epoch = 0 ("auto").  The reference trains in the shuffled order (cymf/bpr.pyx:104,162-169); a step is a window of that
// order inside which the triplets are bucketed by positive item, so the more windows, the closer to the reference's order.
//  * group kernel (small tables): all of a window's triplets of one item are worked at about the same time, from about the same
//    value of its row -- the hottest item's run per window is what has to stay small.  Measured on C2 against the sequential
//    oracle in the reference's order (tools/order_fidelity.py): runs of 5 858 / 1 464 slots diverge /
//    cost 7 % of the norm of H, 366 slots 4.5 %, 91 slots 0.6 % -> at most 128 slots of the hottest item per window, at least 16
//    windows, windows of at least 2 048 slots.
//  * step kernel (large tables): windows of ~2.5 M triplets of the global order (a sharded job: ~5 M, the same on every rank);
//    never fewer than four (see below).  Measured at C3's full size against the sequential oracle in the reference's order, three
//    epochs of SGD (profiles/r03_c3_order_fidelity.md): windows of 5 M / 2 M / 1 M triplets end at loss -4.9 / -1.6 / -0.7 %,
//    held-out Recall@5 -0.0095 / -0.0023 / -0.0009, and cost 0 / 4 / 11 % of the throughput (a window's runs of one item
//    shorten, H[i] is forwarded less) -- 2.5 M keeps Recall@5 and the loss well inside the bars of DESIGN.md section 4 for 3 %.
int32_t choose_steps_per_epoch(const cymf_bpr *h, int64_t hottest_item_count) {
    const int64_t n_global = std::max<int64_t>(h->N_global, 1);
    if (group_kernel_fits(h)) {
        int64_t S = 16;
        while (S < 4096 && hottest_item_count > 128 * S) S *= 2;
        while (S > 1 && n_global / S < 2048) S /= 2;
        return (int32_t)S;
    }
    // ... and at least four of them while a window still holds a quarter of a million triplets: from four windows on the sequential
    // oracle over the bucketed order is indistinguishable from the shuffled order (norm of H 93.1 against 95.3; one
    // window: 72.8)
    // (a sharded job: a window ends with the exchange of the item deltas -- few, large ones)
    const int64_t win = h->comm ? 5000000 : 2500000;
    int64_t S = std::max<int64_t>(4, std::min<int64_t>(4096, (n_global + win / 2) / win));
    while (S > 1 && n_global / S < 262144) S /= 2;
    return (int32_t)S;
}

int build_throughput_layout(cymf_bpr *h) {
    static const bool dbg_t = getenv("CYMF_DEBUG_TIMING") != nullptr;   // host-side phases to stderr
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) { if (dbg_t) { const double t = now(); fprintf(stderr, "[layout] %s %.0f ms\n", what, t - t_prev); t_prev = t; } };
    const int64_t N = h->N;
    if (h->steps_auto) {
        std::vector<int64_t> c((size_t)h->I, 0);
        for (int64_t l = 0; l < N; ++l) c[(size_t)h->h_pos_items[l]]++;
        int64_t mx = 0;
        for (int64_t v : c) mx = std::max(mx, v);
        h->steps_per_epoch = choose_steps_per_epoch(h, mx);
    }
    lap("steps_per_epoch");
    const int32_t S = h->steps_per_epoch;
    // step of a triplet = window of the GLOBAL order it falls into, floor(g S / N_global) -- by the windows' first positions, not by a
    // 128-bit division per triplet; slots sorted by (step, item), triplets of one item in their given order.
    // (10^8 triplets: the passes below are two counting sorts run on a few host threads -- per-thread histograms, offsets by
    // (bucket, thread), scatter -- which produce exactly the serial order; one thread took 3.4 s here, the scatter's cache misses.)
    const int64_t Ng = h->N_global > 0 ? h->N_global : 1;
    std::vector<int64_t> bound((size_t)S + 2);
    for (int32_t v = 0; v <= S; ++v) bound[(size_t)v] = (int64_t)(((__int128)v * Ng + S - 1) / S);   // first position of window v
    bound[(size_t)S + 1] = INT64_MAX;
    const double inv = (double)S / (double)Ng;
    auto step_of_pos = [&](int64_t g) -> int32_t {
        int32_t v = (int32_t)((double)g * inv);
        v = v < 0 ? 0 : (v > S - 1 ? S - 1 : v);
        while (g < bound[(size_t)v]) --v;
        while (v + 1 < S && g >= bound[(size_t)v + 1]) ++v;
        return v;
    };
    const int32_t *const pos_items = h->h_pos_items.data();
    const int32_t *const users_l = h->h_users.data();
    const uint32_t *const gpos = h->h_gpos.data();
    const int32_t I = h->I;
    int T = host_threads(N);
    if ((int64_t)T * I > ((int64_t)32 << 20)) T = (int)std::max<int64_t>(1, ((int64_t)32 << 20) / std::max(I, 1));
    // ---- by item: counting sort of the triplets by positive item
    std::vector<uint32_t> hist((size_t)T * (size_t)I, 0u);
    parallel_chunks(N, T, [&](int t, int64_t b, int64_t e) {
        uint32_t *hs = hist.data() + (size_t)t * I;
        for (int64_t l = b; l < e; ++l) hs[pos_items[l]]++;
    });
    std::vector<int64_t> cnt((size_t)I + 1, 0);   // cnt[i]: first place of item i in the by-item order
    for (int32_t i = 0; i < I; ++i) {
        int64_t n_i = 0;
        for (int t = 0; t < T; ++t) { const uint32_t c = hist[(size_t)t * I + i]; hist[(size_t)t * I + i] = (uint32_t)n_i; n_i += c; }   // -> offset of thread t inside the bucket
        cnt[(size_t)i + 1] = cnt[(size_t)i] + n_i;
    }
    {
        int64_t mx = 0;
        for (int32_t i = 0; i < I; ++i) mx = std::max(mx, cnt[(size_t)i + 1] - cnt[(size_t)i]);
        h->f_item_max = N > 0 ? (double)mx / (double)N : 0.0;
    }
    {   // hot items: at least hot_threshold positives per step
        std::vector<uint32_t> bits(((size_t)I + 31) / 32, 0u);
        for (int32_t i = 0; i < I; ++i)
            if (cnt[(size_t)i + 1] - cnt[(size_t)i] >= (int64_t)h->hot_threshold * S) bits[i >> 5] |= 1u << (i & 31);
        CYMF_TRY(h->d_hot_bits.upload(bits.data(), bits.size(), h->stream));
    }
    lap("item counts, hot bits");
    std::vector<uint32_t> by_item((size_t)N);
    parallel_chunks(N, T, [&](int t, int64_t b, int64_t e) {
        uint32_t *hs = hist.data() + (size_t)t * I;
        for (int64_t l = b; l < e; ++l) {
            const int32_t i = pos_items[l];
            by_item[(size_t)(cnt[(size_t)i] + hs[i]++)] = (uint32_t)l;
        }
    });
    { std::vector<uint32_t>().swap(hist); }
    lap("by_item");
    // ---- by (step, item): counting sort of the by-item order by window
    const int T2 = host_threads(N);
    const bool ident = h->gpos_identity;                      // one rank: triplet l is position l of the global order (no gather of it)
    std::vector<uint16_t> stepq(S <= 65535 ? (size_t)N : 0);  // the window of every place of the by-item order, found once
    std::vector<int64_t> shist((size_t)T2 * (size_t)S, 0);
    parallel_chunks(N, T2, [&](int t, int64_t b, int64_t e) {
        int64_t *hs = shist.data() + (size_t)t * S;
        for (int64_t q = b; q < e; ++q) {
            const uint32_t l = by_item[(size_t)q];
            const int32_t v = step_of_pos(ident ? (int64_t)l : (int64_t)gpos[l]);
            if (!stepq.empty()) stepq[(size_t)q] = (uint16_t)v;
            hs[v]++;
        }
    });
    h->step_off.assign((size_t)S + 1, 0);
    for (int32_t v = 0; v < S; ++v) {
        int64_t n_v = 0;
        for (int t = 0; t < T2; ++t) { const int64_t c = shist[(size_t)t * S + v]; shist[(size_t)t * S + v] = n_v; n_v += c; }
        h->step_off[(size_t)v + 1] = h->step_off[(size_t)v] + n_v;
    }
    lap("step offsets");
    std::vector<int32_t> su((size_t)N), si((size_t)N);
    std::vector<uint32_t> sp((size_t)N), sl((size_t)N);
    parallel_chunks(N, T2, [&](int t, int64_t b, int64_t e) {
        std::vector<int64_t> cur((size_t)S);
        for (int32_t v = 0; v < S; ++v) cur[(size_t)v] = h->step_off[(size_t)v] + shist[(size_t)t * S + v];
        int32_t item = (int32_t)(std::upper_bound(cnt.begin(), cnt.end(), b) - cnt.begin()) - 1;   // the bucket place b lies in
        for (int64_t q = b; q < e; ++q) {   // stable: item order inside each step
            while (q >= cnt[(size_t)item + 1]) ++item;
            const uint32_t l = by_item[(size_t)q];
            const uint32_t g = ident ? l : gpos[l];
            const int64_t p = cur[stepq.empty() ? (size_t)step_of_pos((int64_t)g) : (size_t)stepq[(size_t)q]]++;
            su[(size_t)p] = users_l[l]; si[(size_t)p] = item; sp[(size_t)p] = g; sl[(size_t)p] = l;
        }
    });
    lap("slot arrays");
    CYMF_TRY(h->d_slot_user.upload(su.data(), su.size(), h->stream));
    CYMF_TRY(h->d_slot_item.upload(si.data(), si.size(), h->stream));
    if (h->item_aligned && h->opt != CYMF_OPT_SGD) h->h_slot_item = si;
    if (h->comm) CYMF_TRY(build_step_counts(h, si));
    h->wave_ranges_for = -1;
    CYMF_TRY(h->d_slot_pos.upload(sp.data(), sp.size(), h->stream));
    CYMF_TRY(h->d_slot_local.upload(sl.data(), sl.size(), h->stream));
    lap("uploads");
    {   // membership table over the CSR pattern (device build, a few ms)
        const int64_t nnz = (int64_t)h->h_indices.size();
        unsigned long long cap = 1024;
        while (cap < 2ull * (unsigned long long)nnz) cap <<= 1;
        h->pair_mask = cap - 1;
        CYMF_TRY(h->d_pair_table.alloc((size_t)cap));
        CYMF_HIP(hipMemsetAsync(h->d_pair_table.p, 0xff, (size_t)cap * sizeof(unsigned long long), h->stream));
        hipLaunchKernelGGL(pair_table_build_kernel, dim3(256 * 8), dim3(256), 0, h->stream, h->d_indptr.p, h->d_indices.p, h->U,
                           h->d_pair_table.p, h->pair_mask);
        CYMF_HIP(hipGetLastError());
    }
    CYMF_TRY(h->d_slot_neg[0].alloc((size_t)N));
    CYMF_TRY(h->d_slot_neg[1].alloc((size_t)N));
    CYMF_TRY(h->d_skipped.alloc(1));
    CYMF_TRY(h->d_skipped.zero(h->stream));
    CYMF_TRY(h->d_performed.alloc(1));
    CYMF_TRY(h->d_performed.zero(h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    lap("pair table, buffers");
    return 0;
}

int drain_profile(cymf_bpr *h) {
    for (auto &pe : h->prof_events) {
        float ms = 0;
        CYMF_HIP(hipEventSynchronize(pe.second));
        CYMF_HIP(hipEventElapsedTime(&ms, pe.first, pe.second));
        h->prof_ms += ms;
        h->prof_pool.push_back(pe.first);
        h->prof_pool.push_back(pe.second);
    }
    h->prof_events.clear();
    return 0;
}

}  // namespace

// =====================================================================================
//                                        C ABI
// =====================================================================================
extern "C" int cymf_bpr_create(cymf_bpr **out, int32_t U, int32_t I, int32_t K, int optimizer, double learning_rate,
                               double weight_decay, uint32_t neg_seed, int dtype, int mode, int device) {
    if (!out) return fail(CYMF_ERR_INVALID, "cymf_bpr_create: out is NULL");
    *out = nullptr;
    if (U <= 0 || I <= 0 || K <= 0) return fail(CYMF_ERR_INVALID, "cymf_bpr_create: U, I, K must be positive");
    if (optimizer < 0 || optimizer > 2) return fail(CYMF_ERR_INVALID, "cymf_bpr_create: optimizer id %d", optimizer);
    if (dtype != CYMF_F32 && dtype != CYMF_F64) return fail(CYMF_ERR_INVALID, "cymf_bpr_create: dtype %d", dtype);
    if (mode != CYMF_MODE_EXACT && mode != CYMF_MODE_THROUGHPUT) return fail(CYMF_ERR_INVALID, "cymf_bpr_create: mode %d", mode);
    if (mode == CYMF_MODE_THROUGHPUT && dtype != CYMF_F32)
        return fail(CYMF_ERR_UNSUPPORTED, "throughput mode computes in f32 only (f64 is the exact-order parity path)");
    CYMF_TRY(use_device(device));
    cymf_bpr *h = new cymf_bpr();
    h->U = U; h->I = I; h->K = K; h->opt = optimizer; h->lr = learning_rate; h->wd = weight_decay;
    h->seed = neg_seed; h->dtype = dtype; h->mode = mode; h->device = device;
    // Stream priorities, not for the scheduling order but for the hardware queues: the runtime deals a process's streams onto a
    // handful of hardware queues, and two streams that land on ONE queue run strictly one after the other.  Measured in the
    // eight-rank schedule (rocprofv3 kernel trace): the side stream (index stream, skip tests) shared a queue with the step
    // stream and every "hidden" side kernel sat between two steps -- 0.5 ms of a 1.0 ms step.  Streams of different priority
    // never share a queue: steps normal, side work lowest, the exchange highest.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    h->prio_side = prio_least;
    h->prio_comm = prio_greatest;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&h->rng_stream, hipStreamNonBlocking, h->prio_side);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        e = hipEventCreateWithFlags(&h->ev_gen[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_sampled[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_epoch_done[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_draws_host[b], hipEventDisableTiming);
        if (e == hipSuccess && mode == CYMF_MODE_EXACT) e = hipEventCreateWithFlags(&h->ev_up[b], hipEventDisableTiming);
    }
    if (e == hipSuccess && mode == CYMF_MODE_EXACT) e = hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(CYMF_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e)); }
    int rc = h->d_loss.alloc(1);
    if (rc) { delete h; return rc; }
    if (mode == CYMF_MODE_THROUGHPUT) {
        // HOGWILD updates rows from all 8 XCDs inside one kernel, and the per-XCD L2s are not coherent
        // with each other: with cacheable memory each XCD trains a private copy of a row and the last
        // write-back wins (measured: learning stalls).  Uncached device memory keeps every access at the
        // memory side (Infinity Cache / HBM), which is where these random row gathers are served anyway.
        const int f = getenv("CYMF_BPR_MEMTYPE") ? atoi(getenv("CYMF_BPR_MEMTYPE")) : 2;
        h->f32.W.fine = h->f32.H.fine = h->f32.W0.fine = h->f32.W1.fine = h->f32.H0.fine = h->f32.H1.fine = f;
    }
    if (mode == CYMF_MODE_EXACT) {
        if (const char *el = getenv("CYMF_BPR_EXACT_LEVELS")) h->exact_tickets = !(el[0] == '1');
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cu = prop.multiProcessorCount;
        if (h->n_cu <= 0 || K > 256) h->exact_tickets = false;   // K > 256: per-level launches of the two-pass kernel
        if (h->exact_tickets) {   // rows are handed between wavefronts of one kernel, across XCDs
            for (DevBuf<float> *b : {&h->f32.W, &h->f32.H, &h->f32.W0, &h->f32.W1, &h->f32.H0, &h->f32.H1}) b->fine = 2;
            for (DevBuf<double> *b : {&h->f64.W, &h->f64.H, &h->f64.W0, &h->f64.W1, &h->f64.H0, &h->f64.H1}) b->fine = 2;
            h->d_done.fine = 2;
            h->d_next.fine = 2;
        }
    }
    if (const char *e3 = getenv("CYMF_BPR_XCD_STRIDE")) h->xcd_stride = std::max(1, atoi(e3));
    if (const char *e4 = getenv("CYMF_BPR_DIAG")) h->xcd_stride |= atoi(e4) << 8;
    if (const char *e6 = getenv("CYMF_BPR_PF")) h->step_pf = atoi(e6);   // 8 (default), 84, 164, 168 = (PF, PFJ) pairs
    if (const char *e7 = getenv("CYMF_BPR_ITEM_ALIGNED")) h->item_aligned = e7[0] == '1';
    if (const char *e8 = getenv("CYMF_BPR_ADAPTIVE_RPI")) h->adaptive_rpi_factor = std::max(1, atoi(e8));
    if (const char *e5 = getenv("CYMF_BPR_HOT_THRESHOLD")) h->hot_threshold = std::max(1, atoi(e5));
    if (const char *e1 = getenv("CYMF_BPR_MAX_WAVES")) h->max_waves = std::max(1, atoi(e1));
    if (const char *eg = getenv("CYMF_BPR_GROUPS")) h->group_mode = atoi(eg) ? 1 : 0;
    if (const char *ew = getenv("CYMF_BPR_GROUP_WAVES")) h->group_waves = std::max(1, atoi(ew));
    if (const char *e2 = getenv("CYMF_BPR_ROWS_PER_INFLIGHT")) h->rows_per_inflight = std::max(1, atoi(e2));
    *out = h;
    return 0;
}

extern "C" int cymf_bpr_set_steps_per_epoch(cymf_bpr *h, int32_t steps) {
    if (!h || steps < 0) return fail(CYMF_ERR_INVALID, "cymf_bpr_set_steps_per_epoch: bad arguments");
    if (h->have_data) return fail(CYMF_ERR_INVALID, "cymf_bpr_set_steps_per_epoch must precede cymf_bpr_set_data");
    h->steps_auto = steps == 0;          // 0: chosen from the data at cymf_bpr_set_data (choose_steps_per_epoch)
    h->steps_per_epoch = steps == 0 ? 1 : steps;
    return 0;
}

extern "C" int cymf_bpr_get_steps_per_epoch(cymf_bpr *h, int32_t *steps) {
    if (!h || !steps) return fail(CYMF_ERR_INVALID, "cymf_bpr_get_steps_per_epoch: bad arguments");
    *steps = h->steps_per_epoch;
    return 0;
}

extern "C" int cymf_bpr_set_data(cymf_bpr *h, const int32_t *users, const int32_t *positives, int64_t N,
                                 const int32_t *indptr, const int32_t *indices, const int64_t *global_pos,
                                 int64_t N_global) {
    if (!h || N < 0 || (N > 0 && (!users || !positives)) || !indptr)
        return fail(CYMF_ERR_INVALID, "cymf_bpr_set_data: bad arguments");
    CYMF_TRY(use_device(h->device));
    static const bool dbg_t = getenv("CYMF_DEBUG_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) { if (dbg_t) { const double t = now(); fprintf(stderr, "[set_data] %s %.0f ms\n", what, t - t_prev); t_prev = t; } };
    if (!global_pos) N_global = N;
    if (N_global < N || N_global >= (int64_t)0xffffffffll)
        return fail(CYMF_ERR_INVALID, "cymf_bpr_set_data: N_global=%lld must be in [N, 2^32-1)", (long long)N_global);
    const int64_t nnz = indptr[h->U];
    if (nnz < 0 || (nnz > 0 && !indices)) return fail(CYMF_ERR_INVALID, "cymf_bpr_set_data: bad CSR");
    // host-side validation: a bad index would fault the kernels
    for (int32_t u = 0; u < h->U; ++u) {
        if (indptr[u] > indptr[u + 1]) return fail(CYMF_ERR_INVALID, "cymf_bpr_set_data: indptr not monotone at %d", u);
        for (int32_t p = indptr[u]; p < indptr[u + 1]; ++p) {
            if (indices[p] < 0 || indices[p] >= h->I) return fail(CYMF_ERR_INVALID, "cymf_bpr_set_data: item index out of range");
            if (p > indptr[u] && indices[p - 1] > indices[p])
                return fail(CYMF_ERR_INVALID, "cymf_bpr_set_data: CSR indices of user %d are not sorted", u);
        }
    }
    for (int64_t l = 0; l < N; ++l) {
        if (users[l] < 0 || users[l] >= h->U || positives[l] < 0 || positives[l] >= h->I)
            return fail(CYMF_ERR_INVALID, "cymf_bpr_set_data: triplet %lld out of range", (long long)l);
        if (global_pos && (global_pos[l] < 0 || global_pos[l] >= N_global))
            return fail(CYMF_ERR_INVALID, "cymf_bpr_set_data: global_pos[%lld] out of range", (long long)l);
    }
    lap("validation");
    h->prep[0].epoch = h->prep[1].epoch = -1;   // (exact mode: no schedule made for other data survives)
    h->N = N; h->N_global = N_global;
    h->h_users.assign(users, users + N);
    h->h_pos_items.assign(positives, positives + N);
    h->h_gpos.resize((size_t)N);
    h->gpos_identity = global_pos == nullptr;
    for (int64_t l = 0; l < N; ++l) h->h_gpos[l] = (uint32_t)(global_pos ? global_pos[l] : l);
    h->h_indptr.assign(indptr, indptr + h->U + 1);
    h->h_indices.assign(indices, indices + nnz);
    CYMF_TRY(h->d_indptr.upload(h->h_indptr.data(), h->h_indptr.size(), h->stream));
    CYMF_TRY(h->d_indices.upload(h->h_indices.data(), h->h_indices.size(), h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    lap("host copies, CSR upload");
    if (!h->rng_ready) {   // ONE generator for the whole fit (bpr.pyx:141)
        // >= 2M draws per epoch: chunked jump-ahead generator (rng.hip).  Smaller epochs: the lock-free mode generates several
        // epochs per call with that generator (request_epoch_draws); the exact mode, whose epochs take milliseconds of
        // hand-offs anyway, keeps the one-workgroup walker and one epoch per call.
        const bool big = N_global >= (int64_t)2 << 20;
        h->draw_batch = 1;
        if (!big && h->mode == CYMF_MODE_THROUGHPUT && N_global > 0 && !getenv("CYMF_BPR_NO_DRAW_BATCH"))
            h->draw_batch = std::max<int64_t>(2, std::min<int64_t>(64, (((int64_t)4 << 20) + N_global - 1) / N_global));
        if (h->draw_batch > 1 && !h->gen_stream) {
            CYMF_HIP(hipStreamCreateWithPriority(&h->gen_stream, hipStreamNonBlocking, h->prio_side));
            for (int b = 0; b < 2; ++b) CYMF_HIP(hipEventCreateWithFlags(&h->ev_batch_sampled[b], hipEventDisableTiming));
        }
        CYMF_TRY(h->rng.init(h->seed, (uint64_t)h->I, h->draw_batch > 1 ? h->gen_stream : h->rng_stream, /*parallel=*/big || h->draw_batch > 1));
        h->rng_ready = true;
    }
    lap("generator");
    if (h->mode == CYMF_MODE_THROUGHPUT) CYMF_TRY(build_throughput_layout(h));
    lap("layout (total)");
    h->h_pos_bits.clear();
    if (h->mode == CYMF_MODE_EXACT && (uint64_t)h->U * (uint64_t)h->I <= (1ull << 31)) {   // user_positives (bpr.pyx:146-147) as a bitmap
        h->h_pos_bits.assign((size_t)(((uint64_t)h->U * h->I + 63) >> 6), 0ull);
        for (int32_t u = 0; u < h->U; ++u)
            for (int32_t p = indptr[u]; p < indptr[u + 1]; ++p) {
                const size_t bit = (size_t)u * h->I + indices[p];
                h->h_pos_bits[bit >> 6] |= 1ull << (bit & 63);
            }
    }
    h->have_data = true;
    return 0;
}

template <typename T>
static int upload_store(cymf_bpr *h, BprStore<T> &st, const double *W, const double *H) {
    const size_t nW = (size_t)h->U * h->K, nH = (size_t)h->I * h->K;
    CYMF_TRY(upload_f64(st.W, W, nW, h->stream));
    CYMF_TRY(upload_f64(st.H, H, nH, h->stream));
    if (h->opt == CYMF_OPT_ADAGRAD) {          // accumulators start at ONE (optimizer.pyx:69-70)
        CYMF_TRY(fill_dev<T>(st.W0, nW, (T)1, h->stream));
        CYMF_TRY(fill_dev<T>(st.H0, nH, (T)1, h->stream));
    } else if (h->opt == CYMF_OPT_ADAM) {      // zeros (optimizer.pyx:143-146)
        CYMF_TRY(fill_dev<T>(st.W0, nW, (T)0, h->stream));
        CYMF_TRY(fill_dev<T>(st.W1, nW, (T)0, h->stream));
        CYMF_TRY(fill_dev<T>(st.H0, nH, (T)0, h->stream));
        CYMF_TRY(fill_dev<T>(st.H1, nH, (T)0, h->stream));
    }
    CYMF_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cymf_bpr_upload(cymf_bpr *h, const double *W, const double *H) {
    if (!h || !W || !H) return fail(CYMF_ERR_INVALID, "cymf_bpr_upload: bad arguments");
    CYMF_TRY(use_device(h->device));
    if (h->dtype == CYMF_F32) CYMF_TRY(upload_store(h, h->f32, W, H));
    else CYMF_TRY(upload_store(h, h->f64, W, H));
    if (h->comm) {
        const size_t n = (size_t)h->I * h->K;
        if (h->comm_stream) CYMF_HIP(hipStreamSynchronize(h->comm_stream));   // (a re-upload drops an exchange in flight)
        CYMF_TRY(h->d_snap.alloc(n));
        if (h->overlap_exchange) {
            // padded to a multiple of the world, padding zero: what the reduce-scatter + all-gather form of the exchange (CYMF_COMM_RS_AG=1) needs
            const size_t n_pad = (size_t)comm_padded_count(h->comm, (int64_t)n + DELTA_TAIL);
            for (int b = 0; b < 2; ++b) {
                CYMF_TRY(h->d_local[b].alloc(n_pad)); CYMF_TRY(h->d_glob[b].alloc(n_pad));
                CYMF_TRY(h->d_local[b].zero(h->stream)); CYMF_TRY(h->d_glob[b].zero(h->stream));
            }
            CYMF_TRY(h->d_base.alloc(n));
            CYMF_HIP(hipMemcpyAsync(h->d_base.p, h->f32.H.p, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
            h->exch_pending = false;
        } else {
            CYMF_TRY(h->d_delta.alloc(n + DELTA_TAIL));
        }
        CYMF_HIP(hipMemcpyAsync(h->d_snap.p, h->f32.H.p, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        CYMF_HIP(hipStreamSynchronize(h->stream));
    }
    h->have_params = true;
    return 0;
}

extern "C" int cymf_bpr_download(cymf_bpr *h, double *W, double *H) {
    if (!h || !W || !H) return fail(CYMF_ERR_INVALID, "cymf_bpr_download: bad arguments");
    if (!h->have_params) return fail(CYMF_ERR_INVALID, "cymf_bpr_download before cymf_bpr_upload");
    CYMF_TRY(use_device(h->device));
    if (h->comm && h->overlap_exchange) CYMF_TRY(finish_exchange(h, false));   // the last step's exchange is still in flight
    if (h->comm && !h->user_bounds.empty() && h->dtype == CYMF_F32)               // every rank returns all user rows
        CYMF_TRY(comm_allgatherv(h->comm, h->f32.W.p, h->user_bounds.data(), (int64_t)h->K * (int64_t)sizeof(float), h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    const size_t nW = (size_t)h->U * h->K, nH = (size_t)h->I * h->K;
    if (h->dtype == CYMF_F32) { CYMF_TRY(download_f64(h->f32.W, W, nW, h->stream)); CYMF_TRY(download_f64(h->f32.H, H, nH, h->stream)); }
    else { CYMF_TRY(download_f64(h->f64.W, W, nW, h->stream)); CYMF_TRY(download_f64(h->f64.H, H, nH, h->stream)); }
    return 0;
}

extern "C" int cymf_bpr_steps(cymf_bpr *h, int32_t n_steps, double *loss_sum_out) {
    if (!h || n_steps < 0) return fail(CYMF_ERR_INVALID, "cymf_bpr_steps: bad arguments");
    if (!h->have_params || !h->have_data) return fail(CYMF_ERR_INVALID, "cymf_bpr_steps before upload/set_data");
    if (h->mode != CYMF_MODE_THROUGHPUT) return fail(CYMF_ERR_INVALID, "cymf_bpr_steps needs CYMF_MODE_THROUGHPUT");
    CYMF_TRY(use_device(h->device));
    if (loss_sum_out) CYMF_TRY(h->d_loss.zero(h->stream));
    for (int32_t s = 0; s < n_steps;) {
        const int32_t before = h->step_cursor;
        CYMF_TRY(run_one_step(h, n_steps - s));
        s += (h->step_cursor == 0 ? h->steps_per_epoch : h->step_cursor) - before;   // (the cursor wraps at the end of an epoch)
    }
    if (loss_sum_out) CYMF_TRY(fetch_loss(h, loss_sum_out));
    return 0;
}

extern "C" int cymf_bpr_epochs(cymf_bpr *h, int32_t n_epochs, double *loss_out) {
    if (!h || n_epochs < 0) return fail(CYMF_ERR_INVALID, "cymf_bpr_epochs: bad arguments");
    if (!h->have_params || !h->have_data) return fail(CYMF_ERR_INVALID, "cymf_bpr_epochs before upload/set_data");
    CYMF_TRY(use_device(h->device));
    if (h->mode == CYMF_MODE_EXACT) {
        if (h->comm) return fail(CYMF_ERR_UNSUPPORTED, "exact (sequential-order) mode is single-GPU by definition");
        for (int32_t e = 0; e < n_epochs; ++e) {
            double *lo = loss_out ? loss_out + e : nullptr;
            const bool next = e + 1 < n_epochs;   // the following epoch's schedule is made under this epoch's kernel
            if (h->dtype == CYMF_F32) CYMF_TRY(epoch_exact<float>(h, h->f32, lo, next));
            else CYMF_TRY(epoch_exact<double>(h, h->f64, lo, next));
        }
        return 0;
    }
    if (h->step_cursor != 0) return fail(CYMF_ERR_INVALID, "cymf_bpr_epochs called in the middle of an epoch");
    for (int32_t e = 0; e < n_epochs; ++e) {
        double loss = 0;
        CYMF_TRY(cymf_bpr_steps(h, h->steps_per_epoch, loss_out ? &loss : nullptr));
        if (loss_out) loss_out[e] = h->N ? loss / (double)h->N : 0.0;
    }
    return 0;
}

extern "C" int cymf_bpr_sync(cymf_bpr *h) {
    if (!h) return fail(CYMF_ERR_INVALID, "cymf_bpr_sync: NULL handle");
    CYMF_TRY(use_device(h->device));
    if (h->comm && h->overlap_exchange && h->have_params) CYMF_TRY(finish_exchange(h, false));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    // and the side stream: the next epoch's index stream and skip tests are part of the work this handle has issued
    if (h->rng_stream) CYMF_HIP(hipStreamSynchronize(h->rng_stream));
    if (h->gen_stream) CYMF_HIP(hipStreamSynchronize(h->gen_stream));
    return 0;
}

extern "C" int cymf_bpr_stats(cymf_bpr *h, int64_t *performed, int64_t *skipped) {
    if (!h) return fail(CYMF_ERR_INVALID, "cymf_bpr_stats: NULL handle");
    CYMF_TRY(use_device(h->device));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    if (h->mode == CYMF_MODE_THROUGHPUT && h->have_data) CYMF_TRY(collect_skips(h));
    if (performed) *performed = h->performed;
    if (skipped) *skipped = h->skipped;
    return 0;
}

extern "C" int cymf_bpr_set_profiling(cymf_bpr *h, int on) {
    if (!h) return fail(CYMF_ERR_INVALID, "cymf_bpr_set_profiling: NULL handle");
    h->profiling = on != 0;
    return 0;
}

extern "C" int cymf_bpr_kernel_time(cymf_bpr *h, double *ms_total, int64_t *launches, int64_t *units) {
    if (!h) return fail(CYMF_ERR_INVALID, "cymf_bpr_kernel_time: NULL handle");
    CYMF_TRY(use_device(h->device));
    CYMF_TRY(drain_profile(h));
    if (ms_total) *ms_total = h->prof_ms;
    if (launches) *launches = h->prof_launches;
    if (units) *units = h->prof_units;
    h->prof_ms = 0; h->prof_launches = 0; h->prof_units = 0;
    return 0;
}

extern "C" int cymf_bpr_last_negatives(cymf_bpr *h, int32_t *out, int64_t n) {
    if (!h || !out || n != h->N) return fail(CYMF_ERR_INVALID, "cymf_bpr_last_negatives: n must equal N");
    CYMF_TRY(use_device(h->device));
    if (h->mode == CYMF_MODE_EXACT) {
        if ((int64_t)h->h_last_neg.size() != n) return fail(CYMF_ERR_INVALID, "no epoch has run yet");
        memcpy(out, h->h_last_neg.data(), (size_t)n * sizeof(int32_t));
        return 0;
    }
    if (n == 0) return 0;
    CYMF_TRY(h->d_unsorted_neg.alloc((size_t)n));
    hipLaunchKernelGGL(bpr_unsort_neg_kernel, dim3(ew_blocks(n)), dim3(256), 0, h->stream,
                       h->d_slot_neg[(int)((h->epoch_cursor - (h->step_cursor == 0 ? 1 : 0)) & 1)].p, h->d_slot_local.p,
                       h->d_unsorted_neg.p, n);
    CYMF_HIP(hipGetLastError());
    CYMF_HIP(hipMemcpyAsync(out, h->d_unsorted_neg.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    CYMF_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int cymf_bpr_attach_comm(cymf_bpr *h, cymf_comm *c) {
    if (!h || !c) return fail(CYMF_ERR_INVALID, "cymf_bpr_attach_comm: bad arguments");
    if (h->mode != CYMF_MODE_THROUGHPUT) return fail(CYMF_ERR_UNSUPPORTED, "a communicator needs throughput mode");
    if (h->have_params || h->have_data) return fail(CYMF_ERR_INVALID, "cymf_bpr_attach_comm must precede cymf_bpr_set_data and cymf_bpr_upload");
    h->comm = c;
    if (const char *e = getenv("CYMF_BPR_SYNC_EXCHANGE")) h->overlap_exchange = !(e[0] == '1');
    if (h->overlap_exchange && !h->comm_stream) {
        hipError_t e = hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, h->prio_comm);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_delta_ready, hipEventDisableTiming);
        for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipEventCreateWithFlags(&h->ev_reduced[b], hipEventDisableTiming);
        if (e != hipSuccess) return fail(CYMF_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    return 0;
}

extern "C" int cymf_bpr_set_user_bounds(cymf_bpr *h, const int64_t *bounds) {
    if (!h || !bounds || !h->comm) return fail(CYMF_ERR_INVALID, "cymf_bpr_set_user_bounds: needs a handle with a communicator and the bounds");
    const int world = comm_world(h->comm);
    if (bounds[0] != 0 || bounds[world] != h->U) return fail(CYMF_ERR_INVALID, "cymf_bpr_set_user_bounds: bounds must run from 0 to U");
    for (int r = 0; r < world; ++r)
        if (bounds[r] > bounds[r + 1]) return fail(CYMF_ERR_INVALID, "cymf_bpr_set_user_bounds: bounds not monotone");
    h->user_bounds.assign(bounds, bounds + world + 1);
    return 0;
}

extern "C" int cymf_bpr_destroy(cymf_bpr *h) {
    if (!h) return 0;
    if (!cymf::runtime_alive(h->device)) return 0;   // process exit / runtime already torn down: leak quietly
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->rng_stream) (void)hipStreamSynchronize(h->rng_stream);
    if (h->gen_stream) { (void)hipStreamSynchronize(h->gen_stream); (void)hipStreamDestroy(h->gen_stream); }
    for (int b = 0; b < 2; ++b) if (h->ev_batch_sampled[b]) (void)hipEventDestroy(h->ev_batch_sampled[b]);
    for (auto &pe : h->prof_events) { (void)hipEventDestroy(pe.first); (void)hipEventDestroy(pe.second); }
    for (auto &e : h->prof_pool) (void)hipEventDestroy(e);
    for (int b = 0; b < 2; ++b) {
        if (h->ev_gen[b]) (void)hipEventDestroy(h->ev_gen[b]);
        if (h->ev_sampled[b]) (void)hipEventDestroy(h->ev_sampled[b]);
        if (h->ev_epoch_done[b]) (void)hipEventDestroy(h->ev_epoch_done[b]);
    }
    if (h->comm_stream) { (void)hipStreamSynchronize(h->comm_stream); (void)hipStreamDestroy(h->comm_stream); }
    if (h->ev_delta_ready) (void)hipEventDestroy(h->ev_delta_ready);
    for (int b = 0; b < 2; ++b) if (h->ev_reduced[b]) (void)hipEventDestroy(h->ev_reduced[b]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->rng_stream) (void)hipStreamDestroy(h->rng_stream);
    for (int b = 0; b < 2; ++b) { if (h->h_draws2[b]) (void)hipHostFree(h->h_draws2[b]); if (h->ev_draws_host[b]) (void)hipEventDestroy(h->ev_draws_host[b]); }
    if (h->up_stream) { (void)hipStreamSynchronize(h->up_stream); (void)hipStreamDestroy(h->up_stream); }
    for (int b = 0; b < 2; ++b) if (h->ev_up[b]) (void)hipEventDestroy(h->ev_up[b]);
    delete h;
    return 0;
}
