// bpr_groups.hip -- BPR lock-free mode on SMALL tables (ml-1m-shaped: 6 040 x 3 706, K = 64).
// Replaces the loop of cymf/bpr.pyx:160-171 in its num_threads > 1 regime (HOGWILD, cymf/bpr.pyx:75); per-triplet arithmetic
// as BprModel.forward / backward (cymf/model.pyx:47-87) and the optimizers (cymf/optimizer.pyx:40-160).
//
// Why a second lock-free kernel.  The step kernel of bpr.hip writes W[u] and H[j] back in place; that is only right while two
// wavefronts rarely hold the same row at once, and on a table of a few thousand rows that bound admits fewer than a hundred
// wavefronts, each a latency chain: 1.27 ms per 466 k-triplet epoch on C2, 0.07 of the HBM peak, and the rows it does lose
// cost a quarter of the item factors' norm against the sequential order.  A small problem is not short of
// bandwidth, it is short of independent work per row: every item row is touched ~250 times per epoch.  So here
//   * a triplet is worked by a GROUP of 16 lanes (one DPP row): a wavefront runs FOUR triplets per step, a row of K <= 128
//     factors is E = K / 16 values per lane, the dot product is four DPP steps inside the row and nothing is wave-wide;
//     one wavefront per CU is enough to outrun the write-back below, so few triplets are in flight at any moment;
//   * every write-back is a float-atomic DELTA -- W[u] += new - old, H[j] += new - old, AdaGrad accumulators += g^2 (sums, so
//     adding is exact) -- no concurrent update of a row is ever lost and every slot is applied exactly once; what is left
//     of HOGWILD is staleness (a gradient formed from a value a few microseconds old), and that is bounded by the front;
//   * the groups walk the (step, item)-sorted slots INTERLEAVED in blocks of 16 (group g: blocks g, g + G, ...), so the launch
//     moves through the order as one front of ~n_waves * 64 slots: with steps_per_epoch windows of the shuffled order it
//     follows the reference's own order to within a window plus that front;
//   * inside a block consecutive slots of the same positive item forward H[i] in registers (per-lane selects, no branch, no
//     wait) and add ONE delta when the run ends: item-id bucketing still saves most of the H[i] traffic;
//   * lane l of a group holds elements l, l + 16, ...: one wave instruction touches 4 rows x 64 contiguous bytes, which is
//     the request count of a contiguous 256-B atomic (the memory side works in 64-B requests), not 4x as a 16-B-per-lane
//     layout would.
// Cost: ~2.2 rows of atomics per triplet at the chip's ~1.3 TB/s float-atomic rate (MI355X_MICROARCH.md): affordable exactly
// when the tables are small (C2: 0.25 GB per epoch = 0.19 ms; C3 would pay 100 GB).
// Adam (cymf/optimizer.pyx:126-160): the parameters take atomic deltas; the SECOND moment v = b2 v + (1 - b2) g^2 is a slow
// average (1 - b2 = 0.001), so its increments commute to first order and are added atomically as well -- v must not lose
// updates: the step is lr m^ / sqrt(v^), and a row whose v sees only every c-th of its updates keeps the large steps of its
// first touches c times longer (measured on C2 with v stored plainly: norm of H +50 % against the sequential order).  The FIRST
// moment m = 0.9 m + 0.1 g forgets within ten updates: increments of concurrent groups do not commute (summing c runs of 16
// updates multiplies the row's m by -(c - 1)), so m is stored as the last writer's value, an unbiased estimate of the
// current mean gradient.  A memory of ten updates also means that Adam tolerates far fewer triplets in flight than SGD or
// AdaGrad do (C2, norm of H against the sequential order: 32 / 128 / 256 wavefronts -0.9 % / +6.9 % / +13.6 %; adding m's
// increments atomically instead did not help; neither did work items cut at item boundaries, which hand every run of one
// positive item inside a window to ONE group that walks it in sequence -- C2 at 73 wavefronts: norm of H +7.1 % instead of
// +1.8 %, 4.2 ms per epoch instead of 3.3: the concurrency that matters is the NEGATIVES', which no bucketing orders): the
// host sizes Adam's launch at about one percent of an epoch in flight.
// Also tried (round 3, C2, 30 epochs against the sequential oracle; all removed again):
//   * the c^ updates expected in flight on a row written as ONE sequential block shared out evenly -- c^ = 1 + in-flight triplets
//     x the row's share of the epoch's touches, delta = (1 - b1^c^) / c^, step_k = lr (9 delta m + (1 - 9 delta) g_k) / ..., m +=
//     delta (g_k - m) as an atomic delta, which for c^ = 1 is the reference's update and for c^ concurrent updates leaves the
//     parameter and m where c^ sequential ones would: norm of H +2.1 / +13.9 / +14.3 / +23 % at 73 / 146 / 292 / 512 wavefronts
//     (last writer's m: +1.0 / +6.8 / +17.6 / +15.0 %) -- the stale first moment is not what inflates the norm;
//   * the step of the negative item's row scaled by c^-1/4 (between the sqrt(c) of c normalised steps that are noise and the c of
//     steps that are signal): within the bars at 512 wavefronts (loss +0.03 %, norms +4 / +3 %, Recall@5 -0.002) and at 292 (norms
//     -7 / +6 %), 0.5 instead of 2.5 ms per epoch -- but an exponent without a derivation, fitted on one data set: not shipped.
// What inflates the norm is the NORMALISED step: c triplets that read the same stale row step the same way by lr each, whatever
// the size of their gradients, where a sequence would see its own overshoot; SGD's and AdaGrad's steps shrink with the gradient.
#include "bpr_groups.h"

#include <algorithm>

namespace cymf {
namespace {

constexpr int GL = 16;    // lanes per group = one DPP row
constexpr int GPW = 4;    // groups (triplets in flight per step) per wavefront

__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a DPP row, in every lane of the row
    v += dpp_f32<DPP_QUAD_PERM_1032>(v);
    v += dpp_f32<DPP_QUAD_PERM_2301>(v);
    v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f32<DPP_ROW_MIRROR>(v);
    return v;
}

// value of lane `t` of this lane's group
__device__ __forceinline__ int group_bcast(int v, int t, int lane) {
    return __builtin_amdgcn_ds_bpermute(((lane & ~(GL - 1)) + t) << 2, v);
}

template <int E>
struct GRow {
    float v[E];
};

template <int E, bool FULL>
__device__ __forceinline__ void grow_load(GRow<E> &r, const float *__restrict__ base, int K, int gl, float fill = 0.0f) {
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const int k = gl + GL * q;
        if constexpr (FULL) {
            r.v[q] = base[k];
        } else {   // unconditional load from a clamped index, selected afterwards (no branch per element)
            const float x = base[k < K ? k : 0];
            r.v[q] = k < K ? x : fill;
        }
    }
}

template <int E, bool FULL>
__device__ __forceinline__ void grow_store(const GRow<E> &r, float *__restrict__ base, int K, int gl) {
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const int k = gl + GL * q;
        if (FULL || k < K) base[k] = r.v[q];
    }
}

// base[k] += a[k] - b[k], no-return global_atomic_add_f32: 4 rows x 64 contiguous bytes per wave instruction
template <int E, bool FULL>
__device__ __forceinline__ void grow_atomic_delta(float *__restrict__ base, const GRow<E> &a, const GRow<E> &b, int K, int gl) {
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const int k = gl + GL * q;
        if (FULL || k < K) atomicAdd(base + k, a.v[q] - b.v[q]);
    }
}

template <int E>
__device__ __forceinline__ void grow_select(GRow<E> &dst, bool take_a, const GRow<E> &a, const GRow<E> &b) {
#pragma unroll
    for (int q = 0; q < E; ++q) dst.v[q] = take_a ? a.v[q] : b.v[q];
}

// E values per lane (K <= 16 E), FULL: K == 16 E (no masked lanes), PF: rows gathered PF slots ahead (ring of 2 PF entries)
template <int E, bool FULL, int OPT, int PF>
__global__ __launch_bounds__(64) void bpr_group_kernel(BprGroupDev d, const int32_t *__restrict__ slot_user,
                                                      const int32_t *__restrict__ slot_item, const int32_t *__restrict__ slot_neg,
                                                      int64_t slot_begin, int64_t slot_end, int64_t n_groups,
                                                      double *__restrict__ loss_acc, unsigned long long *__restrict__ performed_acc,
                                                      BprGroupSample smp) {
    constexpr int NS = opt_num_states(OPT);
    constexpr int NSA = NS ? NS : 1;
    constexpr int RING = 2 * PF;
    constexpr float SFILL = OPT == CYMF_OPT_ADAGRAD ? 1.0f : 0.0f;   // masked lanes of accumulator rows (rows.h: Row::load)
    constexpr int SA = OPT == CYMF_OPT_ADAM ? 1 : 0;                  // index of the additive state row (AdaGrad: acc, Adam: v)
    static_assert(GL % RING == 0, "the ring must divide the block");
    const int lane = (int)(threadIdx.x & 63), gl = lane & (GL - 1);
    const int K = FULL ? GL * E : d.K;
    const int64_t gid = (int64_t)blockIdx.x * GPW + (lane >> 4);
    const int64_t n_blocks = (slot_end - slot_begin + GL - 1) / GL;
    const int64_t n_iter = (n_blocks + n_groups - 1) / n_groups;       // wave-uniform: groups without a block idle through it
    float *const Ws[2] = {d.W0, d.W1};
    float *const Hs[2] = {d.H0, d.H1};

    // metadata of the group's block `it`, one slot per lane of the group; j < 0 = nothing to do (skipped draw, or past the end)
    auto load_meta = [&](int64_t it, int32_t &u, int32_t &i, int32_t &j) {
        const int64_t blk = gid + it * n_groups;
        const int64_t s = slot_begin + blk * GL + gl;
        const bool in = it < n_iter && blk < n_blocks && s < slot_end;
        u = in ? slot_user[s] : 0;
        i = in ? slot_item[s] : 0;
        if (smp.draws) {   // wave-uniform: the negatives are resolved here (three dependent reads, two blocks ahead of their use)
            j = -1;
            if (in) {
                const int32_t draw = (int32_t)smp.draws[smp.slot_pos[s]];
                j = pair_table_has(smp.table, smp.mask, u, draw) ? -1 : draw;
                smp.slot_neg_out[s] = j;
            }
        } else {
            j = in ? slot_neg[s] : -1;
        }
        if (j < 0) { u = 0; i = 0; }
    };
    int32_t u_c, i_c, j_c, u_n, i_n, j_n;
    load_meta(0, u_c, i_c, j_c);
    load_meta(1, u_n, i_n, j_n);

    // ring entry e holds the rows (and the metadata) of the slot t with t % RING == e
    GRow<E> wq[RING], jq[RING], iq[RING], swq[RING][NSA], sjq[RING][NSA], siq[RING][NSA];
    int32_t ur[RING], ir[RING], jr[RING];
    // gather the rows of the slot at position tt of the current block (tt >= 16: of the next block) into entry e
    auto issue = [&](const int e, const int tt) {
        const bool nxt = tt >= GL;   // wave-uniform
        const int pos = tt & (GL - 1);
        const int32_t u = group_bcast(nxt ? u_n : u_c, pos, lane);
        const int32_t i = group_bcast(nxt ? i_n : i_c, pos, lane);
        const int32_t jraw = group_bcast(nxt ? j_n : j_c, pos, lane);
        ur[e] = u; ir[e] = i; jr[e] = jraw;
        const int64_t ou = (int64_t)u * K, oj = (int64_t)(jraw < 0 ? 0 : (jraw & 0x3fffffff)) * K;
        grow_load<E, FULL>(wq[e], d.W + ou, K, gl);
        grow_load<E, FULL>(jq[e], d.H + oj, K, gl);
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            grow_load<E, FULL>(swq[e][q], Ws[q] + ou, K, gl, SFILL);
            grow_load<E, FULL>(sjq[e][q], Hs[q] + oj, K, gl, SFILL);
        }
        // the positive item's row only where a run starts: inside a run it is forwarded in registers
        const int ep = (e + RING - 1) % RING;
        const bool run_on = pos != 0 && jraw >= 0 && jr[ep] >= 0 && ir[ep] == i;
        if (jraw >= 0 && !run_on) {
            const int64_t oi = (int64_t)i * K;
            grow_load<E, FULL>(iq[e], d.H + oi, K, gl);
#pragma unroll
            for (int q = 0; q < NS; ++q) grow_load<E, FULL>(siq[e][q], Hs[q] + oi, K, gl, SFILL);
        }
    };
#pragma unroll
    for (int e = 0; e < RING; ++e) {   // (entries whose item row is never gathered -- inside a run -- are read by selects only)
        jr[e] = -1; ir[e] = 0; ur[e] = 0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            iq[e].v[q] = 0.0f;
#pragma unroll
            for (int n = 0; n < NSA; ++n) siq[e][n].v[q] = 0.0f;
        }
    }
#pragma unroll
    for (int e = 0; e < PF; ++e) issue(e, e);

    GRow<E> hcur, hbase, scur[NSA], sbase;
#pragma unroll
    for (int q = 0; q < E; ++q) { hcur.v[q] = 0.0f; hbase.v[q] = 0.0f; sbase.v[q] = 0.0f; }
#pragma unroll
    for (int n = 0; n < NSA; ++n)
#pragma unroll
        for (int q = 0; q < E; ++q) scur[n].v[q] = 0.0f;
    bool pv = false;
    int32_t pitem = -1;
    float loss_lane = 0.0f, l2_lane = 0.0f;
    unsigned int n_done = 0;

    for (int64_t it = 0; it < n_iter; ++it) {
#pragma unroll 1
        for (int t0 = 0; t0 < GL; t0 += RING) {
#pragma unroll
            for (int p = 0; p < RING; ++p) {
                const int t = t0 + p;
                const int32_t u = ur[p], item = ir[p], jraw = jr[p];
                const bool valid = jraw >= 0;                      // uniform inside the group, not across the wavefront
                const int32_t j = jraw & 0x3fffffff;
                const bool same = t != 0 && valid && pv && item == pitem;
                // H[i]: forwarded from the previous slot of the run, or the row gathered for this slot
                GRow<E> hi, shi[NSA];
                grow_select<E>(hi, same, hcur, iq[p]);
                grow_select<E>(hbase, same, hbase, iq[p]);
#pragma unroll
                for (int n = 0; n < NS; ++n) grow_select<E>(shi[n], same, scur[n], siq[p][n]);
                if constexpr (NS >= 1) grow_select<E>(sbase, same, sbase, siq[p][SA]);
                const GRow<E> w_old = wq[p], hj_old = jq[p];
                GRow<E> sw_old, sj_old;
                if constexpr (NS >= 1) { sw_old = swq[p][SA]; sj_old = sjq[p][SA]; }

                // forward (cymf/model.pyx:47-62): x = w . (hi - hj), loss = -log(sigmoid(x)) (+ wd * l2, reduced once at the end)
                float px = 0.0f, pl = 0.0f;
#pragma unroll
                for (int q = 0; q < E; ++q) {
                    px += wq[p].v[q] * (hi.v[q] - jq[p].v[q]);
                    pl += wq[p].v[q] * wq[p].v[q] + hi.v[q] * hi.v[q] + jq[p].v[q] * jq[p].v[q];
                }
                const float x = row16_sum(px);
                const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));          // e^{-|x|}
                const float loss = fmaxf(-x, 0.0f) + 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + ex);
                const float r1 = __builtin_amdgcn_rcpf(1.0f + ex);
                const float sg = x >= 0.0f ? ex * r1 : r1;                                           // 1 / (1 + e^x)
                // backward (cymf/model.pyx:66-87): all three gradients of a component from its pre-update values
#pragma unroll
                for (int q = 0; q < E; ++q) {
                    const float wv = wq[p].v[q], iv = hi.v[q], jv = jq[p].v[q];
                    const float gw = -(sg * (iv - jv) - d.wd * wv);
                    const float gi = -(sg * wv - d.wd * iv);
                    const float gj = -(sg * (-wv) - d.wd * jv);
                    float dummy = 0.0f;
                    opt_update<float, OPT, true>(d.opt, wq[p].v[q], OPT >= 1 ? swq[p][0].v[q] : dummy, OPT == 2 ? swq[p][1].v[q] : dummy, gw);
                    opt_update<float, OPT, true>(d.opt, hi.v[q], OPT >= 1 ? shi[0].v[q] : dummy, OPT == 2 ? shi[1].v[q] : dummy, gi);
                    opt_update<float, OPT, true>(d.opt, jq[p].v[q], OPT >= 1 ? sjq[p][0].v[q] : dummy, OPT == 2 ? sjq[p][1].v[q] : dummy, gj);
                }
                // does the run of this positive item end here?  (the next slot's metadata has been in the ring for PF - 1 steps)
                const int pn = (p + 1) % RING;
                const bool ends = t == GL - 1 || jr[pn] < 0 || ir[pn] != item;
                if (valid) {
                    const int64_t ou = (int64_t)u * K, oj = (int64_t)j * K;
                    grow_atomic_delta<E, FULL>(d.W + ou, wq[p], w_old, K, gl);
                    grow_atomic_delta<E, FULL>(d.H + oj, jq[p], hj_old, K, gl);
                    if constexpr (NS >= 1) {   // the additive state (AdaGrad: sum of g^2; Adam: v) as a delta
                        grow_atomic_delta<E, FULL>(Ws[SA] + ou, swq[p][SA], sw_old, K, gl);
                        grow_atomic_delta<E, FULL>(Hs[SA] + oj, sjq[p][SA], sj_old, K, gl);
                    }
                    if constexpr (OPT == CYMF_OPT_ADAM) {   // m: the last writer's value
                        grow_store<E, FULL>(swq[p][0], Ws[0] + ou, K, gl);
                        grow_store<E, FULL>(sjq[p][0], Hs[0] + oj, K, gl);
                    }
                    if (ends) {
                        const int64_t oi = (int64_t)item * K;
                        grow_atomic_delta<E, FULL>(d.H + oi, hi, hbase, K, gl);
                        if constexpr (NS >= 1) grow_atomic_delta<E, FULL>(Hs[SA] + oi, shi[SA], sbase, K, gl);
                        if constexpr (OPT == CYMF_OPT_ADAM) grow_store<E, FULL>(shi[0], Hs[0] + oi, K, gl);
                    }
                    loss_lane += gl == 0 ? loss : 0.0f;
                    l2_lane += pl;
                }
                n_done += (unsigned int)__popcll(__ballot(valid && gl == 0));
                hcur = hi;
#pragma unroll
                for (int n = 0; n < NS; ++n) scur[n] = shi[n];
                pv = valid;
                pitem = item;
                // refill the entry PF ahead (this block or the next)
                issue((p + PF) % RING, t + PF);
            }
        }
        u_c = u_n; i_c = i_n; j_c = j_n;
        load_meta(it + 2, u_n, i_n, j_n);
    }
    const float tot = wave_sum(loss_lane + d.wd * l2_lane);
    if (lane == 0 && n_done) {
        atomicAdd(loss_acc, (double)tot);
        atomicAdd(performed_acc, (unsigned long long)n_done);
    }
}

template <int E, bool FULL, int OPT>
void launch_inst(const BprGroupDev &d, const int32_t *su, const int32_t *si, const int32_t *sn, int64_t b, int64_t e, int n_waves,
                 double *loss, unsigned long long *perf, hipStream_t s, const BprGroupSample &smp) {
    // rows held per ring entry: 3 (1 + states) E values per lane; the ring is sized to ~256 registers
    constexpr int per_entry = 3 * (1 + opt_num_states(OPT)) * E;
    constexpr int PF = per_entry <= 16 ? 8 : (per_entry <= 32 ? 4 : 2);
    hipLaunchKernelGGL((bpr_group_kernel<E, FULL, OPT, PF>), dim3(n_waves), dim3(64), 0, s, d, su, si, sn, b, e, (int64_t)n_waves * GPW, loss, perf, smp);
}

template <int E, bool FULL>
void launch_opt(int opt, const BprGroupDev &d, const int32_t *su, const int32_t *si, const int32_t *sn, int64_t b, int64_t e,
                int n_waves, double *loss, unsigned long long *perf, hipStream_t s, const BprGroupSample &smp) {
    switch (opt) {
    case CYMF_OPT_SGD: launch_inst<E, FULL, CYMF_OPT_SGD>(d, su, si, sn, b, e, n_waves, loss, perf, s, smp); break;
    case CYMF_OPT_ADAGRAD: launch_inst<E, FULL, CYMF_OPT_ADAGRAD>(d, su, si, sn, b, e, n_waves, loss, perf, s, smp); break;
    default: launch_inst<E, FULL, CYMF_OPT_ADAM>(d, su, si, sn, b, e, n_waves, loss, perf, s, smp); break;
    }
}

}  // namespace

int bpr_group_launch(int opt, const BprGroupDev &d, const int32_t *slot_user, const int32_t *slot_item, const int32_t *slot_neg,
                     int64_t slot_begin, int64_t slot_end, int n_waves, double *loss_acc, unsigned long long *performed_acc,
                     hipStream_t s, const BprGroupSample &sample) {
    if (!bpr_group_supported(d.K)) return fail(CYMF_ERR_UNSUPPORTED, "bpr_group_launch: K=%d", d.K);
    if (slot_end <= slot_begin) return 0;
    const int64_t n_blocks = (slot_end - slot_begin + GL - 1) / GL;
    n_waves = (int)std::max<int64_t>(1, std::min<int64_t>(n_waves, (n_blocks + GPW - 1) / GPW));
    const int K = d.K;
#define GO_(E_, F_) launch_opt<E_, F_>(opt, d, slot_user, slot_item, slot_neg, slot_begin, slot_end, n_waves, loss_acc, performed_acc, s, sample)
    if (K == 64) GO_(4, true);
    else if (K == 128) GO_(8, true);
    else if (K <= 16) GO_(1, false);
    else if (K <= 32) GO_(2, false);
    else if (K <= 64) GO_(4, false);
    else GO_(8, false);
#undef GO_
    CYMF_HIP(hipGetLastError());
    return 0;
}

}  // namespace cymf
