// bpr_groups.h -- BPR lock-free mode on SMALL tables (bpr_groups.hip): four triplets per wavefront, every write-back an
// atomic delta.  Replaces the loop of cymf/bpr.pyx:160-171 in its num_threads > 1 regime where the item-bucketed step
// kernel of bpr.hip is sized by the table, not by the chip.
#pragma once
#include "common.h"
#include "rows.h"

namespace cymf {

struct BprGroupDev {
    float *W, *H;          // (U,K), (I,K)
    float *W0, *W1;        // optimizer state for W (AdaGrad: acc; Adam: m, v)
    float *H0, *H1;        // optimizer state for H
    int K;
    float wd;
    OptParams<float> opt;
};

// Membership structure of the lock-free mode (the reference asks `negative in user_positives[user]` of a std::set per user,
// cymf/bpr.pyx:140,166): ONE open-addressing table over all (user, item) pairs of X, 64-bit keys, load <= 1/2, linear probing.
constexpr unsigned long long PAIR_EMPTY = ~0ull;
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long pair_hash(unsigned long long k) {   // murmur3 finalizer
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
__device__ __forceinline__ bool pair_table_has(const unsigned long long *__restrict__ table, unsigned long long mask, int32_t u, int32_t item) {
    const unsigned long long key = ((unsigned long long)(unsigned int)u << 32) | (unsigned int)item;
    unsigned long long slot = pair_hash(key) & mask;
    while (true) {
        const unsigned long long v = table[slot];
        if (v == key) return true;
        if (v == PAIR_EMPTY) return false;
        slot = (slot + 1) & mask;
    }
}
#endif

// Negatives resolved INSIDE the group kernel (draws != nullptr): slot s of the sorted order is triplet slot_pos[s] of the global
// shuffled order, its negative is draw slot_pos[s] of the epoch (one draw per triplet, skipped ones included, cymf/bpr.pyx:165),
// skipped (-1) when the pair table holds (user, draw).  The value is also written to slot_neg_out[s] (cymf_bpr_last_negatives).
// On a small problem the separate sampling kernel costs half a step kernel's time and runs beside it (C2: 137 us against 280).
struct BprGroupSample {
    const uint32_t *slot_pos = nullptr;
    const uint32_t *draws = nullptr;
    const unsigned long long *table = nullptr;
    unsigned long long mask = 0;
    int32_t *slot_neg_out = nullptr;
};

// can the group kernel serve this shape?  (K <= 128: a row is at most eight values per lane of a 16-lane group)
inline bool bpr_group_supported(int K) { return K >= 1 && K <= 128; }

// One launch over the slots [slot_begin, slot_end) of the (step, item)-sorted order: slot_user / slot_item / slot_neg as the
// step kernel reads them (slot_neg: -1 = skipped draw, bit 30 ignored).  n_waves wavefronts (four groups each) walk the
// 16-slot blocks interleaved -- group g takes blocks g, g + G, g + 2G, ... -- so the launch advances through the order as
// one front of about n_waves * 64 slots.  loss_acc += sum of the performed triplets' losses, performed_acc += their number.
int bpr_group_launch(int opt, const BprGroupDev &d, const int32_t *slot_user, const int32_t *slot_item, const int32_t *slot_neg,
                     int64_t slot_begin, int64_t slot_end, int n_waves, double *loss_acc, unsigned long long *performed_acc,
                     hipStream_t s, const BprGroupSample &sample = BprGroupSample());

}  // namespace cymf
