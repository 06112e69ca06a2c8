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

// can the group kernel serve this shape?  (K <= 128: a row is at most eight values per lane of a 16-lane group)
inline bool bpr_group_supported(int K) { return K >= 1 && K <= 128; }

// One launch over the slots [slot_begin, slot_end) of the (step, item)-sorted order: slot_user / slot_item / slot_neg as the
// step kernel reads them (slot_neg: -1 = skipped draw, bit 30 ignored).  n_waves wavefronts (four groups each) walk the
// 16-slot blocks interleaved -- group g takes blocks g, g + G, g + 2G, ... -- so the launch advances through the order as
// one front of about n_waves * 64 slots.  loss_acc += sum of the performed triplets' losses, performed_acc += their number.
int bpr_group_launch(int opt, const BprGroupDev &d, const int32_t *slot_user, const int32_t *slot_item, const int32_t *slot_neg,
                     int64_t slot_begin, int64_t slot_end, int n_waves, double *loss_acc, unsigned long long *performed_acc,
                     hipStream_t s);

}  // namespace cymf
