// relmf_tiles.h -- RelMF lock-free mode as a stratified tile schedule (relmf_tiles.hip).
#pragma once
#include "common.h"
#include "rows.h"

namespace cymf {

// The U x I cells are cut into B x B tiles (user block b, item block b').  Sub-step s of an epoch runs the B tiles
// (b, (b + s) mod B): no two of them share a user row or an item row, so the workgroups need no atomics on HBM and no
// exchange -- every draw of the epoch is applied exactly once over the B sub-steps.
// One documented exception, Adam only: inside a tile the workers share the item rows through LDS compare-and-swaps, and an
// Adam step whose swap fails (another worker updated the same two elements first) is NOT retried -- its step size does not
// shrink with the gradient, so re-applying it from the moments the other worker has just written moves the row twice
// (measured: norm of H +21 % with retries, +8 % without, against the sequential order).  SGD and AdaGrad retry on the
// value found, so for them no draw's update is ever lost.
struct RelTilePlan {
    int32_t U = 0, I = 0, K = 0, opt = 0;
    int32_t B = 0;        // blocks per side = workgroups per launch = launches per epoch
    int32_t ub = 0;       // users per block  (the last block may be shorter)
    int32_t ib = 0;       // items per block  (<= 64: one lane per item of the block)
    int32_t R = 0;        // row elements per lane, K <= 64 R
    int32_t threads = 0;  // workgroup size of the tile kernel
    int32_t lpd = 16;     // lanes per draw: 16 (four elements per lane and 64-element block) or 8 (eight)
    int64_t N = 0;        // draws per epoch = U * I
    size_t lds_bytes = 0;
};

struct RelTileParams {
    float *W, *H, *W0, *W1, *H0, *H1;
    const float *X, *prop;
    float wd, clip;
    OptParams<float> opt;
};

struct RelTileBufs {
    DevBuf<uint32_t> pass1;               // cells grouped by user block
    DevBuf<uint32_t> sorted[2];           // cells grouped by tile, double-buffered by epoch parity
    DevBuf<uint32_t> cnt1, off1, cur1;    // [B], [B+1], [B]
    DevBuf<uint32_t> cnt2, cur2;          // [B*B]
    DevBuf<uint32_t> toff[2];             // [B*B+1] tile offsets, by epoch parity
};

// false: this (U, I, K, optimizer) is served by the older kernels (K > 256, blocks that would not fit the LDS, ...)
bool relmf_tile_plan(int32_t U, int32_t I, int32_t K, int opt, RelTilePlan *plan);
// groups the epoch's cells (device, N of them) by tile into bufs.sorted[parity] / bufs.toff[parity]; stream order on s
int relmf_tile_bucket(const RelTilePlan &p, const uint32_t *cells, RelTileBufs &bufs, int parity, hipStream_t s);
// the B sub-steps of one epoch
int relmf_tile_epoch(const RelTilePlan &p, const RelTileParams &d, const RelTileBufs &bufs, int parity, int64_t epoch,
                     double *loss_acc, int *err, hipStream_t s);

}  // namespace cymf
