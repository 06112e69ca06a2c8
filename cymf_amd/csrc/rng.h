// rng.h -- device-resident negative-sample index stream (bit-exact std::mt19937 + libstdc++-11
// uniform_int_distribution<long>, i.e. the reference's UniformGenerator, cymf/math.pyx:12-18).
#pragma once
#include "common.h"

namespace cymf {

struct RngState {
    uint32_t mt[624];       // current block of untempered words
    uint32_t idx;           // next word of the block to consume (624 = block exhausted)
    uint32_t pad;
    uint64_t raw_consumed;  // raw 32-bit words consumed so far
    uint64_t draws;         // draws produced so far (accepted words)
};

// One stream = one generator created once and never reseeded (cymf/bpr.pyx:141).
class DeviceRng {
  public:
    // range in [1, 2^32-1]
    int init(uint32_t seed, uint64_t range, hipStream_t s);
    // Discards n_skip draws, then writes the next n draws to d_out[0..n) (device pointer).
    int generate(int64_t n_skip, int64_t n, uint32_t *d_out, hipStream_t s);
    uint64_t range() const { return range_; }

  private:
    DevBuf<RngState> st_;
    uint32_t range_ = 0, thr_ = 0;
};

}  // namespace cymf
