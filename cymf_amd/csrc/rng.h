// rng.h -- device-resident negative-sample index stream (bit-exact std::mt19937 + libstdc++-11
// uniform_int_distribution<long>, i.e. the reference's UniformGenerator, cymf/math.pyx:12-18).
#pragma once
#include <vector>
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
//   serial mode   : one workgroup walks the stream (small problems, high-rejection ranges).
//   parallel mode : the raw stream is cut into chunks of MT_JUMP_WORDS words; chunk start states
//                   come from the jump-ahead polynomial (mt_jump_poly.h), one workgroup per chunk
//                   generates and compacts its accepted words, a last pass gathers them in order.
class DeviceRng {
  public:
    ~DeviceRng();
    // range in [1, 2^62].  Below 2^32: 32-bit draws (generate); from 2^32 on: 64-bit draws by the one-lane walker
    // (generate64; libstdc++'s non-Lemire branches).  parallel = use the chunked generator (the one-workgroup walker still serves
    // ranges whose Lemire rejection rate exceeds ~8 %: the per-chunk rejection list would outgrow 2 MB)
    int init(uint32_t seed, uint64_t range, hipStream_t s, bool parallel = false);
    // Discards n_skip draws, then writes the next n draws to d_out[0..n) (device pointer), in
    // stream order on `s`.  Asynchronous; the stream position is settled lazily (finalize).
    int generate(int64_t n_skip, int64_t n, uint32_t *d_out, hipStream_t s);
    int generate64(int64_t n_skip, int64_t n, uint64_t *d_out, hipStream_t s);   // streams with range >= 2^32 only
    bool wide() const { return wide_; }
    uint64_t range() const { return wide_ ? urange_ + 1 : range_; }
    bool parallel() const { return parallel_; }

  private:
    int generate_parallel(int64_t n_total, int64_t n_skip, uint32_t *d_out, hipStream_t s);
    int finalize();   // host: wait for the last parallel launch and advance raw_pos_
    int ensure_states(int64_t last_chunk, hipStream_t s);

    DevBuf<RngState> st_;
    uint32_t range_ = 0, thr_ = 0;      // (wide: range and Lemire threshold of the HIGH part)
    bool wide_ = false;
    uint64_t urange_ = 0;               // wide: range - 1
    bool parallel_ = false;
    int64_t rej_cap_ = 0;               // rejection-list entries per chunk
    // parallel mode
    uint64_t raw_pos_ = 0;              // next unconsumed raw word of the stream
    DevBuf<uint32_t> poly_, states_, tmp_, counts_, rej_, rej_cnt_;
    int64_t states_cap_ = 0, states_known_ = 0;   // chunk start states [0, states_known_) are valid
    std::vector<uint32_t *> retired_;   // outgrown state tables, freed with the generator: a hipFree is a device-wide synchronisation, and the
                                        // table is outgrown in the middle of an epoch whose kernel the caller wants to work beside
    struct Pending {
        bool active = false;
        hipEvent_t done = nullptr;
        uint32_t *h_counts = nullptr, *h_rej = nullptr, *h_rej_cnt = nullptr;   // pinned
        int64_t cap_chunks = 0;
        int64_t c0 = 0, n_chunks = 0, n_total = 0;
        uint64_t skip0 = 0;
    } pend_;
};

}  // namespace cymf
