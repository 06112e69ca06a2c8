// comm.hip -- the one collective of the user-sharded design: sum of the ranks'
// item-factor deltas, RCCL all-reduce over xGMI, one process per GPU.  The reference has no
// counterpart (single process, OpenMP only: cymf/bpr.pyx:162).
#include <rccl/rccl.h>

#include "common.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>

static_assert(NCCL_UNIQUE_ID_BYTES == CYMF_UNIQUE_ID_BYTES, "unique id size");

// A "local group" is N communicator handles inside ONE process on ONE device (cymf_comm_create_local_group): the
// collectives are small kernels that meet in device memory.  It exists so that the sharded trainers can be run
// with several ranks on a one-GPU test box -- RCCL refuses two ranks on one device -- one host thread per rank.
struct LocalGroup {
    int world = 0, device = 0, refs = 0;
    size_t cap = 0;                      // floats per slot
    float *slots = nullptr;              // [2 parities][world][cap], uncached
    // host-side meeting point of the ranks' threads (no kernel ever waits on the device: a spinning kernel would
    // deadlock against the device-wide synchronisation inside another rank's hipFree)
    std::mutex mu;
    std::condition_variable cv;
    long long arrived = 0;               // total arrivals
    bool broken = false;                 // a rank gave up waiting
};

struct cymf_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    LocalGroup *grp = nullptr;
    long long gen = 0;                   // collectives issued by this rank (all ranks issue the same sequence)
    bool failed = false;                 // an RCCL call on this communicator returned an error: destroy aborts it
};

// RCCL calls that do not involve a communicator
#define CYMF_NCCL(expr)                                                                         \
    do {                                                                                        \
        ncclResult_t r__ = (expr);                                                              \
        if (r__ != ncclSuccess)                                                                 \
            return ::cymf::fail(CYMF_ERR_RCCL, "%s failed: %s", #expr, ncclGetErrorString(r__)); \
    } while (0)

// RCCL calls on communicator c: a failure marks the communicator (cymf_comm_destroy then calls ncclCommAbort: an
// ncclCommDestroy of a communicator with a failed or half-issued collective can block for ever waiting for its peers)
#define CYMF_NCCL_C(c, expr)                                                                    \
    do {                                                                                        \
        ncclResult_t r__ = (expr);                                                              \
        if (r__ != ncclSuccess) {                                                               \
            (c)->failed = true;                                                                 \
            return ::cymf::fail(CYMF_ERR_RCCL, "%s failed on rank %d of %d: %s", #expr, (c)->rank, (c)->world, ncclGetErrorString(r__)); \
        }                                                                                       \
    } while (0)


namespace cymf {
namespace {

__global__ void local_publish_kernel(const float *__restrict__ in, float *__restrict__ slot, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) slot[i] = in[i];
}
// out = reduction over the ranks' slots in rank order (the same bits on every rank); op 0 sum, 1 max
__global__ void local_reduce_kernel(const float *__restrict__ slots, size_t cap, int world, float *__restrict__ out, int64_t n, int op) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = slots[i];
        for (int r = 1; r < world; ++r) { const float w = slots[(size_t)r * cap + i]; v = op == 1 ? (w > v ? w : v) : v + w; }
        out[i] = v;
    }
}
__global__ void local_copy_bytes_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

inline int blocks_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }

// every rank: publish my data into my slot of parity p (and wait for that on the host), meet the other ranks'
// threads, then `consume` all slots (also waited for).  Two parities suffice: a rank reaches collective g+2 only
// after every rank has arrived at g+1, i.e. has left collective g with its reads done.
template <typename Publish, typename Consume>
int local_collective(cymf_comm *c, size_t floats_needed, hipStream_t s, Publish publish, Consume consume) {
    LocalGroup *g = c->grp;
    if (floats_needed > g->cap) return fail(CYMF_ERR_UNSUPPORTED, "local group: %zu floats exceed the slot size %zu", floats_needed, g->cap);
    const int p = (int)(c->gen & 1);
    float *my_slot = g->slots + ((size_t)p * g->world + c->rank) * g->cap;
    publish(my_slot);
    CYMF_HIP(hipGetLastError());
    CYMF_HIP(hipStreamSynchronize(s));
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->arrived++;
        const long long target = (long long)g->world * (c->gen + 1);
        g->cv.notify_all();
        if (!g->cv.wait_for(lk, std::chrono::seconds(20), [&] { return g->arrived >= target || g->broken; }) || g->broken) {
            g->broken = true;
            g->cv.notify_all();
            return fail(CYMF_ERR_RCCL, "local group: rank %d waited 20 s for the others at collective %lld", c->rank, c->gen);
        }
    }
    consume(g->slots + (size_t)p * g->world * g->cap);
    CYMF_HIP(hipGetLastError());
    CYMF_HIP(hipStreamSynchronize(s));   // a rank leaves the collective with its reads of the slots done
    c->gen++;
    return 0;
}

int local_allreduce(cymf_comm *c, const float *d_in, float *d_out, int64_t n, int op, hipStream_t s) {
    return local_collective(c, (size_t)n, s,
        [&](float *slot) { hipLaunchKernelGGL(local_publish_kernel, dim3(blocks_for(n)), dim3(256), 0, s, d_in, slot, n); },
        [&](const float *slots) { hipLaunchKernelGGL(local_reduce_kernel, dim3(blocks_for(n)), dim3(256), 0, s, slots, c->grp->cap, c->world, d_out, n, op); });
}

// The padded reduce-scatter + all-gather layout on the local group (n % world != 0 included), so that the offsets and the
// padding rule of the CYMF_COMM_RS_AG path run on the one-GPU box: every rank publishes its padded input and reduces the
// ranks' copies of ITS shard of padded / world floats into place in d_out; the ranks meet again, publish their shards and
// copy the others' into place.  Two meetings, as the two RCCL calls are two collectives; equal to local_allreduce bit for
// bit (the same rank order of the additions).
int local_rs_ag(cymf_comm *c, const float *d_in, float *d_out, int64_t n, int64_t cap, hipStream_t s) {
    const int64_t w = c->world, padded = (n + w - 1) / w * w, shard = padded / w;
    if (padded > cap) return fail(CYMF_ERR_INVALID, "local group: reduce-scatter + all-gather of %lld floats over %d ranks needs buffers of %lld, have %lld",
                                  (long long)n, c->world, (long long)padded, (long long)cap);
    if ((size_t)padded > c->grp->cap) return fail(CYMF_ERR_INVALID, "local group: %lld floats exceed the slot size %zu", (long long)padded, c->grp->cap);
    float *my_shard = d_out + (size_t)c->rank * shard;
    // reduce-scatter: publish the padded input; reduce the ranks' copies of MY shard
    CYMF_TRY(local_collective(c, (size_t)padded, s,
        [&](float *slot) { hipLaunchKernelGGL(local_publish_kernel, dim3(blocks_for(padded)), dim3(256), 0, s, d_in, slot, padded); },
        [&](const float *slots) {
            hipLaunchKernelGGL(local_reduce_kernel, dim3(blocks_for(shard)), dim3(256), 0, s, slots + (size_t)c->rank * shard, c->grp->cap, c->world, my_shard, shard, 0);
        }));
    // all-gather: publish my reduced shard; copy every rank's shard into place
    return local_collective(c, (size_t)shard, s,
        [&](float *slot) { hipLaunchKernelGGL(local_publish_kernel, dim3(blocks_for(shard)), dim3(256), 0, s, my_shard, slot, shard); },
        [&](const float *slots) {
            for (int r = 0; r < c->world; ++r) {
                if (r == c->rank) continue;
                hipLaunchKernelGGL(local_publish_kernel, dim3(blocks_for(shard)), dim3(256), 0, s, slots + (size_t)r * c->grp->cap, d_out + (size_t)r * shard, shard);
            }
        });
}

}  // namespace
}  // namespace cymf

namespace cymf {

// Errors of enqueued collectives surface asynchronously (a peer died, a link went down): asked for after every call
// that enqueues work, so that the NEXT entry point fails with the reason instead of hanging in a stream wait.
static int comm_check_async(cymf_comm *c, const char *what) {
    ncclResult_t aerr = ncclSuccess;
    ncclResult_t r = ncclCommGetAsyncError(c->comm, &aerr);
    if (r != ncclSuccess || (aerr != ncclSuccess && aerr != ncclInProgress)) {
        c->failed = true;
        return fail(CYMF_ERR_RCCL, "%s: asynchronous RCCL error on rank %d of %d: %s", what, c->rank, c->world,
                    ncclGetErrorString(r != ncclSuccess ? r : aerr));
    }
    return 0;
}

int comm_allreduce_sum_f32(cymf_comm *c, float *d_buf, int64_t n, hipStream_t s) {
    if (c->grp) return local_allreduce(c, d_buf, d_buf, n, 0, s);
    if (c->failed) return fail(CYMF_ERR_RCCL, "communicator of rank %d is in a failed state", c->rank);
    CYMF_NCCL_C(c, ncclAllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, c->comm, s));
    return comm_check_async(c, "ncclAllReduce");
}

// Elements a buffer must hold for comm_allreduce_sum_f32_to: n rounded up to a multiple of the world size (the
// reduce-scatter / all-gather pair works on equal shards; the caller keeps the padding zero).
int64_t comm_padded_count(cymf_comm *c, int64_t n) {
    const int64_t w = c ? c->world : 1;
    return (n + w - 1) / w * w;
}

// d_out = sum over the ranks of d_in (out of place; `cap` = floats BOTH buffers hold).
//  * default: one ncclAllReduce.  RCCL's ring all-reduce already is a reduce-scatter followed by an all-gather, and no
//    run with more than one rank has been measured yet (the development box has one GPU), so the library's own choice of
//    algorithm stands until a measurement says otherwise.
//  * CYMF_COMM_RS_AG=1: the two halves issued explicitly on padded equal shards (per link (world - 1) / world
//    of the table each way, every shard reduced once).  Two calls in stream order (not one group: the all-gather reads what
//    the reduce-scatter wrote, and a group promises no order between its members).  Needs cap >= comm_padded_count(n) and the
//    padding of d_in zero (the callers allocate and zero both at upload); checked here.
static bool comm_rs_ag() {   // (read on every call: the tests switch it inside one process)
    const char *e = getenv("CYMF_COMM_RS_AG");
    return e && e[0] == '1';
}

int comm_allreduce_sum_f32_to(cymf_comm *c, const float *d_in, float *d_out, int64_t n, int64_t cap, hipStream_t s) {
    if (n > cap) return fail(CYMF_ERR_INVALID, "comm_allreduce_sum_f32_to: %lld floats into buffers of %lld", (long long)n, (long long)cap);
    if (c->grp) {
        if (comm_rs_ag()) return local_rs_ag(c, d_in, d_out, n, cap, s);
        return local_allreduce(c, d_in, d_out, n, 0, s);
    }
    if (c->failed) return fail(CYMF_ERR_RCCL, "communicator of rank %d is in a failed state", c->rank);
    if (!comm_rs_ag()) {
        CYMF_NCCL_C(c, ncclAllReduce(d_in, d_out, (size_t)n, ncclFloat32, ncclSum, c->comm, s));
        return comm_check_async(c, "ncclAllReduce");
    }
    const int64_t padded = comm_padded_count(c, n);
    if (padded > cap)
        return fail(CYMF_ERR_INVALID, "comm_allreduce_sum_f32_to: reduce-scatter + all-gather of %lld floats over %d ranks needs buffers of %lld, have %lld",
                    (long long)n, c->world, (long long)padded, (long long)cap);
    const int64_t shard = padded / c->world;
    CYMF_NCCL_C(c, ncclReduceScatter(d_in, d_out + (size_t)c->rank * shard, (size_t)shard, ncclFloat32, ncclSum, c->comm, s));
    CYMF_NCCL_C(c, ncclAllGather(d_out + (size_t)c->rank * shard, d_out, (size_t)shard, ncclFloat32, c->comm, s));
    return comm_check_async(c, "reduce-scatter + all-gather");
}

int comm_world(cymf_comm *c) { return c ? c->world : 1; }
int comm_rank(cymf_comm *c) { return c ? c->rank : 0; }

// In-place all-gather of unequal contiguous row ranges: rank r owns rows [row_bounds[r], row_bounds[r+1]) of the
// table at d_buf and every rank ends with all of them.  One grouped set of broadcasts (RCCL fuses the group).
int comm_allgatherv(cymf_comm *c, void *d_buf, const int64_t *row_bounds, int64_t row_bytes, hipStream_t s) {
    if (c->world == 1) return 0;
    if (c->grp) {   // my rows into my slot (as bytes), then everybody copies every other rank's rows out of the slots
        const int64_t my_bytes = (row_bounds[c->rank + 1] - row_bounds[c->rank]) * row_bytes;
        int64_t max_bytes = 0;
        for (int r = 0; r < c->world; ++r) max_bytes = std::max<int64_t>(max_bytes, (row_bounds[r + 1] - row_bounds[r]) * row_bytes);
        unsigned char *base = static_cast<unsigned char *>(d_buf);
        return local_collective(c, (size_t)(max_bytes + 3) / 4, s,
            [&](float *slot) {
                if (my_bytes > 0)
                    hipLaunchKernelGGL(local_copy_bytes_kernel, dim3(blocks_for(my_bytes)), dim3(256), 0, s, base + row_bounds[c->rank] * row_bytes,
                                       reinterpret_cast<unsigned char *>(slot), my_bytes);
            },
            [&](const float *slots) {
                for (int r = 0; r < c->world; ++r) {
                    const int64_t nb = (row_bounds[r + 1] - row_bounds[r]) * row_bytes;
                    if (r == c->rank || nb <= 0) continue;
                    hipLaunchKernelGGL(local_copy_bytes_kernel, dim3(blocks_for(nb)), dim3(256), 0, s,
                                       reinterpret_cast<const unsigned char *>(slots + (size_t)r * c->grp->cap), base + row_bounds[r] * row_bytes, nb);
                }
            });
    }
    if (c->failed) return fail(CYMF_ERR_RCCL, "communicator of rank %d is in a failed state", c->rank);
    CYMF_NCCL_C(c, ncclGroupStart());
    for (int r = 0; r < c->world; ++r) {
        const int64_t n = (row_bounds[r + 1] - row_bounds[r]) * row_bytes;
        if (n <= 0) continue;
        char *p = static_cast<char *>(d_buf) + row_bounds[r] * row_bytes;
        ncclResult_t rc = ncclBroadcast(p, p, (size_t)n, ncclChar, r, c->comm, s);
        if (rc != ncclSuccess) {
            (void)ncclGroupEnd();
            c->failed = true;
            return ::cymf::fail(CYMF_ERR_RCCL, "ncclBroadcast failed on rank %d of %d: %s", c->rank, c->world, ncclGetErrorString(rc));
        }
    }
    CYMF_NCCL_C(c, ncclGroupEnd());
    return comm_check_async(c, "grouped broadcasts");
}

}  // namespace cymf

using namespace cymf;

extern "C" int cymf_comm_unique_id(char id[CYMF_UNIQUE_ID_BYTES]) {
    if (!id) return fail(CYMF_ERR_INVALID, "cymf_comm_unique_id: NULL buffer");
    ncclUniqueId uid;
    CYMF_NCCL(ncclGetUniqueId(&uid));
    memcpy(id, uid.internal, CYMF_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int cymf_comm_create(cymf_comm **out, const char id[CYMF_UNIQUE_ID_BYTES], int rank, int world, int device) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(CYMF_ERR_INVALID, "cymf_comm_create: bad arguments");
    *out = nullptr;
    CYMF_TRY(use_device(device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, CYMF_UNIQUE_ID_BYTES);
    cymf_comm *c = new cymf_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(CYMF_ERR_RCCL, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    *out = c;
    return 0;
}

extern "C" int cymf_comm_create_local_group(cymf_comm **out, int world, int device, int64_t max_floats) {
    if (!out || world < 1 || world > 16 || max_floats < 1) return fail(CYMF_ERR_INVALID, "cymf_comm_create_local_group: bad arguments");
    CYMF_TRY(use_device(device));
    LocalGroup *g = new LocalGroup();
    g->world = world; g->device = device; g->refs = world; g->cap = ((size_t)max_floats + 63) & ~(size_t)63;
    hipError_t e = hipExtMallocWithFlags((void **)&g->slots, 2 * (size_t)world * g->cap * sizeof(float), hipDeviceMallocUncached);
    if (e != hipSuccess) {
        if (g->slots) (void)hipFree(g->slots);
        delete g;
        return fail(CYMF_ERR_NOMEM, "cymf_comm_create_local_group: %s", hipGetErrorString(e));
    }
    for (int r = 0; r < world; ++r) {
        cymf_comm *c = new cymf_comm();
        c->rank = r; c->world = world; c->device = device; c->grp = g;
        out[r] = c;
    }
    return 0;
}

extern "C" int cymf_comm_destroy(cymf_comm *c) {
    if (!c) return 0;
    if (!cymf::runtime_alive(c->device)) return 0;   // process exit / runtime already torn down: leak quietly
    if (c->comm) {
        // a communicator on which a call failed (or whose peers may be gone) is aborted: ncclCommDestroy would wait for
        // collectives that can never complete
        ncclResult_t aerr = ncclSuccess;
        if (!c->failed && ncclCommGetAsyncError(c->comm, &aerr) == ncclSuccess && aerr == ncclSuccess) (void)ncclCommDestroy(c->comm);
        else (void)ncclCommAbort(c->comm);
    }
    if (c->grp && --c->grp->refs == 0) {   // (handles of a local group are destroyed from one thread, after the ranks have joined)
        (void)hipDeviceSynchronize();
        (void)hipFree(c->grp->slots);
        delete c->grp;
    }
    delete c;
    return 0;
}

extern "C" int cymf_comm_allreduce_f32(cymf_comm *c, float *host_inout, int64_t n, int op) {
    if (!c || !host_inout || n < 0) return fail(CYMF_ERR_INVALID, "cymf_comm_allreduce_f32: bad arguments");
    CYMF_TRY(use_device(c->device));
    // (no device-wide synchronisation on the local group: another rank's thread may be blocked inside a collective)
    hipStream_t s = nullptr;
    CYMF_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const bool rs_ag = op != 1 && comm_rs_ag();       // sums take the same route as the trainers' exchange (tests)
    const size_t cap = rs_ag ? (size_t)comm_padded_count(c, n) : (size_t)n;
    DevBuf<float> din, dout;
    int rc = din.alloc(std::max<size_t>(cap, 1));
    if (!rc) rc = dout.alloc(std::max<size_t>(cap, 1));
    if (!rc) rc = din.zero(s);                        // the padding of the reduce-scatter input must be zero
    if (!rc && n > 0 && hipMemcpyAsync(din.p, host_inout, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) rc = fail(CYMF_ERR_HIP, "copy failed");
    if (!rc && !c->grp && hipStreamSynchronize(s) != hipSuccess) rc = fail(CYMF_ERR_HIP, "sync failed");
    if (!rc) {
        if (rs_ag) rc = comm_allreduce_sum_f32_to(c, din.p, dout.p, n, (int64_t)cap, s);
        else if (c->grp) rc = local_allreduce(c, din.p, dout.p, n, op == 1 ? 1 : 0, s);
        else if (c->failed) rc = fail(CYMF_ERR_RCCL, "communicator of rank %d is in a failed state", c->rank);
        else {
            ncclResult_t r = ncclAllReduce(din.p, dout.p, (size_t)n, ncclFloat32, op == 1 ? ncclMax : ncclSum, c->comm, s);
            if (r != ncclSuccess) { c->failed = true; rc = fail(CYMF_ERR_RCCL, "ncclAllReduce failed on rank %d of %d: %s", c->rank, c->world, ncclGetErrorString(r)); }
            else rc = comm_check_async(c, "ncclAllReduce");
        }
    }
    if (!rc && n > 0 && hipMemcpyAsync(host_inout, dout.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(CYMF_ERR_HIP, "copy failed");
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = fail(CYMF_ERR_HIP, "sync failed");
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    return rc;
}
