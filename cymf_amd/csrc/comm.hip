// comm.hip -- the one collective of the user-sharded design: sum of the ranks'
// item-factor deltas, RCCL all-reduce over xGMI, one process per GPU.  The reference has no
// counterpart (single process, OpenMP only: cymf/bpr.pyx:162).
#include <rccl/rccl.h>

#include "common.h"

static_assert(NCCL_UNIQUE_ID_BYTES == CYMF_UNIQUE_ID_BYTES, "unique id size");

struct cymf_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

#define CYMF_NCCL(expr)                                                                         \
    do {                                                                                        \
        ncclResult_t r__ = (expr);                                                              \
        if (r__ != ncclSuccess)                                                                 \
            return ::cymf::fail(CYMF_ERR_RCCL, "%s failed: %s", #expr, ncclGetErrorString(r__)); \
    } while (0)

namespace cymf {

int comm_allreduce_sum_f32(cymf_comm *c, float *d_buf, int64_t n, hipStream_t s) {
    CYMF_NCCL(ncclAllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, c->comm, s));
    return 0;
}

int comm_allreduce_sum_f32_to(cymf_comm *c, const float *d_in, float *d_out, int64_t n, hipStream_t s) {
    CYMF_NCCL(ncclAllReduce(d_in, d_out, (size_t)n, ncclFloat32, ncclSum, c->comm, s));
    return 0;
}

int comm_world(cymf_comm *c) { return c ? c->world : 1; }
int comm_rank(cymf_comm *c) { return c ? c->rank : 0; }

// In-place all-gather of unequal contiguous row ranges: rank r owns rows [row_bounds[r], row_bounds[r+1]) of the
// table at d_buf and every rank ends with all of them.  One grouped set of broadcasts (RCCL fuses the group).
int comm_allgatherv(cymf_comm *c, void *d_buf, const int64_t *row_bounds, int64_t row_bytes, hipStream_t s) {
    if (c->world == 1) return 0;
    CYMF_NCCL(ncclGroupStart());
    for (int r = 0; r < c->world; ++r) {
        const int64_t n = (row_bounds[r + 1] - row_bounds[r]) * row_bytes;
        if (n <= 0) continue;
        char *p = static_cast<char *>(d_buf) + row_bounds[r] * row_bytes;
        ncclResult_t rc = ncclBroadcast(p, p, (size_t)n, ncclChar, r, c->comm, s);
        if (rc != ncclSuccess) {
            (void)ncclGroupEnd();
            return ::cymf::fail(CYMF_ERR_RCCL, "ncclBroadcast failed: %s", ncclGetErrorString(rc));
        }
    }
    CYMF_NCCL(ncclGroupEnd());
    return 0;
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

extern "C" int cymf_comm_destroy(cymf_comm *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
    return 0;
}

extern "C" int cymf_comm_allreduce_f32(cymf_comm *c, float *host_inout, int64_t n, int op) {
    if (!c || !host_inout || n < 0) return fail(CYMF_ERR_INVALID, "cymf_comm_allreduce_f32: bad arguments");
    CYMF_TRY(use_device(c->device));
    DevBuf<float> d;
    CYMF_TRY(d.upload(host_inout, (size_t)n));
    CYMF_HIP(hipDeviceSynchronize());
    CYMF_NCCL(ncclAllReduce(d.p, d.p, (size_t)n, ncclFloat32, op == 1 ? ncclMax : ncclSum, c->comm, nullptr));
    CYMF_HIP(hipMemcpy(host_inout, d.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
