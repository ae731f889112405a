// Data-parallel exchange of libvq2: an RCCL communicator owned by the library (the one piece of persistent state
// SURVEY 8b allows), one per process = one per GPU.  It carries what the reference moves with
//   dist.all_reduce at vqvae.py:58-59 (through distributed.py:64-72)   -- EMA sums, SUM
//   DistributedDataParallel at train_vqvae.py:166-171                  -- gradients (SUM; Adam divides) + initial broadcast
// over xGMI.  RCCL is resolved at run time (dlopen of librccl.so.1: in a PyTorch process that is the copy torch
// already loaded), so single-GPU users of libvq2.so need no RCCL at all and there is no second copy of the library.
#include "vq2_common.h"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include <string.h>

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;

int load_rccl() {
    if (g_rccl.h) return VQ2_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return vq2::set_error(VQ2_ERR_UNSUPPORTED, "vq2_comm: cannot load RCCL (%s)", dlerror());
    Rccl r;
    r.h = h;
#define VQ2_SYM(field, sym)                                                                     \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym));                               \
    if (!r.field) return vq2::set_error(VQ2_ERR_UNSUPPORTED, "vq2_comm: RCCL lacks %s", sym);
    VQ2_SYM(GetUniqueId, "ncclGetUniqueId")
    VQ2_SYM(CommInitRank, "ncclCommInitRank")
    VQ2_SYM(CommDestroy, "ncclCommDestroy")
    VQ2_SYM(AllReduce, "ncclAllReduce")
    VQ2_SYM(Broadcast, "ncclBroadcast")
    VQ2_SYM(GetErrorString, "ncclGetErrorString")
#undef VQ2_SYM
    g_rccl = r;
    return VQ2_OK;
}

int nccl_fail(const char *what, ncclResult_t e) {
    return vq2::set_error(VQ2_ERR_LAUNCH, "vq2_comm: %s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
}

}  // namespace

static_assert(VQ2_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

extern "C" int vq2_comm_unique_id(void *id) {
    VQ2_REQUIRE(id, "vq2_comm_unique_id: null pointer");
    std::lock_guard<std::mutex> lk(g_mu);
    if (int e = load_rccl()) return e;
    ncclUniqueId uid;
    const ncclResult_t r = g_rccl.GetUniqueId(&uid);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    memcpy(id, &uid, VQ2_COMM_ID_BYTES);
    return VQ2_OK;
}

extern "C" int vq2_comm_init(const void *id, int32_t rank, int32_t world) {
    VQ2_REQUIRE(id && world >= 1 && rank >= 0 && rank < world, "vq2_comm_init: bad arguments (rank %d of %d)", rank, world);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm) return vq2::set_error(VQ2_ERR_INVALID, "vq2_comm_init: a communicator already exists (one per process)");
    if (int e = load_rccl()) return e;
    ncclUniqueId uid;
    memcpy(&uid, id, VQ2_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&c, world, uid, rank);   // binds to the calling thread's current device
    if (r != ncclSuccess) return nccl_fail("ncclCommInitRank", r);
    g_comm = c; g_world = world; g_rank = rank;
    return VQ2_OK;
}

extern "C" int vq2_comm_world(void) { return g_world; }
extern "C" int vq2_comm_rank(void) { return g_rank; }

extern "C" int vq2_comm_allreduce_sum(float *buf, int64_t count, vq2_stream_t stream) {
    VQ2_REQUIRE(buf && count > 0, "vq2_comm_allreduce_sum: bad arguments");
    if (!g_comm) return vq2::set_error(VQ2_ERR_INVALID, "vq2_comm_allreduce_sum: vq2_comm_init was not called");
    const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, g_comm, vq2::to_stream(stream));
    return r == ncclSuccess ? VQ2_OK : nccl_fail("ncclAllReduce", r);
}

extern "C" int vq2_comm_broadcast(float *buf, int64_t count, int32_t root, vq2_stream_t stream) {
    VQ2_REQUIRE(buf && count > 0 && root >= 0, "vq2_comm_broadcast: bad arguments");
    if (!g_comm) return vq2::set_error(VQ2_ERR_INVALID, "vq2_comm_broadcast: vq2_comm_init was not called");
    VQ2_REQUIRE(root < g_world, "vq2_comm_broadcast: root %d outside the communicator (%d ranks)", root, g_world);
    const ncclResult_t r = g_rccl.Broadcast(buf, buf, (size_t)count, ncclFloat, root, g_comm, vq2::to_stream(stream));
    return r == ncclSuccess ? VQ2_OK : nccl_fail("ncclBroadcast", r);
}

extern "C" int vq2_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return VQ2_OK;
    const ncclResult_t r = g_rccl.CommDestroy(g_comm);
    g_comm = nullptr; g_world = 0; g_rank = -1;
    return r == ncclSuccess ? VQ2_OK : nccl_fail("ncclCommDestroy", r);
}
