// comm.hip -- the one collective of the hot path behind the C ABI: the end-of-forward all-gather of a rank's outputs over RCCL / xGMI
// (batch-sharded forward, 256 x 1000 fp32 logits = 1 MB per rank for ViT-Base).  One process per GPU; the caller
// distributes the 128-byte unique id (rank 0 creates it) through whatever bootstrap it has -- the Python mirror uses the
// torch.distributed store, a C host would use its own -- and then owns an opaque communicator handle.
//
// RCCL is bound at run time (dlopen / dlsym), preferring the copy the process has already loaded (torch ships its own librccl with
// the same SONAME): one RCCL per process, no link-time dependency, and the library still loads on a host without RCCL -- the comm
// entry points then fail with MI355_EUNSUPPORTED instead.
#include "common.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {                           // a copy that is already mapped wins (RTLD_NOLOAD)
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        for (int i = 0; !g_rccl.handle && i < 3; ++i) g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.handle) return;
        g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.handle, "ncclGetUniqueId"));
        g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.handle, "ncclCommInitRank"));
        g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(g_rccl.handle, "ncclAllGather"));
        g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.handle, "ncclCommDestroy"));
        g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.handle, "ncclGetErrorString"));
        g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllGather && g_rccl.CommDestroy && g_rccl.GetErrorString;
    });
    return g_rccl;
}

struct Comm {
    ncclComm_t comm;
    int rank, world;
};

int nccl_fail(const char* what, ncclResult_t r) {
    return mi355::fail(MI355_EHIP, "%s: RCCL -> %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
}

}  // namespace

extern "C" {

int mi355_comm_unique_id(void* id_out, size_t id_bytes) {
    MI355_CHECK_ARG(id_out != nullptr && id_bytes >= MI355_COMM_ID_BYTES);
    static_assert(MI355_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!rccl().ok) return mi355::fail(MI355_EUNSUPPORTED, "mi355_comm_unique_id: librccl could not be loaded");
    ncclUniqueId id;
    const ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("mi355_comm_unique_id", r);
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return MI355_OK;
}

int mi355_comm_init(const void* id, size_t id_bytes, int rank, int world, void** comm_out) {
    MI355_CHECK_ARG(id != nullptr && id_bytes >= MI355_COMM_ID_BYTES && comm_out != nullptr && world >= 1 && rank >= 0 && rank < world);
    if (!rccl().ok) return mi355::fail(MI355_EUNSUPPORTED, "mi355_comm_init: librccl could not be loaded");
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    Comm* c = new Comm{nullptr, rank, world};
    const ncclResult_t r = rccl().CommInitRank(&c->comm, world, uid, rank);       // binds to the calling thread's current HIP device
    if (r != ncclSuccess) {
        delete c;
        return nccl_fail("mi355_comm_init", r);
    }
    *comm_out = c;
    return MI355_OK;
}

int mi355_allgather_f32(void* comm, const float* send, float* recv, size_t count, mi355_stream_t stream) {
    MI355_CHECK_ARG(comm != nullptr && send != nullptr && recv != nullptr && count > 0);
    Comm* c = static_cast<Comm*>(comm);
    const ncclResult_t r = rccl().AllGather(send, recv, count, ncclFloat, c->comm, static_cast<hipStream_t>(stream));
    if (r != ncclSuccess) return nccl_fail("mi355_allgather_f32", r);
    return MI355_OK;
}

int mi355_comm_destroy(void* comm) {
    MI355_CHECK_ARG(comm != nullptr);
    Comm* c = static_cast<Comm*>(comm);
    const ncclResult_t r = rccl().CommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) return nccl_fail("mi355_comm_destroy", r);
    return MI355_OK;
}

}  // extern "C"
