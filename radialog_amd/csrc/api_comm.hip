// librdx C ABI, part 4: the one collective of the path -- the all-gather of generated token ids over RCCL / xGMI.
#include <dlfcn.h>

#include "rdx_ctx.h"

// ------------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time: librccl is only needed by multi-GPU jobs, and the process usually has torch's copy loaded already
// (same soname -> the same instance is shared). No RCCL type crosses the C ABI: the unique id travels as 128 opaque bytes.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, rdx_unique_id, int) = nullptr;       // ncclUniqueId is a 128-byte struct passed by value
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    bool load() {
        if (h) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;       // torch's instance, if it is there
        if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (int (*)(void**, int, rdx_unique_id, int))dlsym(h, "ncclCommInitRank");
        AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
        CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
        GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !GetErrorString) { err = "librccl lacks an expected symbol"; h = nullptr; return false; }
        return true;
    }
    const char* what(int rc) { return GetErrorString ? GetErrorString(rc) : "?"; }
};
Rccl g_rccl;
constexpr int kNcclInt32 = 2;        // ncclInt32 in rccl.h's ncclDataType_t
}  // namespace

extern "C" int rdx_comm_unique_id(rdx_unique_id* id_host) {
    if (!id_host) return fail(nullptr, -1, "rdx_comm_unique_id: null argument");
    if (!g_rccl.load()) return fail(nullptr, -6, "rdx_comm_unique_id: %s", g_rccl.err.c_str());
    const int rc = g_rccl.GetUniqueId(id_host);
    if (rc) return fail(nullptr, -6, "ncclGetUniqueId failed: %s", g_rccl.what(rc));
    return 0;
}

extern "C" int rdx_comm_init(rdx_ctx* c, const rdx_unique_id* id_host, int rank, int world) {
    if (!c || !id_host) return fail(c, -1, "rdx_comm_init: null argument");
    if (world <= 0 || rank < 0 || rank >= world) return fail(c, -1, "rdx_comm_init: bad rank %d / world %d", rank, world);
    if (c->comm) return fail(c, -1, "rdx_comm_init: communicator already initialised");
    if (!g_rccl.load()) return fail(c, -6, "rdx_comm_init: %s", g_rccl.err.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = g_rccl.CommInitRank(&c->comm, world, *id_host, rank);
    if (rc) { c->comm = nullptr; return fail(c, -6, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.what(rc)); }
    c->comm_rank = rank; c->comm_world = world;
    return 0;
}

extern "C" int rdx_allgather_tokens(rdx_ctx* c, const int32_t* local, int32_t* global, int rows_local, int n) {
    if (!c || !local || !global || rows_local <= 0 || n <= 0) return fail(c, -1, "rdx_allgather_tokens: bad arguments");
    if (!c->comm) return fail(c, -1, "rdx_allgather_tokens: rdx_comm_init has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = g_rccl.AllGather(local, global, (size_t)rows_local * n, kNcclInt32, c->comm, c->stream);
    if (rc) return fail(c, -6, "ncclAllGather failed: %s", g_rccl.what(rc));
    return 0;
}

extern "C" int rdx_comm_world(rdx_ctx* c) { return (c && c->comm) ? c->comm_world : 0; }

void rdx_comm_release(rdx_ctx* c) {
    if (c->comm && g_rccl.CommDestroy) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
}
