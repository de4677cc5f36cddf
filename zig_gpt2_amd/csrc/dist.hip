// dist.hip — the one collective of the multi-GPU case: prompts are independent units sharded over one process
// per GPU, the weights are replicated — rank 0 loads them (load_gpt, src/main.zig:304-314) and ONE broadcast of the weight
// region of the arena (zg_gpt_weight_arena) carries them to the other GPUs of the node over xGMI; optionally the ranks'
// token matrices are gathered at the end.  Nothing is exchanged while tokens are generated.  The reference has no
// multi-device code.
//
// RCCL is bound at run time (dlopen of librccl.so; a copy already in the process — PyTorch's — is preferred, so that one
// process never runs two): a one-GPU box without RCCL still loads libzgpt2_hip.so, and zg_dist_* then fail with a message.
// Nothing of RCCL is needed at build time either: the handful of types and constants its C API passes are declared here.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "zg_runtime.h"

// the part of rccl.h this file uses (stable NCCL ABI: a 128-byte opaque id, an opaque communicator, result 0 = success,
// ncclUint8 = 1)
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclUint8 = 1;

namespace zg {
namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 0;  // world 0 = no communicator

int load_rccl() {
    if (g_rccl.lib) return ZG_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    // ZGPT2_RCCL_LIB (test hook): bind the six entry points from this library instead — tests/stub_rccl/ moves the bytes through
    // files so that a world of several ranks can run on a one-GPU box, where RCCL refuses two ranks on one device
    if (const char* forced = getenv("ZGPT2_RCCL_LIB")) {
        lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!lib) {
            const char* why = dlerror();
            ZG_REQUIRE(false, ZG_ERR_UNSUPPORTED, "multi-GPU: ZGPT2_RCCL_LIB=%s cannot be loaded (%s)", forced, why ? why : "no loader message");
        }
    }
    for (const char* n : names)  // a copy that is already mapped (torch/lib/librccl.so) first
        if (!lib && (lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    if (!lib)
        for (const char* n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!lib) {
        const char* why = dlerror();  // (may be NULL)
        ZG_REQUIRE(false, ZG_ERR_UNSUPPORTED, "multi-GPU: librccl.so not found (%s)", why ? why : "no loader message");
    }
    Rccl r;
    r.lib = lib;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(lib, "ncclBroadcast"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(lib, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    ZG_REQUIRE(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Broadcast && r.AllGather && r.GetErrorString, ZG_ERR_UNSUPPORTED,
               "multi-GPU: librccl.so lacks an entry point");
    g_rccl = r;
    return ZG_OK;
}

#define ZG_NCCL(expr)                                                                                 \
    do {                                                                                              \
        const ncclResult_t r_ = (expr);                                                               \
        ZG_REQUIRE(r_ == ncclSuccess, ZG_ERR_HIP, "RCCL: %s: %s", #expr, g_rccl.GetErrorString(r_));  \
    } while (0)

}  // namespace

int dist_broadcast(void* buf, size_t bytes, int root, hipStream_t s) {
    ZG_REQUIRE(g_world > 0, ZG_ERR_ARG, "multi-GPU: zg_dist_init has not run");
    ZG_REQUIRE(root >= 0 && root < g_world, ZG_ERR_ARG, "multi-GPU: root %d of %d ranks", root, g_world);
    ZG_NCCL(g_rccl.Broadcast(buf, buf, bytes, ncclUint8, root, g_comm, s));
    return ZG_OK;
}

}  // namespace zg

using namespace zg;

extern "C" {

// Not a collective: can this process bind RCCL at all?  Ranks agree on the answer BEFORE any of them enters zg_dist_init,
// which is a rendezvous (ncclCommInitRank) the others would otherwise wait in for a rank that has already given up.
int zg_dist_available(void) { return load_rccl(); }

int zg_dist_unique_id(void* id_out, size_t id_bytes) {
    ZG_REQUIRE(id_out && id_bytes == ZG_DIST_ID_BYTES, ZG_ERR_ARG, "dist_unique_id: the id is %d bytes", ZG_DIST_ID_BYTES);
    static_assert(sizeof(ncclUniqueId) == ZG_DIST_ID_BYTES, "ncclUniqueId size");
    ZG_TRY(load_rccl());
    ncclUniqueId id;
    ZG_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return ZG_OK;
}

int zg_dist_init(const void* id, size_t id_bytes, int rank, int world_size) {
    ZG_TRY(require_init());
    ZG_REQUIRE(id && id_bytes == ZG_DIST_ID_BYTES && world_size >= 1 && rank >= 0 && rank < world_size, ZG_ERR_ARG,
               "dist_init: rank %d of %d", rank, world_size);
    ZG_REQUIRE(g_world == 0, ZG_ERR_ARG, "dist_init: a communicator exists (zg_dist_finalize first)");
    ZG_TRY(load_rccl());
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ZG_HIP(hipSetDevice(ctx().device));
    ZG_NCCL(g_rccl.CommInitRank(&g_comm, world_size, uid, rank));
    g_rank = rank;
    g_world = world_size;
    return ZG_OK;
}

int zg_dist_world(int* rank, int* world_size) {
    if (rank) *rank = g_rank;
    if (world_size) *world_size = g_world;
    return ZG_OK;
}

int zg_dist_allgather(const void* send, void* recv, size_t bytes_per_rank) {
    ZG_TRY(require_init());
    ZG_REQUIRE(g_world > 0, ZG_ERR_ARG, "dist_allgather: zg_dist_init has not run");
    ZG_REQUIRE(send && recv && is_device_ptr(send) && is_device_ptr(recv), ZG_ERR_ARG, "dist_allgather: device buffers");
    ZG_NCCL(g_rccl.AllGather(send, recv, bytes_per_rank, ncclUint8, g_comm, ctx().stream));
    ZG_HIP(hipStreamSynchronize(ctx().stream));
    return ZG_OK;
}

int zg_dist_finalize(void) {
    if (g_world == 0) return ZG_OK;
    ZG_NCCL(g_rccl.CommDestroy(g_comm));
    g_comm = nullptr;
    g_world = 0;
    g_rank = 0;
    return ZG_OK;
}

}  // extern "C"
