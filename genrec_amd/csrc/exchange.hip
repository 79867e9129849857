// The rollout's one exchange step as part of the library: an RCCL all-gather enqueued on the
// caller's (compute) stream, so that reward kernel -> gather -> advantage kernel are ordered by the
// stream alone, with no host synchronisation and no hop through another runtime's stream.
//
// RCCL is bound at run time (dlopen), not at link time: single-GPU users of the library need no
// RCCL, and a process that already has one loaded (PyTorch brings its own) keeps using that copy.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

struct UniqueId {  // ncclUniqueId: NCCL_UNIQUE_ID_BYTES opaque bytes, passed by value
    char bytes[GRX_COMM_ID_BYTES];
};
typedef int (*get_unique_id_fn)(UniqueId*);
typedef int (*comm_init_rank_fn)(void**, int, UniqueId, int);
typedef int (*comm_destroy_fn)(void*);
typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*error_string_fn)(int);

struct Rccl {
    void* handle = nullptr;
    get_unique_id_fn get_unique_id = nullptr;
    comm_init_rank_fn comm_init_rank = nullptr;
    comm_destroy_fn comm_destroy = nullptr;
    all_gather_fn all_gather = nullptr;
    error_string_fn error_string = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

int bind_rccl(const char* who) {
    std::lock_guard<std::mutex> guard(g_rccl_mutex);
    if (g_rccl.handle) return GRX_OK;
    // a copy that is already in the process first (RTLD_NOLOAD), then the usual names
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    GRX_REQUIRE(h != nullptr, GRX_E_UNSUPPORTED, "%s: librccl.so not found (%s)", who, dlerror());
    Rccl r;
    r.handle = h;
    r.get_unique_id = (get_unique_id_fn)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (comm_destroy_fn)dlsym(h, "ncclCommDestroy");
    r.all_gather = (all_gather_fn)dlsym(h, "ncclAllGather");
    r.error_string = (error_string_fn)dlsym(h, "ncclGetErrorString");
    GRX_REQUIRE(r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather, GRX_E_UNSUPPORTED,
                "%s: the RCCL library lacks an entry point", who);
    g_rccl = r;
    return GRX_OK;
}

const char* rccl_error(int code) { return g_rccl.error_string ? g_rccl.error_string(code) : "?"; }

}  // namespace

struct grx_comm {
    void* comm;
    int n_ranks, rank, device;
};

extern "C" int grx_comm_unique_id(void* id_out) {
    const char* who = "grx_comm_unique_id";
    GRX_REQUIRE(id_out != nullptr, GRX_E_INVALID, "%s: null pointer", who);
    int rc = bind_rccl(who);
    if (rc != GRX_OK) return rc;
    UniqueId id;
    memset(&id, 0, sizeof(id));
    const int e = g_rccl.get_unique_id(&id);
    GRX_REQUIRE(e == 0, GRX_E_HIP, "%s: ncclGetUniqueId failed: %s", who, rccl_error(e));
    memcpy(id_out, id.bytes, GRX_COMM_ID_BYTES);
    return GRX_OK;
}

extern "C" int grx_comm_create(const void* id, int32_t n_ranks, int32_t rank, grx_comm** out) {
    const char* who = "grx_comm_create";
    GRX_REQUIRE(id != nullptr && out != nullptr, GRX_E_INVALID, "%s: null pointer", who);
    GRX_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, GRX_E_INVALID, "%s: rank %d of %d", who, rank, n_ranks);
    int rc = bind_rccl(who);
    if (rc != GRX_OK) return rc;
    grx_comm* c = new grx_comm();
    c->comm = nullptr;
    c->n_ranks = n_ranks;
    c->rank = rank;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        grx_set_error("%s: no current device", who);
        return GRX_E_HIP;
    }
    UniqueId uid;
    memcpy(uid.bytes, id, GRX_COMM_ID_BYTES);
    const int e = g_rccl.comm_init_rank(&c->comm, n_ranks, uid, rank);
    if (e != 0) {
        delete c;
        grx_set_error("%s: ncclCommInitRank failed: %s", who, rccl_error(e));
        return GRX_E_HIP;
    }
    *out = c;
    return GRX_OK;
}

extern "C" int grx_comm_destroy(grx_comm* c) {
    if (!c) return GRX_OK;
    // gathers of this communicator may still be queued on the caller's streams: let them finish first
    (void)hipDeviceSynchronize();
    if (c->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->comm);
    delete c;
    return GRX_OK;
}

extern "C" int grx_comm_allgather(grx_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    const char* who = "grx_comm_allgather";
    GRX_REQUIRE(c != nullptr && c->comm != nullptr, GRX_E_INVALID, "%s: null communicator", who);
    if (bytes_per_rank == 0) return GRX_OK;
    GRX_REQUIRE(send != nullptr && recv != nullptr, GRX_E_INVALID, "%s: null buffer", who);
    int dev = -1;
    GRX_CHECK_HIP(hipGetDevice(&dev));
    GRX_REQUIRE(dev == c->device, GRX_E_INVALID, "%s: the communicator belongs to device %d, current device is %d", who,
                c->device, dev);
    const int e = g_rccl.all_gather(send, recv, bytes_per_rank, /*ncclInt8*/ 0, c->comm, (hipStream_t)stream);
    GRX_REQUIRE(e == 0, GRX_E_HIP, "%s: ncclAllGather failed: %s", who, rccl_error(e));
    return GRX_OK;
}
