// The data-parallel gradient exchange through RCCL called FROM THE LIBRARY (the two flat
// gradient buckets, sum all-reduce over xGMI).  The reference has no distributed code; round 2 issued these collectives through
// torch.distributed, whose c10d watchdog thread polls collective end events from another thread and aborts the process when a
// poll lands inside a stream capture (a 0.35 s sleep papered over it).  Here the library owns its communicator: ncclAllReduce is
// enqueued on the update's own stream inside fbhip_update_many_dp's capture -- one graph per rank per n steps, collectives
// included, no process-group object and no watchdog on the hot path.
//
// librccl is opened at run time (dlopen of the path the host hands over -- the copy torch bundles, so that one HIP runtime serves
// the process); nothing links against it and a single-GPU user never loads it.  The unique id travels through the caller (any
// channel: the host mirror broadcasts its 128 bytes once over its control-plane group).
#include "host.h"

#include <dlfcn.h>

namespace fbhip {
namespace host {

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x: rccl.h:40-52, 436-466)
typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclSum = 0, kNcclFloat32 = 7;

struct Api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional: the release path after a FAILURE (a peer may never arrive)
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};
Api g_api;
std::mutex g_api_mu;

int fail(fbhip_ctx* c, const std::string& what) {
    g_err = what;
    if (c) c->err = what;
    return FBHIP_E_STATE;
}

int check(fbhip_ctx* c, ncclResult_t r, const char* call) {
    if (r == 0) return FBHIP_OK;
    return fail(c, std::string("fbhip: ") + call + " failed: " + (g_api.GetErrorString ? g_api.GetErrorString(r) : "rccl error") + " (" + std::to_string(r) + ")");
}

}  // namespace

int rccl_load(const char* path) {
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.lib != nullptr) return FBHIP_OK;
    const char* cand[] = {path, "librccl.so", "librccl.so.1"};
    void* lib = nullptr;
    for (const char* p : cand)
        if (p != nullptr && p[0] != 0 && (lib = dlopen(p, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (lib == nullptr) return fail(nullptr, std::string("fbhip: cannot open librccl (") + (dlerror() ? dlerror() : "not found") + ")");
    Api a;
    a.lib = lib;
#define SYM(field, name)                                                                       \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, name));                           \
    if (a.field == nullptr) return fail(nullptr, std::string("fbhip: librccl has no ") + name);
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString") SYM(GetVersion, "ncclGetVersion")
#undef SYM
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(lib, "ncclCommAbort"));      // (absent: ncclCommDestroy is used)
    g_api = a;
    return FBHIP_OK;
}

int rccl_unique_id(void* out128) {
    if (g_api.lib == nullptr) return fail(nullptr, "fbhip: librccl not loaded (fbhip_rccl_load)");
    ncclUniqueId id;
    RC(check(nullptr, g_api.GetUniqueId(&id), "ncclGetUniqueId"));
    memcpy(out128, id.internal, sizeof(id.internal));
    return FBHIP_OK;
}

int rccl_version() {
    int v = 0;
    if (g_api.lib == nullptr || g_api.GetVersion(&v) != 0) return 0;
    return v;
}

// collective set-up (connections per algorithm / message size) cannot happen inside a capture: run both buckets' all-reduces
// once, eagerly, on scratch of the same sizes -- the warm-up iterations of any graph-capture recipe
static int rccl_warm_up(fbhip_ctx* c, hipStream_t s) {
    const int64_t n_fb = c->L[FBHIP_NET_FORWARD].numel + c->L[FBHIP_NET_BACKWARD].numel, n_ac = c->L[FBHIP_NET_ACTOR].numel;
    // the split-K slab of the bound workspace is free between updates and larger than either bucket at supported dims? not
    // guaranteed: reduce the gradient buffers themselves after zeroing them (they are rewritten by the next backward pass)
    HIPCK(c, hipMemsetAsync(c->fb_g, 0, (size_t)n_fb * sizeof(float), s));
    RC(check(c, g_api.AllReduce(c->fb_g, c->fb_g, (size_t)n_fb, kNcclFloat32, kNcclSum, (ncclComm_t)c->rccl_comm, s), "ncclAllReduce (warm-up, fb bucket)"));
    if (n_ac > 0 && c->a_g != nullptr) {
        HIPCK(c, hipMemsetAsync(c->a_g, 0, (size_t)n_ac * sizeof(float), s));
        RC(check(c, g_api.AllReduce(c->a_g, c->a_g, (size_t)n_ac, kNcclFloat32, kNcclSum, (ncclComm_t)c->rccl_comm, s), "ncclAllReduce (warm-up, actor bucket)"));
    }
    HIPCK(c, hipStreamSynchronize(s));
    return FBHIP_OK;
}

int rccl_init(fbhip_ctx* c, const void* id128, int world, int rank, hipStream_t s) {
    if (g_api.lib == nullptr) return fail(c, "fbhip: librccl not loaded (fbhip_rccl_load)");
    if (world < 1 || rank < 0 || rank >= world || id128 == nullptr) return fail(c, "fbhip_rccl_init: bad argument");
    if (c->rccl_comm != nullptr) { (void)g_api.CommDestroy((ncclComm_t)c->rccl_comm); c->rccl_comm = nullptr; }
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    ncclComm_t comm = nullptr;
    RC(check(c, g_api.CommInitRank(&comm, world, id, rank), "ncclCommInitRank"));
    c->rccl_comm = comm; c->rccl_world = world; c->rccl_rank = rank;
    const int rc = rccl_warm_up(c, s);
    if (rc != FBHIP_OK) {                      // a half-initialised communicator must not outlive the error: the update's all-reduce
        // lambda prefers rccl_comm over bound peers.  ABORT, not destroy: the warm-up collective may still be
        // outstanding because a peer failed after the rendezvous, and ncclCommDestroy would wait for it
        (void)(g_api.CommAbort ? g_api.CommAbort(comm) : g_api.CommDestroy(comm));
        c->rccl_comm = nullptr; c->rccl_world = 0; c->rccl_rank = 0;
    }
    return rc;
}

// abort = true: the release after a failure agreed by the ranks (rccl.py: a peer's init failed after the rendezvous) -- collectives
// of this communicator may be outstanding with no partner; ncclCommAbort tears down without waiting for them
void rccl_release(fbhip_ctx* c, bool abort) {
    if (c->rccl_comm != nullptr && g_api.lib != nullptr)
        (void)((abort && g_api.CommAbort) ? g_api.CommAbort((ncclComm_t)c->rccl_comm) : g_api.CommDestroy((ncclComm_t)c->rccl_comm));
    c->rccl_comm = nullptr;
}

// sum all-reduce of bucket ``which`` (0: fb gradients, 1: actor gradients) in place, on ``s`` (capturable)
int rccl_allreduce(fbhip_ctx* c, int which, hipStream_t s) {
    if (c->rccl_comm == nullptr) return fail(c, "fbhip: no RCCL communicator bound (fbhip_rccl_init)");
    const int64_t n = which == 0 ? c->L[FBHIP_NET_FORWARD].numel + c->L[FBHIP_NET_BACKWARD].numel : c->L[FBHIP_NET_ACTOR].numel;
    float* buf = which == 0 ? c->fb_g : c->a_g;
    if (n <= 0 || buf == nullptr) return FBHIP_OK;
    return check(c, g_api.AllReduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, (ncclComm_t)c->rccl_comm, s), "ncclAllReduce");
}

}  // namespace host
}  // namespace fbhip
