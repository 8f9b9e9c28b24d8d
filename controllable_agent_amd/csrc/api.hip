// C ABI of libfbhip.so (include/fbhip.h): context and binding, the update entry points (hipGraph capture / replay of the
// schedules of schedule.hip), the data-parallel and inference entry points, and the per-kernel test exports.
//
// Memory model: torch owns every byte.  The library computes offsets (fbhip_layout_*, fbhip_workspace_bytes), the host
// allocates flat fp32 tensors and binds them; after that an update is a few dozen asynchronous kernel launches on the caller's
// stream with no allocation and no host synchronisation, captured once into a hipGraph and replayed.
#include "host.h"
#include <chrono>

using namespace fbhip;
using namespace fbhip::host;

// ---- launching graphs with parallel branches ----------------------------------------------------------------------------
// hipGraphLaunch of an exec whose graph has n > 1 branches can walk off the end of the exec's parallel-stream list and crash
// the process (ROCm 7.0's libamdhip64 as bundled with torch 2.10: hip::Graph::UpdateStreams).  At instantiate time the runtime
// creates n extra NORMAL-priority streams; at every launch it hands them to the branches, SKIPPING each extra stream that shares
// its hardware queue with the launch stream -- without a bound on the index.  One such collision is provided for; two (both
// extra streams of a two-branch graph on the launch stream's queue) are not.  Streams take the least-used of the 4 hardware
// queues of their priority class, so it depends on the process's history of stream creations / destructions: a process that
// builds and drops many contexts hit it in 3 of 6 suite runs in round 2, one in five in round 3 (backtrace:
// profiles/r03_segv_backtrace.txt); tools/graph_queue_collision.hip reproduces it in ~40 trials of random stream churn.
// The hardware queue of a HIGH-priority stream comes from another pool and can never be shared with those extra streams (same
// tool: 3000 of 3000 trials): graphs with parallel branches are therefore launched from a high-priority stream of the
// library's own, ordered behind and ahead of the caller's stream with two events (no host synchronisation).
namespace {

constexpr int MAX_DEVICES = 64;
std::mutex g_launch_mu;
hipStream_t g_launch_stream[MAX_DEVICES] = {};    // one per device (of the CURRENT device of the calling thread), never destroyed

hipError_t launch_stream(hipStream_t* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_launch_mu);
    if (g_launch_stream[dev] == nullptr) {
        int lo = 0, hi = 0;
        e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e != hipSuccess) return e;
        if (hi >= lo) return hipErrorNotSupported;     // no high-priority class: the guard above cannot be given
        e = hipStreamCreateWithPriority(&g_launch_stream[dev], hipStreamNonBlocking, hi);
        if (e != hipSuccess) return e;
    }
    *out = g_launch_stream[dev];
    return hipSuccess;
}

// May this process replay graphs with parallel branches at all?  The guard above rests on two properties of the HIP runtime that no
// public API exposes (hardware-queue identity is runtime-internal: tools/stream_queue_probe.hip reads it through object offsets of
// one libamdhip64 build, which a product must not do): (1) a high-priority stream's hardware queue never coincides with the
// queues of the normal-priority streams an exec creates -- verified for the HIP 7.0 runtime torch 2.10 bundles
// (tools/graph_queue_collision.hip: 3000 of 3000 trials) --, or (2) the runtime does not have the unbounded skip loop at all --
// verified for HIP 7.2.  So: branched graphs on HIP 7.0.x (with the guard) and on HIP >= 7.2, provided the device has a
// high-priority class; on any other runtime the library REFUSES them and builds its n-step graphs single-queue (the plain form:
// same kernels and results, 3-6 % slower) instead of silently running the configuration that crashed.  FBHIP_BRANCHED_GRAPHS=1
// forces them on an unlisted runtime, =0 refuses them everywhere (read at every call: tests force the fallback with it).
const char* branched_graphs_verdict(bool* ok) {
    static const char* cached = nullptr;
    static bool cached_ok = false;
    static std::once_flag once;
    std::call_once(once, [] {
        int v = 0, lo = 0, hi = 0;
        if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); cached = "refused: hipRuntimeGetVersion failed"; return; }
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hi >= lo) { (void)hipGetLastError(); cached = "refused: the device has no high-priority stream class"; return; }
        const int major = v / 10000000, minor = (v / 100000) % 100;
        if (major == 7 && minor == 0) { cached_ok = true; cached = "allowed: HIP 7.0 runtime, launched from a high-priority stream (verified workaround)"; }
        else if (major > 7 || (major == 7 && minor >= 2)) { cached_ok = true; cached = "allowed: HIP >= 7.2 runtime (no unbounded skip in hip::Graph::UpdateStreams)"; }
        else cached = "refused: unverified HIP runtime version for graphs with parallel branches (FBHIP_BRANCHED_GRAPHS=1 forces them)";
    });
    const char* e = getenv("FBHIP_BRANCHED_GRAPHS");
    if (e != nullptr && e[0] == '0') { *ok = false; return "refused: FBHIP_BRANCHED_GRAPHS=0"; }
    if (e != nullptr && e[0] == '1' && !cached_ok) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo) { *ok = true; return "allowed: forced by FBHIP_BRANCHED_GRAPHS=1"; }
        (void)hipGetLastError();
    }
    *ok = cached_ok;
    return cached;
}

int launch_graph(fbhip_ctx* c, hipGraphExec_t exec, hipStream_t s, bool branches) {
    if (!branches) {
        HIPCK(c, hipGraphLaunch(exec, s));
        return FBHIP_OK;
    }
    {   // the caller's stream is itself of the high-priority class: its hardware queue comes from the other pool already -- launch
        // there, without the two event hops (measured round 4, HISTORY.md: per-update launches of a branched
        // graph run at 1108 update-steps/s this way, at 513 through the hop under the runtime's default dependency handling)
        int prio = 0, lo = 0, hi = 0;
        if (s != nullptr && hipStreamGetPriority(s, &prio) == hipSuccess && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo && prio == hi) {
            HIPCK(c, hipGraphLaunch(exec, s));
            return FBHIP_OK;
        }
        (void)hipGetLastError();
    }
    hipStream_t ls = nullptr;
    HIPCK(c, launch_stream(&ls));
    if (!c->ev_in) HIPCK(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    if (!c->ev_out) HIPCK(c, hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    HIPCK(c, hipEventRecord(c->ev_in, s));
    HIPCK(c, hipStreamWaitEvent(ls, c->ev_in, 0));
    HIPCK(c, hipGraphLaunch(exec, ls));
    HIPCK(c, hipEventRecord(c->ev_out, ls));
    HIPCK(c, hipStreamWaitEvent(s, c->ev_out, 0));
    return FBHIP_OK;
}

// The legacy default stream and work on another stream.  A caller that sits on the legacy (null) stream expects the entry point's
// work to come after its EARLIER null-stream work and its LATER null-stream work to see what the entry point enqueued elsewhere.
// Events recorded on / waited for by the null stream do that, but leave a command pending on the null stream while the n-step
// graph is enqueued and runs -- and with one there the branched graphs of this library ran 1.5x slower on MI355X / ROCm 7.0
// (measured round 3: 650 vs 970 SF update-steps/s, 700 vs 1117 FB; pending work on a NON-blocking stream costs nothing).  So
// neither direction touches the null stream: both go through per-device BLOCKING helper streams and the runtime's own
// legacy-stream rule (a blocking stream's command waits for earlier null-stream work; a null-stream command waits for every
// blocking stream's earlier work), which applies at the moment the caller really uses the null stream.
std::mutex g_gate_mu;
hipStream_t g_gate_in[MAX_DEVICES] = {}, g_gate_out[MAX_DEVICES] = {};      // per device: the legacy-stream rule is per device

int gate_streams(fbhip_ctx* c, hipStream_t* in, hipStream_t* out) {
    int dev = 0;
    HIPCK(c, hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) { c->err = g_err = "fbhip: device ordinal out of range"; return FBHIP_E_INVALID; }
    std::lock_guard<std::mutex> lk(g_gate_mu);
    if (g_gate_in[dev] == nullptr) HIPCK(c, hipStreamCreateWithFlags(&g_gate_in[dev], hipStreamDefault));
    if (g_gate_out[dev] == nullptr) HIPCK(c, hipStreamCreateWithFlags(&g_gate_out[dev], hipStreamDefault));
    *in = g_gate_in[dev]; *out = g_gate_out[dev];
    return FBHIP_OK;
}

// later legacy-stream work after everything enqueued on s so far
int order_legacy_after(fbhip_ctx* c, hipStream_t s) {
    if (s == nullptr) return FBHIP_OK;                 // already on the legacy stream
    hipStream_t gin = nullptr, gout = nullptr;
    RC(gate_streams(c, &gin, &gout));
    if (!c->ev_gate) HIPCK(c, hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming));
    HIPCK(c, hipEventRecord(c->ev_gate, s));
    HIPCK(c, hipStreamWaitEvent(gout, c->ev_gate, 0));
    return FBHIP_OK;
}

// later work on s after everything enqueued on the legacy stream so far
int order_after_legacy(fbhip_ctx* c, hipStream_t s) {
    if (s == nullptr) return FBHIP_OK;
    hipStream_t gin = nullptr, gout = nullptr;
    RC(gate_streams(c, &gin, &gout));
    if (!c->ev_gate_in) HIPCK(c, hipEventCreateWithFlags(&c->ev_gate_in, hipEventDisableTiming));
    HIPCK(c, hipEventRecord(c->ev_gate_in, gin));           // a blocking stream's marker: behind the null stream's earlier commands
    HIPCK(c, hipStreamWaitEvent(s, c->ev_gate_in, 0));
    return FBHIP_OK;
}

// contexts whose destruction was asked for while a stream capture was open on their stream (hipGraphExecDestroy / hipHostFree /
// hipDeviceSynchronize are illegal there): destroyed at the next entry point that is not inside a capture
std::mutex g_reap_mu;
std::vector<fbhip_ctx*> g_reap;

void destroy_now(fbhip_ctx* ctx) {
    (void)hipDeviceSynchronize();
    rccl_release(ctx);
    for (auto& g : ctx->graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto& g : ctx->infer_graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto e : ctx->events) (void)hipEventDestroy(e);
    if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
    if (ctx->ev_out) (void)hipEventDestroy(ctx->ev_out);
    if (ctx->ev_gate) (void)hipEventDestroy(ctx->ev_gate);
    if (ctx->ev_gate_in) (void)hipEventDestroy(ctx->ev_gate_in);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    if (ctx->h_metrics) (void)hipHostFree(ctx->h_metrics);
    if (ctx->d_pubseq) (void)hipFree(ctx->d_pubseq);
    if (ctx->d_xm_part) (void)hipFree(ctx->d_xm_part);
    delete ctx;
}

bool capture_open(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// Every live context, for "is ANY stream this library has been handed capturing right now?": a caller on the legacy stream gives
// each agent a stream of its own, so the context being destroyed (garbage-collected inside ANOTHER agent's capture, say) knows
// nothing about the stream the open capture sits on.
std::mutex g_live_mu;
std::vector<fbhip_ctx*> g_live;

bool any_capture_open(hipStream_t also) {
    if (also != nullptr && capture_open(also)) return true;
    std::vector<hipStream_t> streams;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        for (fbhip_ctx* c : g_live)
            if (c->last_stream != nullptr) streams.push_back(c->last_stream);
    }
    for (hipStream_t s : streams)
        if (capture_open(s)) return true;
    return false;
}

void reap(hipStream_t s) {
    if (any_capture_open(s)) return;
    std::vector<fbhip_ctx*> dead;
    {
        std::lock_guard<std::mutex> lk(g_reap_mu);
        dead.swap(g_reap);
    }
    for (fbhip_ctx* d : dead) destroy_now(d);
}

}  // namespace

// =================================================================================================== C ABI
extern "C" {

int fbhip_abi_version(void) { return FBHIP_ABI_VERSION; }

const char* fbhip_last_error(const fbhip_ctx* ctx) { return (ctx && !ctx->err.empty()) ? ctx->err.c_str() : g_err.c_str(); }

int fbhip_branched_graphs(const char** why) {
    bool ok = false;
    const char* w = branched_graphs_verdict(&ok);
    if (why != nullptr) *why = w;
    return ok ? 1 : 0;
}

int fbhip_device_ok(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        g_err = "fbhip: no HIP device";
        return FBHIP_E_NODEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_err = std::string("fbhip: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return FBHIP_E_NODEVICE;
    }
    return FBHIP_OK;
}

int64_t fbhip_net_numel(const fbhip_dims* dims, int net) {
    if (check_dims(dims) != FBHIP_OK || net < 0 || net > 2) return FBHIP_E_INVALID;
    return build_layout(*dims, net).numel;
}
int64_t fbhip_net_param_count(const fbhip_dims* dims, int net) {
    if (check_dims(dims) != FBHIP_OK || net < 0 || net > 2) return FBHIP_E_INVALID;
    return build_layout(*dims, net).nparams;
}
int fbhip_layout_count(const fbhip_dims* dims, int net) {
    if (check_dims(dims) != FBHIP_OK || net < 0 || net > 2) return FBHIP_E_INVALID;
    return (int)build_layout(*dims, net).slots.size();
}
int fbhip_layout_entry(const fbhip_dims* dims, int net, int idx, fbhip_tensor_desc* out) {
    if (check_dims(dims) != FBHIP_OK || net < 0 || net > 2 || !out) return FBHIP_E_INVALID;
    const NetLayout L = build_layout(*dims, net);
    if (idx < 0 || idx >= (int)L.slots.size()) { g_err = "fbhip: layout index out of range"; return FBHIP_E_INVALID; }
    const Slot& s = L.slots[idx];
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", s.name.c_str());
    out->offset = s.off; out->rows = s.rows; out->cols = s.cols; out->ld = s.ld;
    return FBHIP_OK;
}
size_t fbhip_workspace_bytes(const fbhip_dims* dims) {
    if (check_dims(dims) != FBHIP_OK) return 0;
    // two complete sets (fbhip_update_many pipelines consecutive steps)
    return 2 * carve(*dims, nullptr).total_bytes;
}

int fbhip_create(const fbhip_dims* dims, fbhip_ctx** out) {
    if (!out) return FBHIP_E_INVALID;
    RC(check_dims(dims));
    fbhip_ctx* c = new fbhip_ctx();
    c->d = *dims;
    c->sq.on = dims->boltzmann ? 1 : 0;
    for (int n = 0; n < 3; ++n) c->L[n] = build_layout(*dims, n);
    // pinned staging of the batch-1 entry points (a few hundred bytes; host memory, not device memory)
    if (hipHostMalloc((void**)&c->h_in, act_in_floats(*dims) * sizeof(float), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_out, 64 * sizeof(float), hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();             // no usable device here (e.g. the CPU-only build check): fast path disabled
        c->h_in = c->h_out = nullptr;
    } else {
        memset(c->h_in, 0, act_in_floats(*dims) * sizeof(float));
        memset(c->h_out, 0, 64 * sizeof(float));
        // the metrics' way out of a running step (metrics_publish_kernel): fine-grained pinned host memory + a 4-byte device counter.
        // Without them (allocation refused) metrics travel by copy + synchronise as before.
        if (hipHostMalloc((void**)&c->h_metrics, 2 * FBHIP_NUM_METRICS * sizeof(float), hipHostMallocCoherent) != hipSuccess ||
            hipMalloc((void**)&c->d_pubseq, 16 * sizeof(unsigned int)) != hipSuccess || hipMemset(c->d_pubseq, 0, 16 * sizeof(unsigned int)) != hipSuccess ||
            hipMalloc((void**)&c->d_xm_part, EXTRA_METRICS_MAX_BLOCKS * 4 * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            if (c->h_metrics) (void)hipHostFree(c->h_metrics);
            if (c->d_pubseq) (void)hipFree(c->d_pubseq);
            if (c->d_xm_part) (void)hipFree(c->d_xm_part);
            c->h_metrics = nullptr; c->d_pubseq = nullptr; c->d_xm_part = nullptr;
        } else {
            memset(c->h_metrics, 0, 2 * FBHIP_NUM_METRICS * sizeof(float));
            // the same way out for the batch-1 results (act / compute_z_correl): FBHIP_INFER_DIRECT=0 keeps copy node + synchronise
            const char* e = getenv("FBHIP_INFER_DIRECT");
            c->infer_direct = !(e && e[0] == '0');
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.push_back(c);
    }
    reap(nullptr);                               // contexts parked by a destroy inside a capture die at the next entry point outside one
    *out = c;
    return FBHIP_OK;
}

int fbhip_destroy(fbhip_ctx* ctx) {
    if (!ctx) return FBHIP_OK;
    // The context's launches are asynchronous: its graphs, events and side stream (and, on the caller's side, the buffers it was
    // bound to) may still be in use by work in flight: destroy_now drains the device first.  None of that is legal while a stream
    // capture is open on ANY stream a live context was last called on (a caller's torch-level capture around agent calls, with a
    // garbage-collected agent's __del__ landing inside it): then the context goes to the reaper and dies at the next entry point
    // outside a capture.
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.erase(std::remove(g_live.begin(), g_live.end(), ctx), g_live.end());
    }
    if (any_capture_open(ctx->last_stream)) {
        std::lock_guard<std::mutex> lk(g_reap_mu);
        g_reap.push_back(ctx);
        return FBHIP_OK;
    }
    reap(ctx->last_stream);
    destroy_now(ctx);
    return FBHIP_OK;
}

int fbhip_bind_buffers(fbhip_ctx* c, float* fb_params, float* fb_grads, float* fb_adam_m, float* fb_adam_v,
                       float* fb_targets, float* actor_params, float* actor_grads, float* actor_adam_m,
                       float* actor_adam_v, void* workspace, size_t workspace_bytes) {
    if (!c) return FBHIP_E_INVALID;
    const bool has_actor = c->d.discrete == 0;      // (DiscreteFBAgent: no actor, its four pointers are ignored)
    if (!fb_params || !fb_grads || !fb_adam_m || !fb_adam_v || !fb_targets || !workspace ||
        (has_actor && (!actor_params || !actor_grads || !actor_adam_m || !actor_adam_v))) { c->err = g_err = "fbhip: null buffer"; return FBHIP_E_INVALID; }
    const size_t one = carve(c->d, nullptr).total_bytes, need = 2 * one;
    if (workspace_bytes < need) { c->err = g_err = "fbhip: workspace too small (fbhip_workspace_bytes)"; return FBHIP_E_INVALID; }
    if (((uintptr_t)workspace & 255) || ((uintptr_t)fb_params & 15) || ((uintptr_t)fb_grads & 15) ||
        ((uintptr_t)fb_targets & 15) || (has_actor && (((uintptr_t)actor_params & 15) || ((uintptr_t)actor_grads & 15)))) {
        c->err = g_err = "fbhip: buffers must be 16-byte aligned (workspace 256)";
        return FBHIP_E_INVALID;
    }
    RC(fbhip_device_ok());
    c->fb_p = fb_params; c->fb_g = fb_grads; c->fb_m = fb_adam_m; c->fb_v = fb_adam_v; c->fb_t = fb_targets;
    c->a_p = actor_params; c->a_g = actor_grads; c->a_m = actor_adam_m; c->a_v = actor_adam_v;
    c->sets[0] = carve(c->d, workspace);
    c->sets[1] = carve(c->d, (char*)workspace + one);
    c->cur = 0;
    // state that is not per-step lives once: optimiser / RNG counters, metrics, the batch-1 staging buffers
    c->sets[1].st = c->sets[0].st; c->sets[1].metrics = c->sets[0].metrics;
    c->sets[1].act_in = c->sets[0].act_in; c->sets[1].act_vec = c->sets[0].act_vec; c->sets[1].act_out = c->sets[0].act_out;
    c->ws_lo = (const char*)workspace; c->ws_bytes = need;
    const int64_t nf = c->L[FBHIP_NET_FORWARD].numel;
    c->F_p = fwd_p(fb_params, c->L[0]); c->F_g = fwd_p(fb_grads, c->L[0]); c->F_t = fwd_p(fb_targets, c->L[0]);
    c->K_p = bwd_p(fb_params + nf, c->L[1]); c->K_g = bwd_p(fb_grads + nf, c->L[1]); c->K_t = bwd_p(fb_targets + nf, c->L[1]);
    c->I_p = icm_p(fb_params + nf, c->L[1]); c->I_g = icm_p(fb_grads + nf, c->L[1]);
    c->M_p = mu_p(fb_params + nf, c->L[1]); c->M_g = mu_p(fb_grads + nf, c->L[1]); c->M_t = mu_p(fb_targets + nf, c->L[1]);
    if (has_actor) { c->A_p = act_p(actor_params, c->L[2]); c->A_g = act_p(actor_grads, c->L[2]); }
    for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
    c->graphs.clear();
    for (auto& g : c->infer_graphs) (void)hipGraphExecDestroy(g.exec);
    c->infer_graphs.clear();
    HIPCK(c, pairwise_prepare(c->d.batch, c->d.z_dim));
    HIPCK(c, gemm_init());
    HIPCK(c, inverse_prepare());
    if (has_actor) HIPCK(c, actor_head_bwd_prepare(c->d.hidden_dim, c->d.action_dim));
    if (has_actor) HIPCK(c, policy_head_prepare(c->d.hidden_dim, c->d.action_dim, head_width(c->d)));
    c->bound = true;
    return FBHIP_OK;
}

int fbhip_replay_bind(fbhip_ctx* c, const float* observation, const float* action, const float* discount,
                      const float* goal, const int32_t* episode_len, const int64_t* cum_len, int32_t n_episodes,
                      int32_t t1, int32_t fixed_length) {
    if (!c) return FBHIP_E_INVALID;
    if (!observation || !action || !discount || !episode_len || n_episodes < 1 || t1 < 2) { c->err = g_err = "fbhip: bad replay storage"; return FBHIP_E_INVALID; }
    if (c->d.use_goal && !goal) { c->err = g_err = "fbhip: goal storage required when use_goal"; return FBHIP_E_INVALID; }
    if (!fixed_length && !cum_len) { c->err = g_err = "fbhip: cum_len required for variable-length episodes"; return FBHIP_E_INVALID; }
    c->rv.observation = observation; c->rv.action = action; c->rv.discount = discount; c->rv.goal = goal;
    c->rv.episode_len = episode_len; c->rv.cum_len = cum_len; c->rv.n_episodes = n_episodes; c->rv.t1 = t1;
    c->rv.fixed_length = fixed_length;
    for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
    c->graphs.clear();
    c->replay_bound = true;
    return FBHIP_OK;
}

int fbhip_set_policy_squash(fbhip_ctx* c, float temp, float log_std_min, float log_std_max) {
    if (!c) return FBHIP_E_INVALID;
    if (!c->d.boltzmann) { c->err = g_err = "fbhip_set_policy_squash: the context was not created with boltzmann"; return FBHIP_E_STATE; }
    if (!(log_std_min < log_std_max)) { c->err = g_err = "fbhip_set_policy_squash: empty log_std interval"; return FBHIP_E_INVALID; }
    c->sq = Squash{1, temp, log_std_min, log_std_max};
    for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);        // the values are baked into captured launches
    c->graphs.clear();
    for (auto& g : c->infer_graphs) (void)hipGraphExecDestroy(g.exec);
    c->infer_graphs.clear();
    return FBHIP_OK;
}

size_t fbhip_embeddings_floats(const fbhip_dims* d) {
    if (check_dims(d) != FBHIP_OK) return 0;
    return (size_t)6 * d->batch * pad4(d->z_dim) + (size_t)d->batch;
}

int fbhip_export_embeddings(fbhip_ctx* c, float* out, void* stream) {
    RC(need_bound(c, false));
    if (!out) return FBHIP_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const fbhip_dims& d = c->d;
    Ws& w = c->W();
    const size_t ps = (size_t)d.batch * pad4(d.z_dim);
    const float* src[6] = {w.fsO.F1.p, w.fsO.F2.p, d.norm_z ? w.bsO.Bm.p : w.bsO.y.p,
                           w.fsT.F1.p, w.fsT.F2.p, d.norm_z ? w.bsA.Bm.p : w.bsA.y.p};
    for (int m = 0; m < 6; ++m)
        HIPCK(c, hipMemcpyAsync(out + m * ps, src[m], ps * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCK(c, hipMemcpyAsync(out + 6 * ps, w.disc, (size_t)d.batch * sizeof(float), hipMemcpyDeviceToDevice, s));
    return FBHIP_OK;
}

int fbhip_bind_global_batch(fbhip_ctx* c, const float* panels, const float* discount, int32_t global_rows,
                            int32_t row_offset) {
    if (!c) return FBHIP_E_INVALID;
    if (global_rows == 0) { panels = discount = nullptr; row_offset = 0; }
    else if (!panels || !discount || global_rows < c->d.batch || row_offset < 0 || row_offset + c->d.batch > global_rows ||
             (global_rows != c->d.batch && ((c->d.batch & 31) || (row_offset & 31))) || ((uintptr_t)panels & 15)) {
        c->err = g_err = "fbhip_bind_global_batch: need 16-byte aligned panels and batch / row_offset multiples of 32 inside global_rows";
        return FBHIP_E_INVALID;
    }
    if (panels != c->gb_panels || discount != c->gb_discount || global_rows != c->gb_rows || row_offset != c->gb_off) {
        for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);        // the pointers are baked into captured launches
        c->graphs.clear();
    }
    c->gb_panels = panels; c->gb_discount = discount; c->gb_rows = global_rows; c->gb_off = row_offset;
    return FBHIP_OK;
}

int fbhip_set_seed(fbhip_ctx* c, uint64_t seed, uint32_t rank) {
    if (!c) return FBHIP_E_INVALID;
    c->seed = seed; c->rank = rank;
    for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
    c->graphs.clear();
    for (auto& g : c->infer_graphs) (void)hipGraphExecDestroy(g.exec);
    c->infer_graphs.clear();
    return FBHIP_OK;
}

int fbhip_set_step_counts(fbhip_ctx* c, int32_t fb_steps, int32_t actor_steps, void* stream) {
    RC(need_bound(c, false));
    hipStream_t s = (hipStream_t)stream;
    StepState h{};
    HIPCK(c, hipMemcpyAsync(&h, c->W().st, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCK(c, hipStreamSynchronize(s));
    h.fb_t = fb_steps; h.actor_t = actor_steps;
    HIPCK(c, hipMemcpyAsync(c->W().st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    HIPCK(c, hipStreamSynchronize(s));
    return FBHIP_OK;
}

int fbhip_get_step_counts(fbhip_ctx* c, int32_t* host_fb_steps, int32_t* host_actor_steps, void* stream) {
    RC(need_bound(c, false));
    hipStream_t s = (hipStream_t)stream;
    StepState h{};
    HIPCK(c, hipMemcpyAsync(&h, c->W().st, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCK(c, hipStreamSynchronize(s));
    if (host_fb_steps) *host_fb_steps = h.fb_t;
    if (host_actor_steps) *host_actor_steps = h.actor_t;
    return FBHIP_OK;
}

int fbhip_get_rng_counts(fbhip_ctx* c, uint32_t* host_update_count, uint32_t* host_act_count, void* stream) {
    RC(need_bound(c, false));
    hipStream_t s = (hipStream_t)stream;
    StepState h{};
    HIPCK(c, hipMemcpyAsync(&h, c->W().st, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCK(c, hipStreamSynchronize(s));
    if (host_update_count) *host_update_count = h.update_count;
    if (host_act_count) *host_act_count = h.act_count;
    return FBHIP_OK;
}

int fbhip_set_rng_counts(fbhip_ctx* c, uint32_t update_count, uint32_t act_count, void* stream) {
    RC(need_bound(c, false));
    hipStream_t s = (hipStream_t)stream;
    StepState h{};
    HIPCK(c, hipMemcpyAsync(&h, c->W().st, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCK(c, hipStreamSynchronize(s));
    h.update_count = update_count; h.act_count = act_count;
    HIPCK(c, hipMemcpyAsync(c->W().st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    HIPCK(c, hipStreamSynchronize(s));
    return FBHIP_OK;
}

// how many metrics_publish_kernel launches a call with this phase mask enqueues (schedule.hip: after the actor loss; after the FB
// metrics for an agent without an actor)
static unsigned int publishes(const fbhip_ctx* c, const fbhip_hparams* hp, int mask) {
    if (!hp->want_metrics || c->h_metrics == nullptr) return 0;
    return (mask & (c->d.discrete ? FBHIP_PHASE_FB_BWD_A : FBHIP_PHASE_ACTOR_GRAD)) ? 1u : 0u;
}

int fbhip_update(fbhip_ctx* c, const fbhip_hparams* hp, const fbhip_inject* inject, int32_t phase_mask,
                 int32_t use_graph, void* stream) {
    RC(need_bound(c, (phase_mask & FBHIP_PHASE_SAMPLE) != 0));
    RC(check_hparams(c, hp));
    hipStream_t s = (hipStream_t)stream;
    c->last_stream = s;
    if (!use_graph) {
        RC(enqueue_update(c, *hp, inject, phase_mask, s));
        c->pub_issued += publishes(c, hp, phase_mask);
        return FBHIP_OK;
    }
    reap(s);
    for (auto& g : c->graphs) {
        if (g.n_steps == 1 && g.set == c->cur && g.mask == phase_mask && memcmp(&g.hp, hp, sizeof(*hp)) == 0 && g.has_inj == (inject != nullptr) &&
            (!inject || memcmp(&g.inj, inject, sizeof(*inject)) == 0)) {
            HIPCK(c, hipGraphLaunch(g.exec, s));
            c->pub_issued += publishes(c, hp, phase_mask);
            return FBHIP_OK;
        }
    }
    hipGraph_t graph = nullptr;
    HIPCK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_update(c, *hp, inject, phase_mask, s);
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != FBHIP_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    HIPCK(c, e);
    GraphEntry ge{};
    ge.mask = phase_mask; ge.hp = *hp; ge.has_inj = inject != nullptr; ge.n_steps = 1; ge.set = c->cur;
    if (inject) ge.inj = *inject;
    e = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIPCK(c, e);
    ++c->graph_captures;
    if (c->graphs.size() >= 16) { (void)hipGraphExecDestroy(c->graphs.front().exec); c->graphs.erase(c->graphs.begin()); }
    c->graphs.push_back(ge);
    HIPCK(c, hipGraphLaunch(ge.exec, s));
    c->pub_issued += publishes(c, hp, phase_mask);
    return FBHIP_OK;
}

int fbhip_fb_early_grad_range(const fbhip_dims* dims, int64_t* offset, int64_t* count) {
    if (check_dims(dims) != FBHIP_OK || !offset || !count) return FBHIP_E_INVALID;
    const NetLayout L = build_layout(*dims, FBHIP_NET_FORWARD);
    // the heads' hidden layers (F1.0 F2.0, weights then biases: 2H x feat + 2H, 98 % of the heads' parameters) are laid out
    // contiguously before the output layers F1.2 F2.2, whose thin weight gradients are produced in the LAST backward round
    *offset = (int64_t)L.by_name.at("F1.0.weight").off;
    *count = (int64_t)L.by_name.at("F1.2.weight").off - *offset;
    return FBHIP_OK;
}

int fbhip_select_workspace_set(fbhip_ctx* c, int32_t which) {
    if (!c || which < 0 || which > 1) return FBHIP_E_INVALID;
    c->cur = which;
    return FBHIP_OK;
}

constexpr int DP_GRAPH_BIT = 1 << 20;         // graph-cache key: the data-parallel variant of an n-step graph
// injs: NULL (device-drawn batches) or n_steps inject structs, one per step (parity runs through the pipelined graph)
static int update_many_impl(fbhip_ctx* c, const fbhip_hparams* hp, int32_t n_steps, const fbhip_inject* injs, void* stream,
                            bool dp = false, bool launch = true) {
    RC(need_bound(c, true));
    RC(check_hparams(c, hp));
    if (n_steps < 1 || n_steps > 64) { c->err = g_err = "fbhip_update_many: bad argument"; return FBHIP_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    c->last_stream = s;
    reap(s);
    // The single-rank graph pipelines consecutive steps on a second capture branch (FBHIP_UPDATE_PIPELINE=0, read at every call:
    // the plain form, bit-identical to single updates).  The DATA-PARALLEL graph is single-queue by construction: one stream,
    // every step's phases and the two all-reduces in program order.  Its branched form (round 3) ran at 471 vs 1112
    // update-steps/s depending on what else lived in the process -- cross-queue dependencies inside a replayed graph are
    // resolved by this runtime in a way a library cannot control -- for 3-6 % when it ran well.
    const char* pe = getenv("FBHIP_UPDATE_PIPELINE");
    bool branched_ok = false;
    (void)branched_graphs_verdict(&branched_ok);
    // Round 6: the data-parallel graph takes the SAME fork when branched graphs are usable (they are launched from a high-priority
    // stream, launch_graph: the instability of round 3's branched data-parallel graph was the launch queue, found in round 4):
    // step t+1's head on the side branch beside step t's actor phase, its actor all-reduce and its actor step -- the actor bucket
    // (8.9 MB) travels while the next head computes, and the per-rank rate before communication is the pipelined single-GPU rate.
    // FBHIP_DP_PIPELINE=0 (read at every call) keeps the single-queue chain: the fallback every rank can agree on.
    const char* dpe = getenv("FBHIP_DP_PIPELINE");
    const bool pipe_dp = dp && branched_ok && !(dpe && dpe[0] == '0') && !(pe && pe[0] == '0') && n_steps > 1 && !c->d.discrete;
    const bool pipe = (!dp && branched_ok && !(pe && pe[0] == '0') && n_steps > 1 && !c->d.discrete) || pipe_dp;     // (discrete: no actor phase to overlap with)
    for (auto& g : c->graphs) {
        if (g.n_steps == n_steps && g.set == c->cur && g.mask == (FBHIP_PHASE_ALL | (dp ? DP_GRAPH_BIT : 0)) && g.has_inj == (injs != nullptr) &&
            g.branches == pipe &&
            (!injs || (g.injs.size() == (size_t)n_steps && memcmp(g.injs.data(), injs, sizeof(*injs) * n_steps) == 0)) && memcmp(&g.hp, hp, sizeof(*hp)) == 0)
        {
            if (!launch) return FBHIP_OK;
            RC(launch_graph(c, g.exec, s, g.branches));
            c->pub_issued += (unsigned int)n_steps * publishes(c, hp, FBHIP_PHASE_ALL);
            return FBHIP_OK;
        }
    }
    // Software pipeline over the steps.  Step t's actor phase is ONE dependency chain of ~20 small launches; step t+1's
    // sampling, z mixing, B passes and online ForwardMap pass depend on step t only through its FB optimiser step (new
    // forward_net / backward_net / targets), which precedes the actor phase.  So they are captured as a second branch
    // beside the actor phase (own stream, own workspace set) and rejoin before step t+1's target chain, which needs the
    // new actor.  Kernels, their order inside each step and every operand are unchanged.  Results are bit-identical to
    // n_steps single updates whenever the regrouped launches keep every GEMM's K-slicing (small dims); at walker dims a few
    // small-output GEMMs of the target chain are sliced differently once they no longer share a launch with the online
    // chain: fp32 summation order, not math (tests/test_update_parity_gpu.py covers both).  Measured (walker, 8 steps per launch): 951 -> 965
    // updates/s.  (Zipping the two programs round by round into the SAME launches instead was slower, 933/s: one tile
    // configuration per launch makes the thin GEMMs of one program stragglers of the other's fat ones.)
    if (pipe) {
        if (!c->side) HIPCK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        while ((int)c->events.size() < 3 * 64) {
            hipEvent_t ev;
            HIPCK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            c->events.push_back(ev);
        }
    }
    hipGraph_t graph = nullptr;
    HIPCK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = FBHIP_OK;
    hipError_t he = hipSuccess;
    const bool has_actor = !c->d.discrete;
    const int64_t n_fb = c->L[FBHIP_NET_FORWARD].numel + c->L[FBHIP_NET_BACKWARD].numel, n_ac = c->L[FBHIP_NET_ACTOR].numel;
    auto allreduce = [&](int which) -> int {                     // every rank's gradients, summed in place
        if (c->rccl_comm != nullptr) return rccl_allreduce(c, which, s);          // ncclAllReduce on the capture stream (rccl.hip)
        HIPCK(c, launch_peer_allreduce(c->peers, which, which == 0 ? n_fb : n_ac, s));   // peer-access kernels (peer.hip)
        return (int)FBHIP_OK;
    };
    if (pipe_dp) {
        const int HEAD = FBHIP_PHASE_SAMPLE | FBHIP_PHASE_FB_FWD_ONLINE;
        const int GRAD = FBHIP_PHASE_FB_FWD_TARGET | FBHIP_PHASE_FB_BWD | FBHIP_PHASE_ACTOR_FWD;
        const int cur0 = c->cur;
        rc = enqueue_update(c, *hp, nullptr, HEAD, s);
        for (int i = 0; i < n_steps && rc == FBHIP_OK && he == hipSuccess; ++i) {
            rc = enqueue_update(c, *hp, nullptr, GRAD, s);
            if (rc == FBHIP_OK) rc = allreduce(0);
            if (rc == FBHIP_OK) rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_FB_STEP, s);
            if (rc != FBHIP_OK) break;
            const bool more = i + 1 < n_steps;
            if (more) {                          // fork: V of this step's actor phase, then the next step's head on the twin workspace set
                if ((he = hipEventRecord(c->events[2 * i], s)) != hipSuccess) break;
                if ((he = hipStreamWaitEvent(c->side, c->events[2 * i], 0)) != hipSuccess) break;
                rc = enqueue_actor_v(c, c->side);
                if (rc != FBHIP_OK) break;
                if ((he = hipEventRecord(c->events[128 + i], c->side)) != hipSuccess) break;
                c->cur ^= 1;
                rc = enqueue_update(c, *hp, nullptr, HEAD, c->side);
                c->cur ^= 1;
                if (rc != FBHIP_OK) break;
                c->v_ready = c->events[128 + i];
            }
            rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_ACTOR_GRAD, s);
            c->v_ready = nullptr;
            if (rc == FBHIP_OK && has_actor) rc = allreduce(1);        // ... beside the next step's head
            if (rc == FBHIP_OK) rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_ACTOR_STEP, s);
            if (rc != FBHIP_OK) break;
            if (more) {                          // join, then continue on the set the head filled
                if ((he = hipEventRecord(c->events[2 * i + 1], c->side)) != hipSuccess) break;
                if ((he = hipStreamWaitEvent(s, c->events[2 * i + 1], 0)) != hipSuccess) break;
                c->cur ^= 1;
            }
        }
        c->cur = cur0;
    } else if (dp) {
        for (int i = 0; i < n_steps && rc == FBHIP_OK; ++i) {
            rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_SAMPLE | FBHIP_PHASE_FB_GRAD | FBHIP_PHASE_ACTOR_FWD, s);
            if (rc == FBHIP_OK) rc = allreduce(0);
            if (rc == FBHIP_OK) rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_FB_STEP | FBHIP_PHASE_ACTOR_GRAD, s);
            if (rc == FBHIP_OK && has_actor) rc = allreduce(1);
            if (rc == FBHIP_OK) rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_ACTOR_STEP, s);
        }
    } else if (!pipe) {
        for (int i = 0; i < n_steps && rc == FBHIP_OK; ++i) rc = enqueue_update(c, *hp, injs ? &injs[i] : nullptr, FBHIP_PHASE_ALL, s);
    } else {
        const int HEAD = FBHIP_PHASE_SAMPLE | FBHIP_PHASE_FB_FWD_ONLINE;
        const int MID = FBHIP_PHASE_FB_FWD_TARGET | FBHIP_PHASE_FB_BWD | FBHIP_PHASE_ACTOR_FWD | FBHIP_PHASE_FB_STEP;   // (both FB_BWD bits)
        const int TAIL = FBHIP_PHASE_ACTOR_GRAD | FBHIP_PHASE_ACTOR_STEP;
        const int cur0 = c->cur;
        if (rc == FBHIP_OK) rc = enqueue_update(c, *hp, injs ? &injs[0] : nullptr, HEAD, s);
        for (int i = 0; i < n_steps && rc == FBHIP_OK && he == hipSuccess; ++i) {
            rc = enqueue_update(c, *hp, nullptr, MID, s);
            if (rc != FBHIP_OK) break;
            const bool more = i + 1 < n_steps;
            if (more) {                          // fork: V of this step's actor phase, then the next step's head on the twin workspace set
                // (capturing the side branch only when the actor phase reaches its first consumer of V, so that the critical path
                //  MID -> TAIL -> MID keeps ONE queue and the join has 100 us of slack, was measured in round 5: the trace shows the
                //  path on one queue, the rate moved by -0.5..+0.5 % -- a cross-queue edge costs its ~18 us wherever it lands)
                if ((he = hipEventRecord(c->events[2 * i], s)) != hipSuccess) break;
                if ((he = hipStreamWaitEvent(c->side, c->events[2 * i], 0)) != hipSuccess) break;
                rc = enqueue_actor_v(c, c->side);
                if (rc != FBHIP_OK) break;
                if ((he = hipEventRecord(c->events[128 + i], c->side)) != hipSuccess) break;
                c->cur ^= 1;
                rc = enqueue_update(c, *hp, injs ? &injs[i + 1] : nullptr, HEAD, c->side);
                c->cur ^= 1;
                if (rc != FBHIP_OK) break;
                c->v_ready = c->events[128 + i];
            }
            rc = enqueue_update(c, *hp, nullptr, TAIL, s);
            c->v_ready = nullptr;
            if (more) {                          // join, then continue on the set the head filled
                if ((he = hipEventRecord(c->events[2 * i + 1], c->side)) != hipSuccess) break;
                if ((he = hipStreamWaitEvent(s, c->events[2 * i + 1], 0)) != hipSuccess) break;
                c->cur ^= 1;
            }
        }
        c->cur = cur0;
    }
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != FBHIP_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (he != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); HIPCK(c, he); }
    HIPCK(c, e);
    if (const char* dot = getenv("FBHIP_GRAPH_DOT")) (void)hipGraphDebugDotPrint(graph, dot, 0);       // diagnostics: what a capture really holds
    GraphEntry ge{};
    ge.mask = FBHIP_PHASE_ALL | (dp ? DP_GRAPH_BIT : 0); ge.hp = *hp; ge.has_inj = injs != nullptr; ge.n_steps = n_steps; ge.set = c->cur;
    // (cache key: EVERY step's inject struct -- a caller that reuses its per-step buffers replays the same graph; one whose allocator
    // hands back the same addresses for step 0 only must not)
    if (injs) { ge.inj = injs[0]; ge.injs.assign(injs, injs + n_steps); }
    ge.branches = pipe;
    e = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIPCK(c, e);
    ++c->graph_captures;
    if (c->graphs.size() >= 16) { (void)hipGraphExecDestroy(c->graphs.front().exec); c->graphs.erase(c->graphs.begin()); }
    c->graphs.push_back(ge);
    if (!launch) return FBHIP_OK;
    RC(launch_graph(c, ge.exec, s, ge.branches));
    c->pub_issued += (unsigned int)n_steps * publishes(c, hp, FBHIP_PHASE_ALL);
    return FBHIP_OK;
}

int fbhip_update_many(fbhip_ctx* c, const fbhip_hparams* hp, int32_t n_steps, void* stream) {
    return update_many_impl(c, hp, n_steps, nullptr, stream);
}

int64_t fbhip_graph_captures(const fbhip_ctx* c) { return c ? c->graph_captures : -1; }

int fbhip_dp_bind_peers(fbhip_ctx* c, int32_t world, int32_t rank, float* const* fb_grad_ptrs, float* const* actor_grad_ptrs,
                        int32_t* const* flag_ptrs, int32_t* local_state) {
    RC(need_bound(c, false));
    for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);            // the pointers are baked into captured launches
    c->graphs.clear();
    if (world <= 1) { c->peers = PeerComm{}; return FBHIP_OK; }
    const bool has_actor = c->d.discrete == 0;
    if (world > PEER_MAX_WORLD || rank < 0 || rank >= world || !fb_grad_ptrs || !flag_ptrs || !local_state || (has_actor && !actor_grad_ptrs)) {
        c->err = g_err = "fbhip_dp_bind_peers: bad argument (world <= 8, 0 <= rank < world, non-null pointer tables)"; return FBHIP_E_INVALID;
    }
    PeerComm pc{};
    pc.world = world; pc.rank = rank; pc.state = (PeerState*)local_state;
    for (int q = 0; q < world; ++q) {
        pc.bucket[0][q] = fb_grad_ptrs[q]; pc.bucket[1][q] = has_actor ? actor_grad_ptrs[q] : nullptr; pc.flags[q] = flag_ptrs[q];
        if (!pc.bucket[0][q] || !pc.flags[q] || (has_actor && !pc.bucket[1][q]) || ((uintptr_t)pc.bucket[0][q] & 15) || ((uintptr_t)pc.bucket[1][q] & 15)) {
            c->err = g_err = "fbhip_dp_bind_peers: null / unaligned peer pointer"; return FBHIP_E_INVALID;
        }
    }
    if (pc.bucket[0][rank] != c->fb_g || (has_actor && pc.bucket[1][rank] != c->a_g)) {
        c->err = g_err = "fbhip_dp_bind_peers: entry [rank] must be this context's own gradient buffers"; return FBHIP_E_INVALID;
    }
    c->peers = pc;
    return FBHIP_OK;
}

int fbhip_peer_allreduce(fbhip_ctx* c, int32_t which, void* stream) {
    RC(need_bound(c, false));
    if (c->peers.world < 2) { c->err = g_err = "fbhip_peer_allreduce: no peers bound (fbhip_dp_bind_peers)"; return FBHIP_E_STATE; }
    if (which < 0 || which > 1 || (which == 1 && c->d.discrete)) return FBHIP_E_INVALID;
    const int64_t n = which == 0 ? c->L[FBHIP_NET_FORWARD].numel + c->L[FBHIP_NET_BACKWARD].numel : c->L[FBHIP_NET_ACTOR].numel;
    HIPCK(c, launch_peer_allreduce(c->peers, which, n, (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_update_many_dp(fbhip_ctx* c, const fbhip_hparams* hp, int32_t n_steps, void* stream) {
    RC(need_bound(c, true));
    if (c->peers.world < 2 && c->rccl_comm == nullptr) { c->err = g_err = "fbhip_update_many_dp: no transport bound (fbhip_rccl_init or fbhip_dp_bind_peers)"; return FBHIP_E_STATE; }
    return update_many_impl(c, hp, n_steps, nullptr, stream, /*dp=*/true);
}

int fbhip_update_many_dp_prepare(fbhip_ctx* c, const fbhip_hparams* hp, int32_t n_steps, void* stream) {
    RC(need_bound(c, true));
    if (c->peers.world < 2 && c->rccl_comm == nullptr) { c->err = g_err = "fbhip_update_many_dp_prepare: no transport bound (fbhip_rccl_init or fbhip_dp_bind_peers)"; return FBHIP_E_STATE; }
    return update_many_impl(c, hp, n_steps, nullptr, stream, /*dp=*/true, /*launch=*/false);
}

int fbhip_rccl_load(const char* library_path) { return rccl_load(library_path); }
int fbhip_rccl_version(void) { return rccl_version(); }
int fbhip_rccl_unique_id(void* out_128_bytes) {
    if (!out_128_bytes) return FBHIP_E_INVALID;
    return rccl_unique_id(out_128_bytes);
}
int fbhip_rccl_init(fbhip_ctx* c, const void* unique_id_128_bytes, int32_t world, int32_t rank, void* stream) {
    RC(need_bound(c, false));
    HIPCK(c, hipDeviceSynchronize());                                       // (an exec destroyed under an in-flight launch corrupts the runtime)
    for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);            // the communicator is baked into captured launches
    c->graphs.clear();
    if (unique_id_128_bytes == nullptr && world == 0) {          // release; rank != 0: after a failure, by ncclCommAbort
        rccl_release(c, rank != 0); c->rccl_world = 0; c->rccl_rank = 0; return FBHIP_OK;
    }
    return rccl_init(c, unique_id_128_bytes, world, rank, (hipStream_t)stream);
}

int fbhip_dp_status(fbhip_ctx* c, int32_t* host_status, void* stream) {
    RC(need_bound(c, false));
    if (!host_status) return FBHIP_E_INVALID;
    *host_status = 0;
    if (c->peers.world < 2) return FBHIP_OK;
    PeerState h{};
    HIPCK(c, hipMemcpyAsync(&h, c->peers.state, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCK(c, hipStreamSynchronize((hipStream_t)stream));
    *host_status = h.status;
    return FBHIP_OK;
}

int fbhip_order_legacy_stream_after(fbhip_ctx* c, void* stream) {
    if (!c) return FBHIP_E_INVALID;
    return order_legacy_after(c, (hipStream_t)stream);
}

int fbhip_order_stream_after_legacy(fbhip_ctx* c, void* stream) {
    if (!c) return FBHIP_E_INVALID;
    return order_after_legacy(c, (hipStream_t)stream);
}

int fbhip_update_many_injected(fbhip_ctx* c, const fbhip_hparams* hp, int32_t n_steps, const fbhip_inject* injects, void* stream) {
    if (!injects) { if (c) c->err = g_err = "fbhip_update_many_injected: null injects"; return FBHIP_E_INVALID; }
    return update_many_impl(c, hp, n_steps, injects, stream);
}

int fbhip_read_metrics(fbhip_ctx* c, float* host_out, void* stream) {
    RC(need_bound(c, false));
    if (!host_out) return FBHIP_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    HIPCK(c, hipMemcpyAsync(host_out, c->W().metrics, FBHIP_NUM_METRICS * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCK(c, hipStreamSynchronize(s));
    return FBHIP_OK;
}

int fbhip_wait_metrics(fbhip_ctx* c, float* host_out) {
    RC(need_bound(c, false));
    if (!host_out) return FBHIP_E_INVALID;
    if (c->h_metrics != nullptr) {
        const unsigned int* seq = reinterpret_cast<const unsigned int*>(c->h_metrics + FBHIP_NUM_METRICS);
        const unsigned int want = c->pub_issued;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned int spins = 0;; ++spins) {
            if ((int)(__atomic_load_n(seq, __ATOMIC_ACQUIRE) - want) >= 0) {
                memcpy(host_out, c->h_metrics, FBHIP_NUM_METRICS * sizeof(float));
                return FBHIP_OK;
            }
            __builtin_ia32_pause();
            if ((spins & 0x3fff) == 0x3fff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;
        }
        // the number never arrived (a launch that failed after it was counted?): drain the stream, take what the device holds and
        // re-align the count with what was really published
        HIPCK(c, hipStreamSynchronize(c->last_stream));
        c->pub_issued = __atomic_load_n(seq, __ATOMIC_ACQUIRE);
    }
    return fbhip_read_metrics(c, host_out, c->last_stream);
}

int fbhip_workspace_view(fbhip_ctx* c, const char* name, float** ptr, int32_t* rows, int32_t* cols, int32_t* ld) {
    RC(need_bound(c, false));
    if (!name || !ptr) return FBHIP_E_INVALID;
    Ws& w = c->W();
    const fbhip_dims& d = c->d;
    const int B = d.batch, o = d.obs_dim, a = d.action_dim, z = d.z_dim;
    const int aoff = geom_of(d).single ? o + z : o;
    std::map<std::string, Buf> m = {
        {"Xoa", w.Xoa}, {"Xoz", w.Xoz}, {"Xnoz", w.Xnoz}, {"Xnoa", w.Xnoa}, {"Xopi", w.Xopi}, {"next_goal", w.next_goal},
        {"backward_input", w.bin}, {"future_goal", w.fgoal}, {"z", w.z}, {"zrand", w.zrand}, {"F1", w.fsO.F1}, {"F2", w.fsO.F2}, {"Bm", d.norm_z ? w.bsO.Bm : w.bsO.y},
        {"y", w.bsO.y}, {"tF1", w.fsT.F1}, {"tF2", w.fsT.F2}, {"tB", d.norm_z ? w.bsA.Bm : w.bsA.y}, {"dF1", w.dF1}, {"dF2", w.dF2},
        {"dBm", w.dBm}, {"dy", d.norm_z ? w.dy : w.dBm}, {"mu", w.as.mu}, {"d_premu", w.a_dpremu},
        {"dp", w.dp}, {"dh", w.dh}, {"dt1a", w.dt1a}, {"a_dp", w.a_dp}, {"actor_p", w.as.p}, {"actor_h", w.as.h},
        {"actor_premu", w.as.premu}, {"online_p", w.fsO.p}, {"online_h", w.fsO.h},
        {"phi2", w.bsS.Bm}, {"dphi2", w.dBm2}, {"dy2", w.dy2}};          // SFAgent: [phi(goal) ; phi(next_goal)] and its gradients, 2B rows
    Buf b;
    const std::string n(name);
    if (m.count(n)) b = m[n];
    else if (n == "obs") { b = w.Xoz; b.cols = o; }
    else if (n == "next_obs") { b = w.Xnoz; b.cols = o; }
    else if (n == "action") { b = w.Xoa; b.p += aoff; b.cols = a; }
    else if (n == "next_action") { b = w.Xnoa; b.p += aoff; b.cols = a; }
    else if (n == "pi_action") { b = w.Xopi; b.p += aoff; b.cols = a; }
    else if (n == "discount") { b.p = w.disc; b.rows = B; b.cols = 1; b.ld = 1; }
    else if (n == "metrics") { b.p = w.metrics; b.rows = 1; b.cols = FBHIP_NUM_METRICS; b.ld = FBHIP_NUM_METRICS; }
    else if (n == "z_gauss") { b.p = w.so.z_gauss; b.rows = B; b.cols = z; b.ld = z; }
    else if (n == "ep_idx") { b.p = (float*)w.so.ep_idx; b.rows = 1; b.cols = B; b.ld = B; }
    else if (n == "step_idx") { b.p = (float*)w.so.step_idx; b.rows = 1; b.cols = B; b.ld = B; }
    else if (n == "perm") { b.p = (float*)w.so.perm; b.rows = 1; b.cols = B; b.ld = B; }
    else if (n == "mix_uniform") { b.p = w.so.mix_uniform; b.rows = 1; b.cols = B; b.ld = B; }
    else if (n == "rand_weight") { b.p = w.rw; b.rows = B; b.cols = B; b.ld = B; }
    else if (n == "rand_weight_u") { b.p = w.rw_u; b.rows = 1; b.cols = B; b.ld = B; }
    else if (n == "future_idx") { b.p = (float*)w.so.future_idx; b.rows = 1; b.cols = B; b.ld = B; }
    else if (n == "future_uniform") { b.p = w.so.future_uniform; b.rows = 1; b.cols = B; b.ld = B; }
    else if (n == "eps_next") { b.p = w.so.eps_next; b.rows = B; b.cols = a; b.ld = a; }
    else if (n == "eps_actor") { b.p = w.so.eps_actor; b.rows = B; b.cols = a; b.ld = a; }
    else { c->err = g_err = "fbhip: unknown workspace view '" + n + "'"; return FBHIP_E_INVALID; }
    *ptr = b.p;
    if (rows) *rows = b.rows;
    if (cols) *cols = b.cols;
    if (ld) *ld = b.ld;
    return FBHIP_OK;
}

// ---- inference entry points ----------------------------------------------------------------------------------
// ---- batch-1 fast path: one hipGraph = H2D of the staged inputs + 4 launches + D2H of the result ---------------
namespace {

enum { INFER_ACT = 0, INFER_ZCORREL = 1, INFER_DISCRETE_ACT = 2 };

GemvProblem GV(const float* x, const float* W, int ldw, const float* bias, float* y, int N, int K, bool relu,
               const float* ln_g = nullptr, const float* ln_b = nullptr, int n_ln = 0) {
    GemvProblem p{};
    p.x = x; p.W = W; p.bias = bias; p.y = y; p.ln_g = ln_g; p.ln_b = ln_b;
    p.N = N; p.K = K; p.ldw = ldw; p.n_ln = n_ln; p.relu = relu ? 1 : 0;
    return p;
}

int enqueue_act(fbhip_ctx* c, float stddev, int eval_mode, bool has_noise, hipStream_t s) {
    const fbhip_dims& d = c->d;
    const int o = d.obs_dim, z = d.z_dim, a = d.action_dim, H = d.hidden_dim;
    Ws& w = c->W();
    const ActP& A = c->A_p;
    float* pre1o = w.act_vec; float* pre1z = pre1o + 2048; float* h = pre1z + 2048; float* pv = h + 2048;
    const size_t nin = act_in_floats(d);       // obs, z, noise and -- in the last float -- the exploration stddev
    HIPCK(c, hipMemcpyAsync(w.act_in, c->h_in, nin * sizeof(float), hipMemcpyHostToDevice, s));
    // Actor.forward (fb_modules.py:107-121): first layers (the weight's zero pad columns absorb whatever follows
    // obs / [obs|z] in the staging vector); preprocess == 0 has ONE branch on [obs|z] ...
    const Geom gm = actor_geom_of(d);
    GemvGroup g1{}; g1.n = gm.single ? 1 : 2;
    g1.p[0] = GV(w.act_in, A.o.W1, A.o.ld1, A.o.b1, pre1o, H, A.o.ld1, false);
    g1.p[1] = GV(w.act_in, A.oz.W1, A.oz.ld1, A.oz.b1, pre1z, H, A.oz.ld1, false);
    HIPCK(c, launch_gemv_group(g1, s));
    // ... LayerNorm + tanh as the prologue of the second layers, ReLU ...
    GemvGroup g2{}; g2.n = gm.single ? 1 : 2;
    g2.p[0] = GV(pre1o, A.o.W2, H, A.o.b2, h, gm.Fo, H, true, A.o.g1, A.o.be1, H);
    g2.p[1] = GV(pre1z, A.oz.W2, H, A.oz.b2, h + gm.Fo, gm.Fo, H, true, A.oz.g1, A.oz.be1, H);
    HIPCK(c, launch_gemv_group(g2, s));
    // ... (trunk layer: one more Linear + ReLU, fb_modules.py:116-117) policy hidden layer ...
    const float* feat = h;
    int nfeat = gm.hw;
    if (gm.trunk) {
        float* tr = pv + 2048;
        GemvGroup gt{}; gt.n = 1;
        gt.p[0] = GV(h, A.Wt, gm.hw, A.bt, tr, H, gm.hw, true);
        HIPCK(c, launch_gemv_group(gt, s));
        feat = tr; nfeat = H;
    }
    if (!gm.boltz) {
        GemvGroup g3{}; g3.n = 1;
        g3.p[0] = GV(feat, A.W3, nfeat, A.b3, pv, H, nfeat, true);
        HIPCK(c, launch_gemv_group(g3, s));
        feat = pv;
    }
    // ... head + TruncatedNormal / SquashedNormal
    HIPCK(c, launch_act_head(feat, A.W4, H, A.b4, a, H, stddev, eval_mode, has_noise ? w.act_in + act_noise_off(d) : nullptr,
                             c->seed, c->rank, w.st, w.act_out, c->sq, s, w.act_in + (act_in_floats(d) - 1),
                             c->infer_direct ? c->h_out : nullptr, c->infer_direct ? c->d_pubseq + 2 : nullptr));
    if (!c->infer_direct) HIPCK(c, hipMemcpyAsync(c->h_out, w.act_out, (size_t)a * sizeof(float), hipMemcpyDeviceToHost, s));
    (void)o; (void)z;
    return FBHIP_OK;
}

// DiscreteFBAgent.act without exploration (discrete_fb.py:258-268) for ONE observation: the GEMV chain of the batch-1 path
// through forward_net (trunk on [obs|z], heads z * A wide) + the selection row kernel; activations live in row 0 of the
// target-side ForwardMap set
int enqueue_discrete_act(fbhip_ctx* c, hipStream_t s) {
    const fbhip_dims& d = c->d;
    const int o = d.obs_dim, z = d.z_dim, H = d.hidden_dim, zA = fhead_out(d);
    Ws& w = c->W();
    const FwdP& F = c->F_p;
    FSet& S = w.fsT;
    HIPCK(c, hipMemcpyAsync(w.act_in, c->h_in, act_noise_off(d) * sizeof(float), hipMemcpyHostToDevice, s));
    GemvGroup g1{}; g1.n = 1;
    g1.p[0] = GV(w.act_in, F.oa.W1, F.oa.ld1, F.oa.b1, S.pre1a.p, H, F.oa.ld1, false);
    HIPCK(c, launch_gemv_group(g1, s));
    GemvGroup g2{}; g2.n = 1;
    g2.p[0] = GV(S.pre1a.p, F.oa.W2, H, F.oa.b2, S.h.p, H, H, true, F.oa.g1, F.oa.be1, H);
    HIPCK(c, launch_gemv_group(g2, s));
    GemvGroup g3{}; g3.n = 1;
    g3.p[0] = GV(S.h.p, F.Wt, H, F.bt, S.tr.p, H, H, true);
    HIPCK(c, launch_gemv_group(g3, s));
    GemvGroup g4{}; g4.n = 1;
    g4.p[0] = GV(S.tr.p, F.W3s, H, F.b3s, S.p.p, 2 * H, H, true);
    HIPCK(c, launch_gemv_group(g4, s));
    GemvGroup g5{}; g5.n = 2;
    g5.p[0] = GV(S.p.p, F.W4[0], H, F.b4[0], S.Fall1.p, zA, H, false);
    g5.p[1] = GV(S.p.p + H, F.W4[1], H, F.b4[1], S.Fall2.p, zA, H, false);
    HIPCK(c, launch_gemv_group(g5, s));
    HIPCK(c, launch_discrete_select(S.Fall1.p, S.Fall2.p, pad4(zA), w.act_in + o, z, S.F1.p, S.F2.p, pad4(z), w.nextq,
                                    (int32_t*)w.act_out, 1, z, d.action_dim, d.boltzmann, c->sq.temp, s));
    HIPCK(c, hipMemcpyAsync(c->h_out, w.act_out, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    return FBHIP_OK;
}

int enqueue_zcorrel(fbhip_ctx* c, hipStream_t s) {
    const fbhip_dims& d = c->d;
    const int g = d.goal_dim, z = d.z_dim, Hb = d.backward_hidden_dim, Lb = pad64(Hb);
    Ws& w = c->W();
    const BwdP& K = c->K_p;
    float* pre1 = w.act_vec; float* r2 = pre1 + 2048; float* y = r2 + 2048;
    HIPCK(c, hipMemcpyAsync(w.act_in, c->h_in, (act_z_off(d) + z) * sizeof(float), hipMemcpyHostToDevice, s));
    if (d.backward_identity) {                   // IdentityMap: B(goal) = goal, no projection
        HIPCK(c, launch_zcorrel(w.act_in, w.act_in + act_z_off(d), z, 0, w.act_out, s, c->infer_direct ? c->h_out : nullptr,
                                c->infer_direct ? c->d_pubseq + 2 : nullptr));
        if (!c->infer_direct) HIPCK(c, hipMemcpyAsync(c->h_out, w.act_out, sizeof(float), hipMemcpyDeviceToHost, s));
        return FBHIP_OK;
    }
    // BackwardMap.forward (fb_modules.py:223-230) on the padded layout: pad rows of W1 / W2 are zero
    GemvGroup g1{}; g1.n = 1;
    g1.p[0] = GV(w.act_in, K.W1, pad32(g), K.b1, pre1, Lb, pad32(g), false);
    HIPCK(c, launch_gemv_group(g1, s));
    GemvGroup g2{}; g2.n = 1;
    g2.p[0] = GV(pre1, K.W2, Lb, K.b2, r2, Lb, Lb, true, K.g1, K.be1, Hb);
    HIPCK(c, launch_gemv_group(g2, s));
    GemvGroup g3{}; g3.n = 1;
    g3.p[0] = GV(r2, K.W3, Lb, K.b3, y, z, Lb, false);
    HIPCK(c, launch_gemv_group(g3, s));
    HIPCK(c, launch_zcorrel(y, w.act_in + act_z_off(d), z, d.norm_z, w.act_out, s, c->infer_direct ? c->h_out : nullptr,
                            c->infer_direct ? c->d_pubseq + 2 : nullptr));
    if (!c->infer_direct) HIPCK(c, hipMemcpyAsync(c->h_out, w.act_out, sizeof(float), hipMemcpyDeviceToHost, s));
    return FBHIP_OK;
}

// replay (or capture + replay) the graph of one fast-path call, then wait for its result
int run_infer_graph(fbhip_ctx* c, int kind, float stddev, int eval_mode, bool has_noise, hipStream_t s) {
    hipGraphExec_t exec = nullptr;
    for (auto& g : c->infer_graphs)
        if (g.kind == kind && g.eval_mode == eval_mode && g.has_noise == (int)has_noise) exec = g.exec;      // (stddev travels in the staged inputs)
    if (!exec) {
        hipGraph_t graph = nullptr;
        HIPCK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = kind == INFER_ACT ? enqueue_act(c, stddev, eval_mode, has_noise, s)
                       : kind == INFER_DISCRETE_ACT ? enqueue_discrete_act(c, s) : enqueue_zcorrel(c, s);
        hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc != FBHIP_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        HIPCK(c, e);
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIPCK(c, e);
        if (c->infer_graphs.size() >= 8) { (void)hipGraphExecDestroy(c->infer_graphs.front().exec); c->infer_graphs.erase(c->infer_graphs.begin()); }
        c->infer_graphs.push_back(InferGraph{kind, eval_mode, (int)has_noise, stddev, exec});
    }
    HIPCK(c, hipGraphLaunch(exec, s));
    if (c->infer_direct && kind != INFER_DISCRETE_ACT) {
        // the graph's last kernel writes the result into h_out and then a sequence number: spin on it (a few microseconds after the
        // kernel's store -- no D2H copy node, no interrupt / signal wait of hipStreamSynchronize); the stream drains on its own
        const unsigned int want = ++c->infer_issued;
        const unsigned int* seq = reinterpret_cast<const unsigned int*>(c->h_out + INFER_SEQ_SLOT);
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned int spins = 0;; ++spins) {
            if ((int)(__atomic_load_n(seq, __ATOMIC_ACQUIRE) - want) >= 0) return FBHIP_OK;
            __builtin_ia32_pause();
            if ((spins & 0x3fff) == 0x3fff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;
        }
        HIPCK(c, hipStreamSynchronize(s));       // (never arrived: drain, take what the kernel stored, re-align the count)
        c->infer_issued = __atomic_load_n(seq, __ATOMIC_ACQUIRE);
        return FBHIP_OK;
    }
    HIPCK(c, hipStreamSynchronize(s));
    return FBHIP_OK;
}

}  // namespace

int fbhip_act(fbhip_ctx* c, const float* host_obs, const float* host_z, const float* host_noise, float stddev,
              int32_t eval_mode, float* host_action_out, void* stream) {
    RC(need_bound(c, false));
    if (!host_obs || !host_z || !host_action_out) { c->err = g_err = "fbhip_act: null argument"; return FBHIP_E_INVALID; }
    if (c->d.discrete) { c->err = g_err = "fbhip_act: discrete context has no actor (use fbhip_discrete_act)"; return FBHIP_E_STATE; }
    if (!c->h_in) { c->err = g_err = "fbhip_act: pinned staging unavailable"; return FBHIP_E_STATE; }
    if (c->d.hidden_dim > 2048 || actor_geom_of(c->d).hw > 2048) { c->err = g_err = "fbhip_act: layer wider than 2048"; return FBHIP_E_INVALID; }
    const fbhip_dims& d = c->d;
    memcpy(c->h_in, host_obs, (size_t)d.obs_dim * sizeof(float));
    memcpy(c->h_in + d.obs_dim, host_z, (size_t)d.z_dim * sizeof(float));
    for (size_t i = (size_t)d.obs_dim + d.z_dim; i < act_noise_off(d); ++i) c->h_in[i] = 0.f;
    const bool has_noise = host_noise != nullptr && !eval_mode;
    if (has_noise) memcpy(c->h_in + act_noise_off(d), host_noise, (size_t)d.action_dim * sizeof(float));
    c->h_in[act_in_floats(d) - 1] = stddev;
    RC(run_infer_graph(c, INFER_ACT, stddev, eval_mode ? 1 : 0, has_noise, (hipStream_t)stream));
    memcpy(host_action_out, c->h_out, (size_t)d.action_dim * sizeof(float));
    return FBHIP_OK;
}

int fbhip_discrete_act_host(fbhip_ctx* c, const float* host_obs, const float* host_z, int32_t* host_action_out, void* stream) {
    RC(need_bound(c, false));
    if (!host_obs || !host_z || !host_action_out) { c->err = g_err = "fbhip_discrete_act_host: null argument"; return FBHIP_E_INVALID; }
    if (!c->d.discrete) { c->err = g_err = "fbhip_discrete_act_host: the context was not created with discrete"; return FBHIP_E_STATE; }
    if (!c->h_in) { c->err = g_err = "fbhip_discrete_act_host: pinned staging unavailable"; return FBHIP_E_STATE; }
    const fbhip_dims& d = c->d;
    memcpy(c->h_in, host_obs, (size_t)d.obs_dim * sizeof(float));
    memcpy(c->h_in + d.obs_dim, host_z, (size_t)d.z_dim * sizeof(float));
    for (size_t i = (size_t)d.obs_dim + d.z_dim; i < act_noise_off(d); ++i) c->h_in[i] = 0.f;
    RC(run_infer_graph(c, INFER_DISCRETE_ACT, 0.f, 1, false, (hipStream_t)stream));
    memcpy(host_action_out, c->h_out, sizeof(int32_t));
    return FBHIP_OK;
}

int fbhip_z_correl(fbhip_ctx* c, const float* host_goal, const float* host_z, float* host_out, void* stream) {
    RC(need_bound(c, false));
    if (!host_goal || !host_z || !host_out) { c->err = g_err = "fbhip_z_correl: null argument"; return FBHIP_E_INVALID; }
    if (!c->h_in) { c->err = g_err = "fbhip_z_correl: pinned staging unavailable"; return FBHIP_E_STATE; }
    const fbhip_dims& d = c->d;
    memcpy(c->h_in, host_goal, (size_t)d.goal_dim * sizeof(float));
    for (size_t i = (size_t)d.goal_dim; i < act_z_off(d); ++i) c->h_in[i] = 0.f;
    memcpy(c->h_in + act_z_off(d), host_z, (size_t)d.z_dim * sizeof(float));
    RC(run_infer_graph(c, INFER_ZCORREL, 0.f, 0, false, (hipStream_t)stream));
    *host_out = c->h_out[0];
    return FBHIP_OK;
}

int fbhip_actor_forward(fbhip_ctx* c, const float* obs, int32_t ld_obs, const float* z, int32_t ld_z, int32_t rows,
                        const float* noise, float stddev, float clip, float* action_out, int32_t ld_out, void* stream) {
    RC(need_bound(c, false));
    if (!obs || !z || !action_out || rows < 1) return FBHIP_E_INVALID;
    if (c->d.discrete) { c->err = g_err = "fbhip_actor_forward: discrete context has no actor"; return FBHIP_E_STATE; }
    hipStream_t s = (hipStream_t)stream;
    const fbhip_dims& d = c->d;
    Ws& w = c->W();
    const int La = pad4(head_width(d));
    for (int r0 = 0; r0 < rows; r0 += d.batch) {
        const int n = rows - r0 < d.batch ? rows - r0 : d.batch;
        HIPCK(c, launch_concat2(w.Xoz.p, w.Xoz.ld, obs + (size_t)r0 * ld_obs, ld_obs, d.obs_dim, z + (size_t)r0 * ld_z, ld_z,
                                d.z_dim, n, s));
        RC(actor_fwd(c, c->A_p, w.Xoz.p, w.Xoz.ld, w.Xoz.p, w.Xoz.ld, w.as, n, s));
        HIPCK(c, launch_policy_sample(w.as.premu.p, La, noise ? noise + (size_t)r0 * d.action_dim : nullptr, d.action_dim,
                                      stddev, clip, nullptr, 0, action_out + (size_t)r0 * ld_out, ld_out, n, d.action_dim, c->sq, s));
    }
    return FBHIP_OK;
}

int fbhip_backward_map(fbhip_ctx* c, int32_t which, const float* goal, int32_t ld_goal, int32_t rows, float* out,
                       int32_t ld_out, void* stream) {
    RC(need_bound(c, false));
    if (!goal || !out || rows < 1) return FBHIP_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const fbhip_dims& d = c->d;
    Ws& w = c->W();
    for (int r0 = 0; r0 < rows; r0 += d.batch) {
        const int n = rows - r0 < d.batch ? rows - r0 : d.batch;
        RC(backward_map_fwd(c, which ? c->K_t : c->K_p, goal + (size_t)r0 * ld_goal, ld_goal, w.bsA, n, s));
        const Buf& bo = d.norm_z ? w.bsA.Bm : w.bsA.y;
        HIPCK(c, launch_concat2(out + (size_t)r0 * ld_out, ld_out, bo.p, bo.ld, d.z_dim, nullptr, 0, 0, n, s));
    }
    return FBHIP_OK;
}

int fbhip_forward_map(fbhip_ctx* c, int32_t which, const float* obs, int32_t ld_obs, const float* z, int32_t ld_z,
                      const float* action, int32_t ld_act, int32_t rows, float* f1_out, float* f2_out, int32_t ld_out,
                      void* stream) {
    RC(need_bound(c, false));
    if (!obs || !z || !action || !f1_out || !f2_out || rows < 1) return FBHIP_E_INVALID;
    if (c->d.discrete) { c->err = g_err = "fbhip_forward_map: discrete context (use fbhip_discrete_act)"; return FBHIP_E_STATE; }
    hipStream_t s = (hipStream_t)stream;
    const fbhip_dims& d = c->d;
    Ws& w = c->W();
    for (int r0 = 0; r0 < rows; r0 += d.batch) {
        const int n = rows - r0 < d.batch ? rows - r0 : d.batch;
        if (geom_of(d).single) {                 // one panel [obs | z | action]
            HIPCK(c, launch_concat2(w.Xoa.p, w.Xoa.ld, obs + (size_t)r0 * ld_obs, ld_obs, d.obs_dim, z + (size_t)r0 * ld_z, ld_z, d.z_dim, n, s));
            HIPCK(c, launch_concat2(w.Xoa.p + d.obs_dim + d.z_dim, w.Xoa.ld, action + (size_t)r0 * ld_act, ld_act, d.action_dim, nullptr, 0, 0, n, s));
        } else {
            HIPCK(c, launch_concat2(w.Xoz.p, w.Xoz.ld, obs + (size_t)r0 * ld_obs, ld_obs, d.obs_dim, z + (size_t)r0 * ld_z, ld_z, d.z_dim, n, s));
            HIPCK(c, launch_concat2(w.Xoa.p, w.Xoa.ld, obs + (size_t)r0 * ld_obs, ld_obs, d.obs_dim, action + (size_t)r0 * ld_act, ld_act, d.action_dim, n, s));
        }
        RC(forward_map_fwd(c, which ? c->F_t : c->F_p, w.Xoa.p, w.Xoa.ld, w.Xoz.p, w.Xoz.ld, w.fsT, n, s));
        HIPCK(c, launch_concat2(f1_out + (size_t)r0 * ld_out, ld_out, w.fsT.F1.p, w.fsT.F1.ld, d.z_dim, nullptr, 0, 0, n, s));
        HIPCK(c, launch_concat2(f2_out + (size_t)r0 * ld_out, ld_out, w.fsT.F2.p, w.fsT.F2.ld, d.z_dim, nullptr, 0, 0, n, s));
    }
    return FBHIP_OK;
}

int fbhip_discrete_act(fbhip_ctx* c, int32_t which, const float* obs, int32_t ld_obs, const float* z, int32_t ld_z,
                       int32_t rows, int32_t* action_out, float* next_q_out, float* f1_out, float* f2_out, int32_t ld_out,
                       void* stream) {
    RC(need_bound(c, false));
    if (!obs || !z || rows < 1) return FBHIP_E_INVALID;
    if (!c->d.discrete) { c->err = g_err = "fbhip_discrete_act: the context was not created with discrete"; return FBHIP_E_STATE; }
    hipStream_t s = (hipStream_t)stream;
    const fbhip_dims& d = c->d;
    Ws& w = c->W();
    for (int r0 = 0; r0 < rows; r0 += d.batch) {
        const int n = rows - r0 < d.batch ? rows - r0 : d.batch;
        HIPCK(c, launch_concat2(w.Xoz.p, w.Xoz.ld, obs + (size_t)r0 * ld_obs, ld_obs, d.obs_dim, z + (size_t)r0 * ld_z, ld_z, d.z_dim, n, s));
        Chain ch;
        forward_map_fwd_chain(c, which ? c->F_t : c->F_p, w.Xoz.p, w.Xoz.ld, w.Xoz.p, w.Xoz.ld, w.fsT, n, ch, true, 1,
                              z + (size_t)r0 * ld_z, ld_z);
        RC(run_chain(c, ch, s));
        if (action_out) HIPCK(c, hipMemcpyAsync(action_out + r0, w.greedy, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        if (next_q_out) HIPCK(c, hipMemcpyAsync(next_q_out + r0, w.nextq, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        if (f1_out) HIPCK(c, launch_concat2(f1_out + (size_t)r0 * ld_out, ld_out, w.fsT.F1.p, w.fsT.F1.ld, d.z_dim, nullptr, 0, 0, n, s));
        if (f2_out) HIPCK(c, launch_concat2(f2_out + (size_t)r0 * ld_out, ld_out, w.fsT.F2.p, w.fsT.F2.ld, d.z_dim, nullptr, 0, 0, n, s));
    }
    return FBHIP_OK;
}

// ---- individually testable kernels -------------------------------------------------------------------------------
int fbhip_gemm(const float* A, int32_t lda, int32_t a_kcontig, const float* B, int32_t ldb, int32_t b_kcontig, float* C,
               int32_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, const float* aux, int32_t ldaux,
               int32_t epi, float* colsum, void* stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || epi < 0 || epi > 4) { g_err = "fbhip_gemm: bad argument"; return FBHIP_E_INVALID; }
    if ((epi == EPI_BIAS || epi == EPI_BIAS_RELU) && !bias) { g_err = "fbhip_gemm: bias required"; return FBHIP_E_INVALID; }
    if ((epi == EPI_MASK_RELU || epi == EPI_TANH_BWD) && !aux) { g_err = "fbhip_gemm: aux required"; return FBHIP_E_INVALID; }
    { fbhip_ctx* none = nullptr; HIPCK(none, gemm_init()); }
    return run_gemms(nullptr, {P(A, lda, a_kcontig, B, ldb, b_kcontig, C, ldc, M, N, K, bias, epi, aux, ldaux, colsum)},
                     (hipStream_t)stream);
}

int fbhip_gemm_cfg(const float* A, int32_t lda, int32_t a_kcontig, const float* B, int32_t ldb, int32_t b_kcontig,
                   float* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t cfg, void* stream) {
    if (cfg < 0 || cfg >= CFG_COUNT) return FBHIP_E_INVALID;
    GemmGroup g{};
    GemmProblem p = P(A, lda, a_kcontig, B, ldb, b_kcontig, C, ldc, M, N, K);
    gemm_problem_finalize(p, cfg);
    p.tile_start = 0;
    g.p[0] = p; g.n = 1; g.total_tiles = p.tiles_m * p.tiles_n;
    fbhip_ctx* none = nullptr;
    HIPCK(none, gemm_init());
    HIPCK(none, launch_gemm_group(g, cfg, (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_head(const float* x, int32_t ldx, const float* w, int32_t ldw, const float* bias, float* c, int32_t ldc, float* out2,
               int32_t ldo, float* norms, float scale, int32_t rows, int32_t N, int32_t K, int32_t replicas, void* stream) {
    fbhip_ctx* none = nullptr;
    if (!x || !w || !bias || !c || replicas < 1 || replicas > HEAD_MAX_GROUP) { g_err = "fbhip_head: bad argument"; return FBHIP_E_INVALID; }
    HeadGroup g{};
    for (int i = 0; i < replicas; ++i) g.p[g.n++] = HeadProblem{x, ldx, w, ldw, bias, c, ldc, out2, ldo, norms, scale, rows, N, K};
    if (!head_ok(g.p[0])) { g_err = "fbhip_head: needs N <= 64, K % 4 == 0, 16-byte aligned rows, ldc / ldo >= pad4(N)"; return FBHIP_E_INVALID; }
    HIPCK(none, launch_head_group(g, (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_ln_tanh_fwd(const float* x, int32_t ldx, const float* gamma, const float* beta, float* y, int32_t ldy,
                      float* stats, int32_t rows, int32_t n, void* stream) {
    fbhip_ctx* none = nullptr;
    HIPCK(none, launch_ln_tanh_fwd(x, ldx, gamma, beta, y, ldy, stats, rows, n, (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_ln_tanh_bwd(const float* dy, int32_t lddy, const float* y, int32_t ldy, const float* x, int32_t ldx,
                      const float* stats, const float* gamma, float* dx, int32_t lddx, float* dgamma, float* dbeta,
                      float* partials, int32_t rows, int32_t n, void* stream) {
    fbhip_ctx* none = nullptr;
    HIPCK(none, launch_ln_tanh_bwd(dy, lddy, y, ldy, x, ldx, stats, gamma, dx, lddx, dgamma, dbeta, partials, rows, n,
                                   (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_l2norm_fwd(const float* y, int32_t ldy, float* out, int32_t ldo, float* norms, int32_t rows, int32_t d,
                     void* stream) {
    fbhip_ctx* none = nullptr;
    HIPCK(none, launch_l2norm_fwd(y, ldy, out, ldo, norms, rows, d, sqrtf((float)d), (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_l2norm_bwd(const float* dB, int32_t lddb, const float* y, int32_t ldy, const float* norms, float* dy,
                     int32_t lddy, int32_t rows, int32_t d, void* stream) {
    fbhip_ctx* none = nullptr;
    HIPCK(none, launch_l2norm_bwd(dB, lddb, y, ldy, norms, dy, lddy, rows, d, (hipStream_t)stream));
    return FBHIP_OK;
}

size_t fbhip_pairwise_scratch_floats(int32_t B, int32_t d) { return pairwise_scratch_floats(B, d); }

int fbhip_pairwise_fb(const float* F1, const float* F2, const float* Bm, const float* tF1, const float* tF2,
                      const float* tB, const float* discount, int32_t B, int32_t d, int32_t ld, float ortho_coef,
                      float* dF1, float* dF2, float* dB, float* metrics, float* scratch, void* stream) {
    fbhip_ctx* none = nullptr;
    if (!scratch) { g_err = "fbhip_pairwise_fb: scratch required"; return FBHIP_E_INVALID; }
    HIPCK(none, pairwise_prepare(B, d));
    HIPCK(none, launch_pairwise_fb(F1, F2, Bm, tF1, tF2, tB, discount, B, d, ld, ortho_coef, dF1, dF2, dB, metrics, scratch,
                                   (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_pairwise_fb_block(const float* F1, const float* F2, const float* Bm, const float* tF1, const float* tF2,
                            const float* tB, const float* discount, int32_t B, int32_t d, int32_t ld, float ortho_coef,
                            int32_t row_offset, int32_t rows, float* dF1, float* dF2, float* dB, float* metrics,
                            float* scratch, void* stream) {
    fbhip_ctx* none = nullptr;
    if (!scratch) { g_err = "fbhip_pairwise_fb_block: scratch required"; return FBHIP_E_INVALID; }
    HIPCK(none, pairwise_prepare(B, d));
    HIPCK(none, launch_pairwise_fb_block(F1, F2, Bm, tF1, tF2, tB, discount, B, d, ld, ortho_coef, row_offset, rows, dF1, dF2,
                                         dB, metrics, scratch, (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_adam_ema(float* params, const float* grads, float* m, float* v, float* target, int64_t numel, float lr,
                   int32_t t, float grad_scale, float tau, void* stream) {
    fbhip_ctx* none = nullptr;
    HIPCK(none, launch_adam_ema(params, grads, m, v, target, numel, lr, lr, numel, grad_scale, tau, nullptr, 0, t,
                                (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_inverse(const float* A, int32_t lda, int32_t d, float scale, float* out, int32_t ldo, void* stream) {
    fbhip_ctx* none = nullptr;
    if (A == nullptr || out == nullptr || d < 1 || d > 128 || lda < d || ldo < d) { g_err = "fbhip_inverse: 1 <= d <= 128, lda / ldo >= d"; return FBHIP_E_INVALID; }
    HIPCK(none, launch_inverse(A, lda, d, scale, out, ldo, (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_actor_loss(const float* F1, const float* F2, int32_t ldf, const float* z, int32_t ldz, const float* mu,
                     int32_t ldmu, const float* action, int32_t lda, float stddev, float* dF1, float* dF2, float* metrics,
                     float* scratch, int32_t rows, int32_t d, int32_t a, void* stream) {
    fbhip_ctx* none = nullptr;
    HIPCK(none, launch_actor_loss(F1, F2, ldf, z, ldz, mu, ldmu, action, lda, stddev, dF1, dF2, metrics, scratch, rows, d, a,
                                  (hipStream_t)stream));
    return FBHIP_OK;
}

// per-kernel test exports of the two a-wide seams of the actor (rowops.hip row kernels / headtiles.hip MFMA tiles; the launchers
// pick the form: FBHIP_HEAD_TILES, batch rows): tests/test_kernels_gpu.py compares both forms with fp64 statements
int fbhip_policy_head(const float* P, int32_t ldp, const float* W4, int32_t ldw4, const float* b4, const float* noise, float stddev,
                      float clip, float* premu, float* mu, float* action, int32_t ld_out, const float* base, int32_t ldb,
                      const float* W1a, int32_t ldw1, const float* gamma, const float* beta, float* t1, int32_t ldt1, float* stats,
                      int32_t rows, int32_t H, int32_t a, void* stream) {
    fbhip_ctx* none = nullptr;
    if (!P || !W4 || !b4 || !premu || rows < 1) { g_err = "fbhip_policy_head: bad argument"; return FBHIP_E_INVALID; }
    if (!policy_head_ok(H, a) || (base != nullptr && !policy_first_ok(H, a, a))) { g_err = "fbhip_policy_head: unsupported (H, a)"; return FBHIP_E_INVALID; }
    HIPCK(none, policy_head_prepare(H, a, a));
    PolicyHeadJobs jobs{};
    jobs.n = 1;
    jobs.j[0] = PolicyHeadJob{P, ldp, premu, noise, mu, action, ld_out, base, ldb, W1a, ldw1, gamma, beta, t1, ldt1, stats};
    HIPCK(none, launch_policy_head(jobs, W4, ldw4, b4, ld_out, a, stddev, clip, ld_out, rows, H, a, a, Squash{0, 1.f, -5.f, 2.f},
                                   (hipStream_t)stream));
    return FBHIP_OK;
}

int fbhip_actor_head_bwd(const float* dt1, int32_t ldt, const float* lnY, const float* lnX, const float* lnStats, const float* lnGamma,
                         const float* W1a, int32_t ldw1, const float* mu, int32_t ldmu, const float* W4, int32_t ldw4, const float* P,
                         float* dpremu, int32_t ldd, float* dp, int32_t rows, int32_t H, int32_t a, void* stream) {
    fbhip_ctx* none = nullptr;
    if (!dt1 || !W1a || !mu || !W4 || !P || !dpremu || !dp || rows < 1 || !actor_head_bwd_ok(H, a)) { g_err = "fbhip_actor_head_bwd: bad argument"; return FBHIP_E_INVALID; }
    HIPCK(none, actor_head_bwd_prepare(H, a));
    HIPCK(none, launch_actor_head_bwd(dt1, ldt, W1a, ldw1, mu, ldmu, W4, ldw4, P, ldt, dpremu, ldd, dp, ldt, rows, H, a, (hipStream_t)stream,
                                      lnY, ldt, lnX, ldt, lnStats, lnGamma));
    return FBHIP_OK;
}

}  // extern "C"

