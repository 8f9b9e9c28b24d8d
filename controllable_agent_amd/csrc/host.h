// Internal declarations of the host side of libfbhip.so, shared by its three translation units:
//   layout.hip    flat-buffer layout of the nets (reference state_dict names / order), dimension checks, workspace carving
//   schedule.hip  the launch schedule of one update: grouped-GEMM policy, round-based merged scheduling, the network passes
//                 as chains of stages, FBDDPGAgent.update / DiscreteFBAgent.update (build_update) and SFAgent.update (build_update_sf)
//   api.hip       the C ABI of include/fbhip.h (context, binding, update entry points, inference, per-kernel test exports)
#pragma once
#include "common.h"
#include "fbhip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace fbhip {
namespace host {

extern thread_local std::string g_err;      // last error of this thread (contexts keep their own copy)

// ------------------------------------------------------------------------------------------------ layout (layout.hip)
struct Slot { std::string name; int64_t off; int rows, cols, ld; };
struct NetLayout {
    std::vector<Slot> slots;            // reference parameters() order
    std::map<std::string, Slot> by_name;
    int64_t numel = 0;                  // padded floats
    int64_t nparams = 0;                // logical parameter count
};

// Internal padding (zero rows / columns that provably stay zero under the update): first-layer
// input widths are padded to a multiple of 32 and the BackwardMap hidden width (526 by default) to a multiple of 64,
// so every GEMM tile of those layers is interior and every K chunk full -> the branch-free loader applies.
inline int pad32(int x) { return (x + 31) & ~31; }
inline int pad64(int x) { return (x + 63) & ~63; }

struct Geom { bool single, trunk, boltz; int Fo, hw, feat; };
Geom geom_of(const fbhip_dims& d);
Geom actor_geom_of(const fbhip_dims& d);
inline int head_width(const fbhip_dims& d) { return d.boltzmann ? 2 * d.action_dim : d.action_dim; }
inline int panel_action_cols(const fbhip_dims& d) { return d.discrete ? 0 : d.action_dim; }
inline int fhead_out(const fbhip_dims& d) { return d.discrete ? d.z_dim * d.action_dim : d.z_dim; }
NetLayout build_layout(const fbhip_dims& d, int net);
int check_dims(const fbhip_dims* d);

// ------------------------------------------------------------------------------------------------ workspace (layout.hip)
struct Buf { float* p = nullptr; int rows = 0, cols = 0, ld = 0; };
struct BSet { Buf pre1, t1, r2, y, Bm; float* stats = nullptr; float* norms = nullptr; };
struct FSet { Buf pre1a, t1a, pre1z, t1z, h, tr, p, F1, F2, Fall1, Fall2; float* statsA = nullptr; float* statsZ = nullptr; };   // Fall: discrete heads' [B, z * A]
struct ASet { Buf pre1o, t1o, pre1z, t1z, h, tr, p, premu, mu; float* statsO = nullptr; float* statsZ = nullptr; };

struct Ws {
    StepState* st = nullptr;
    float* metrics = nullptr;
    SampleOut so{};
    Buf Xoa, Xoz, Xnoz, Xnoa, Xopi, Xo, next_goal, bin, fgoal, z, zrand;
    float* disc = nullptr;
    ASet asT;                // actor(next_obs) of the target chain: its own set, so that the actor's pass on obs can share its rounds
    BSet bsA, bsO, bsM, bsF; // target / online passes on next_goal, z-mix pass on backward_input[perm], hindsight pass
    FSet fsT, fsO;
    ASet as;
    Buf dFall1, dFall2;      // discrete: dF scattered back to the [B, z * A] head outputs
    float* act_idx = nullptr;           // discrete: the sampled transitions' action indices [B] (as stored: floats)
    float* nextq = nullptr;             // discrete: next_Q [B] (discrete_fb.py:297, :302), read by the q_loss
    int32_t* greedy = nullptr;          // discrete: arg-max action per row of the last selection
    Buf dF1, dF2, dBm, dy, dp, dtr, dh, dt1a, dt1z, b_dr2, b_dt1, a_dpremu, a_dp, a_dact, cov, inv_cov, BinvC;
    float* ln_partials = nullptr;
    float* ln_partials_b = nullptr;     // backward_net's own scratch: its backward runs concurrently with forward_net's
    float* splitk = nullptr;            // split-K partial slab
    float* pw_scratch = nullptr;
    float* rw = nullptr;                // rand_weight: [B, B] mixing weights, [B] row scales, and the mixed rows
    float* rw_u = nullptr;
    Buf ymixw;
    // SFAgent (dims.sf): the feature pass runs on 2 batch rows -- [goal ; next_goal] -- so that one backward sums both uses
    Buf goal2;                          // [2B, g]: bin = rows [0, B), next_goal = rows [B, 2B)
    Buf pgoal;                          // [B, g]: next_goal[perm], input of the z-mix (mix_ratio > 0)
    BSet bsS;                           // feature_net activations, 2B rows
    Buf dBm2, dy2, s_dr2, s_dt1;        // its gradient panels, 2B rows
    Buf Xga;                            // svd_p: [goal | action] input of mu_net, B rows
                                        // (mu_net's activations live in bsM, the z-mix set SFAgent does not use: mu = bsM.y)
    Buf dmu, m_dr2, m_dt1;              // its gradient panels
    float* ln_partials_m = nullptr;     // its LayerNorm-backward partial sums (it shares rounds with feature_net's backward)
    float* c99 = nullptr;               // svd_sr: the constant 0.99 "discount" of its target term, one per row
    Buf dmu_y;                          // contrastive: d(mu_net's pre-projection output)
    Buf dphi_o;                         // svd_sr: the orthonormality share of d phi (a second pairwise launch), added to the first
    Buf icat, ih1, ih2, ipre, d_ipre, d_ih1, d_ih2;          // icm: inverse-dynamics activations / gradients
    Buf zeroF, lapS1, lapS2;                                 // lap: the zero F panel and two throw-away dF panels of the pairwise pass
    float* act_in = nullptr;            // batch-1 fast path: [obs | z | 0.. | noise] / [goal | 0.. | z] as staged by the host
    float* act_vec = nullptr;           // its activation vectors
    float* act_out = nullptr;           // action (a floats) or the z correlation (1 float)
    size_t total_bytes = 0;
};
size_t act_noise_off(const fbhip_dims& d);
size_t act_z_off(const fbhip_dims& d);
size_t act_in_floats(const fbhip_dims& d);
Ws carve(const fbhip_dims& d, void* base);

// ------------------------------------------------------------------------------------------------ weights (layout.hip)
struct TrunkP { float *W1, *b1, *g1, *be1, *W2, *b2; int k1, ld1; };
struct FwdP { TrunkP oa, oz; float *Wt = nullptr, *bt = nullptr; float *W3s, *b3s, *W4[2], *b4[2]; };   // Wt: add_trunk
struct BwdP { float *W1, *b1, *g1, *be1, *W2, *b2, *W3, *b3; };
struct IcmP { float *W1 = nullptr, *b1, *W2, *b2, *W3, *b3; };     // the feature learner's head mlp (in the backward segment)
// dims.sf -> the head mlp(in, Hb, 'irelu', Hb, 'irelu', out) that the feature learner trains feature_net through (sf.py):
//   1 icm          inverse_dynamic_net   in = 2 z  (cat[phi, next_phi])   out = a (tanh), target = action        :194-213
//   4 autoencoder  decoder               in = z    (phi)                  out = g,        target = goal          :249-262
//   5 transition   forward_dynamic_net   in = z + a (cat[phi, action])    out = g,        target = next_goal     :215-227
//   7 latent       forward_dynamic_net   in = z + a (cat[phi, action])    out = z,        target = target_feature_net(next_goal) :230-246
//                  (target_feature_net = the backward segment of the TARGET buffer: own init, moved at 0.01 before phi_opt.step())
// (2 lap and 3 random have none)
bool sf_head_dims(const fbhip_dims& d, int* in, int* out, const char** prefix);
struct ActP { TrunkP o, oz; float *Wt = nullptr, *bt = nullptr; float *W3, *b3, *W4, *b4; };
FwdP fwd_p(float* base, const NetLayout& L);
BwdP bwd_p(float* base, const NetLayout& L);
IcmP icm_p(float* base, const NetLayout& L);
BwdP mu_p(float* base, const NetLayout& L);               // svd_p's mu_net (same module structure as BackwardMap.B, no projection)
ActP act_p(float* base, const NetLayout& L);

struct GraphEntry { int mask; fbhip_hparams hp; bool has_inj; fbhip_inject inj; hipGraphExec_t exec; int n_steps; int set;
                    bool branches = false;                 // parallel branches: launched through launch_graph's high-priority stream (api.hip)
                    std::vector<fbhip_inject> injs; };     // multi-step injected graphs: every step's struct is part of the cache key
struct InferGraph { int kind; int eval_mode; int has_noise; float stddev; hipGraphExec_t exec; };

}  // namespace host
}  // namespace fbhip

struct fbhip_ctx {
    fbhip_dims d;
    fbhip::host::NetLayout L[3];
    float *fb_p = nullptr, *fb_g = nullptr, *fb_m = nullptr, *fb_v = nullptr, *fb_t = nullptr;
    float *a_p = nullptr, *a_g = nullptr, *a_m = nullptr, *a_v = nullptr;
    bool bound = false, replay_bound = false;
    fbhip::host::Ws sets[2];                              // two complete workspace sets: fbhip_update_many alternates them so that step
    int cur = 0;                             // t+1's sampling and online forward passes can run beside step t's actor phase
    fbhip::host::Ws& W() { return sets[cur]; }            // the set kernels are currently enqueued on
    const char* ws_lo = nullptr;
    size_t ws_bytes = 0;
    hipStream_t side = nullptr;              // second capture branch of fbhip_update_many
    std::vector<hipEvent_t> events;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;   // bridge events of launch_graph (api.hip)
    hipEvent_t ev_gate = nullptr, ev_gate_in = nullptr;   // fbhip_order_legacy_stream_after / fbhip_order_stream_after_legacy
    hipStream_t last_stream = nullptr;      // the stream of the last update call (fbhip_destroy asks it whether a capture is open)
    hipEvent_t v_ready = nullptr;            // set while the actor phase of a pipelined graph is being built: V comes from the side branch
    fbhip::ReplayView rv{};
    uint64_t seed = 0;
    uint32_t rank = 0;
    fbhip::host::FwdP F_p, F_g, F_t;
    fbhip::host::BwdP K_p, K_g, K_t;
    fbhip::host::BwdP M_p, M_g, M_t;                      // dims.sf == 6 / 8 (svd_p / svd_sr): mu_net (M_t: svd_sr's target_mu_net)
    fbhip::host::IcmP I_p, I_g;                           // dims.sf == 1
    fbhip::host::ActP A_p, A_g;
    std::vector<fbhip::host::GraphEntry> graphs;
    int64_t graph_captures = 0;              // update graphs captured + instantiated so far (fbhip_graph_captures)
    std::vector<fbhip::host::InferGraph> infer_graphs;    // batch-1 fast path (fbhip_act / fbhip_z_correl)
    float* h_metrics = nullptr;              // pinned: FBHIP_NUM_METRICS floats + the sequence number the publish kernel writes last
    unsigned int* d_pubseq = nullptr;        // device: [0] number of publishes so far (advanced in-graph), [1] extra_metrics ticket
    double* d_xm_part = nullptr;             // device: partial sums of extra_metrics_wide_kernel
    unsigned int infer_issued = 0;           // host: batch-1 results requested through the direct-to-host path (d_pubseq[2] counts them)
    bool infer_direct = false;               // act / z_correl write their result into h_out themselves (no D2H node, no stream sync)
    unsigned int pub_issued = 0;             // host: number of publishes ENQUEUED so far (graph replays included)
    float* h_in = nullptr;                   // pinned host staging, same layout as w.act_in
    float* h_out = nullptr;                  // pinned: action / correlation
    const float* gb_panels = nullptr;        // global-batch data parallel (fbhip_bind_global_batch): [6][gb_rows][Lz]
    const float* gb_discount = nullptr;      //   F1, F2, B, tF1, tF2, tB of ALL ranks' rows, and their discounts [gb_rows]
    int gb_rows = 0, gb_off = 0;             //   this rank owns rows [gb_off, gb_off + batch)
    fbhip::PeerComm peers{};                        // fbhip_dp_bind_peers (world >= 2: bound)
    void* rccl_comm = nullptr;                      // fbhip_rccl_init: the library's own RCCL communicator (rccl.hip)
    int rccl_world = 0, rccl_rank = 0;
    fbhip::Squash sq{0, 1.f, -5.f, 2.f};            // boltzmann: temp, log_std_bounds (fb_ddpg.py:70-71); fbhip_set_policy_squash
    std::function<int(const fbhip::PolicyHeadJobs&, hipStream_t)> run_policy_heads;   // set by the update that declares Ops::ph
    fbhip::ColReduceJobs cr_pending{};              // LayerNorm column reduces waiting for the next split-K reduce launch (flush_round)
    std::string err;
};

#define HIPCK(ctx, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) {                                                                           \
            char buf__[512];                                                                               \
            snprintf(buf__, sizeof(buf__), "fbhip: %s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                     __FILE__, __LINE__);                                                                  \
            fbhip::host::g_err = buf__;                                                                                 \
            if (ctx) (ctx)->err = buf__;                                                                   \
            return FBHIP_E_HIP;                                                                            \
        }                                                                                                  \
    } while (0)

#define RC(expr)                          \
    do {                                  \
        int rc__ = (expr);                \
        if (rc__ != FBHIP_OK) return rc__; \
    } while (0)

namespace fbhip {
namespace host {

// ------------------------------------------------------------------------------------------------ schedule (schedule.hip)
GemmProblem P(const float* A, int lda, int akc, const float* B, int ldb, int bkc, float* C, int ldc, int M, int N,
              int K, const float* bias = nullptr, int epi = EPI_NONE, const float* aux = nullptr, int ldaux = 0,
              float* colsum = nullptr);
constexpr size_t SPLITK_SLAB_FLOATS = (size_t)6 << 20;     // 24 MiB
int run_gemms(fbhip_ctx* ctx, std::vector<GemmProblem> v, hipStream_t s);

struct Ops {
    std::vector<GemmProblem> gemms;
    std::vector<LnFwdProblem> lnf;
    std::vector<LnBwdProblem> lnb;
    std::vector<HeadProblem> heads;           // embedding heads (+ projection) of this round that head_kernel can run (fused.hip)
    std::vector<L2Problem> l2n;               // run after this round's GEMMs
    std::vector<PolicyHeadJob> ph;            // fused policy heads of this round (one launch for all of them)
    std::vector<DiscreteHeadJob> dh;          // discrete: the selection / gather row jobs of this round (one launch)
    int dh_rows = 0, dh_ldz = 0;
    std::vector<std::function<int(hipStream_t)>> post;
};
using Stage = std::function<void(Ops&)>;
using Chain = std::vector<Stage>;
using Round = std::vector<Stage>;
using Program = std::vector<Round>;
int run_chain(fbhip_ctx* c, Chain& ch, hipStream_t s);
int run_program(fbhip_ctx* c, Program& p, hipStream_t s);

void forward_map_fwd_chain(fbhip_ctx* c, const FwdP& W, const float* Xa, int lda, const float* Xz, int ldz, FSet& S,
                           int rows, Chain& out, bool with_heads = true, int disc_mode = 0, const float* disc_z = nullptr,
                           int disc_ldz = 0, int oa_base_k = 0, int part = 0);
// oa_base_k > 0: the obs_action trunk's first layer is finished by the policy-head kernel (PolicyHeadJob::base): the chain's first
// round holds only its action-free part (K = oa_base_k input columns), its LayerNorm round only the other trunk.
// part 1: only what does NOT depend on the action (that first round, the obs_z trunk up to its half of h: three rounds);
// part 2: only what does (the obs_action trunk's second layer onward), for a caller that ran part 1 earlier
int forward_map_fwd(fbhip_ctx* c, const FwdP& W, const float* Xa, int lda, const float* Xz, int ldz, FSet& S, int rows, hipStream_t s);
void backward_map_fwd_chain(fbhip_ctx* c, const BwdP& W, const float* X, int ldx, BSet& S, int rows, Chain& out,
                            bool with_projection = true, int in_dim = -1);     // in_dim: input width if not goal_dim (mu_net)
int backward_map_fwd(fbhip_ctx* c, const BwdP& W, const float* X, int ldx, BSet& S, int rows, hipStream_t s);
void actor_fwd_chain(fbhip_ctx* c, const ActP& W, const float* Xo, int ldo, const float* Xz, int ldz, ASet& S, int rows,
                     Chain& out, bool with_head = true);
int actor_fwd(fbhip_ctx* c, const ActP& W, const float* Xo, int ldo, const float* Xz, int ldz, ASet& S, int rows, hipStream_t s);

// selected phases of one update() onto a stream (FBDDPGAgent / DiscreteFBAgent: build_update; SFAgent: build_update_sf)
int enqueue_update(fbhip_ctx* c, const fbhip_hparams& hp, const fbhip_inject* inj, int mask, hipStream_t s);
int enqueue_actor_v(fbhip_ctx* c, hipStream_t s);
int check_hparams(fbhip_ctx* c, const fbhip_hparams* hp);
int need_bound(fbhip_ctx* c, bool replay);
// the library's own RCCL transport (rccl.hip)
int rccl_load(const char* path);
int rccl_unique_id(void* out128);
int rccl_version();
int rccl_init(fbhip_ctx* c, const void* id128, int world, int rank, hipStream_t s);
void rccl_release(fbhip_ctx* c, bool abort = false);
int rccl_allreduce(fbhip_ctx* c, int which, hipStream_t s);

}  // namespace host
}  // namespace fbhip
