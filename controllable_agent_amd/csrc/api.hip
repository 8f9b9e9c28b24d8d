// C-ABI of libfbhip.so (include/fbhip.h): layout of the flat parameter buffers, workspace carving, and the
// launch sequence of one FBDDPGAgent.update() (url_benchmark/agent/fb_ddpg.py:427-520) on gfx950.
//
// Memory model: torch owns every byte.  The library computes offsets (fbhip_layout_*, fbhip_workspace_bytes),
// the host allocates flat fp32 tensors and binds them; after that an update is ~100 asynchronous kernel
// launches on the caller's stream with no allocation and no host synchronisation, so the whole step is
// hipGraph-capturable (fbhip_update(use_graph=1) captures once and replays).
#include "common.h"
#include "fbhip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <algorithm>
#include <vector>

using namespace fbhip;

namespace {

thread_local std::string g_err;

// ------------------------------------------------------------------------------------------------ layout
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

struct LayoutBuilder {
    NetLayout L;
    int64_t cur = 0;
    // logical [rows x cols]; physical leading dimension ld (>= cols) and phys_rows (>= rows) allocated
    void mat(const std::string& n, int rows, int cols, int ld = 0, int phys_rows = 0) {
        Slot s{n, cur, rows, cols, ld > 0 ? ld : pad4(cols)};
        cur += (int64_t)(phys_rows > 0 ? phys_rows : rows) * s.ld;
        L.by_name[n] = s;
        L.nparams += (int64_t)rows * cols;
    }
    void vec(const std::string& n, int len, int phys_len = 0) {
        Slot s{n, cur, 1, len, phys_len > 0 ? phys_len : pad4(len)};
        cur += s.ld;
        L.by_name[n] = s;
        L.nparams += len;
    }
    void trunk(const std::string& p, int in, int H, int Fd) {     // mlp(in, H, "ntanh", Fd, "irelu"), fb_modules.py:60-78
        mat(p + ".0.weight", H, in, pad32(in)); vec(p + ".0.bias", H);
        vec(p + ".1.weight", H); vec(p + ".1.bias", H);
        mat(p + ".3.weight", Fd, H); vec(p + ".3.bias", Fd);
    }
    NetLayout finish(const std::vector<std::string>& order) {
        for (const auto& n : order) L.slots.push_back(L.by_name.at(n));
        L.numel = cur;
        return L;
    }
};

// Front-end geometry of ForwardMap / Actor (fb_modules.py:90-103, 165-178).  preprocess (default): TWO LayerNorm branches
// (in -> H -> Fd) whose outputs are concatenated (2 Fd wide), optionally followed by a Linear(2Fd, H) + ReLU trunk
// (add_trunk).  preprocess == 0: ONE LayerNorm branch on the concatenated input with Fd := H, always followed by the
// trunk's last Linear(H, H) + ReLU -- the same pipeline with one branch.
struct Geom { bool single, trunk, boltz; int Fo, hw, feat; };
Geom geom_of(const fbhip_dims& d) {
    Geom g;
    g.boltz = false;
    g.single = d.preprocess == 0 || d.discrete != 0;
    g.trunk = g.single || d.add_trunk != 0;
    g.Fo = g.single ? d.hidden_dim : d.feature_dim;          // a branch's output width
    g.hw = g.single ? d.hidden_dim : 2 * d.feature_dim;      // concatenated branch outputs
    g.feat = g.trunk ? d.hidden_dim : g.hw;                  // what feeds the heads / the policy
    return g;
}
// The actor's own geometry.  boltzmann: DiagGaussianActor (fb_modules.py:129-151) = ONE LayerNorm branch on [obs|z]
// (H -> H, the "policy" mlp's first two Linears) feeding the [loc | raw log-std] head directly: no trunk layer, no
// policy hidden layer; preprocess / add_trunk do not apply to it.
Geom actor_geom_of(const fbhip_dims& d) {
    Geom g = geom_of(d);
    if (d.boltzmann) { g.boltz = true; g.single = true; g.trunk = false; g.Fo = g.hw = g.feat = d.hidden_dim; }
    return g;
}
inline int head_width(const fbhip_dims& d) { return d.boltzmann ? 2 * d.action_dim : d.action_dim; }
// DiscreteFBAgent (dims.discrete, discrete_fb.py:52-101): action_dim is A, the NUMBER of actions.  The ForwardMap has no action
// input (one trunk on [obs|z]) and its heads emit one embedding per action: z * A outputs, (k, a) at column k * A + a.
inline int panel_action_cols(const fbhip_dims& d) { return d.discrete ? 0 : d.action_dim; }
inline int fhead_out(const fbhip_dims& d) { return d.discrete ? d.z_dim * d.action_dim : d.z_dim; }

std::vector<std::string> trunk_names(const std::string& p) {
    return {p + ".0.weight", p + ".0.bias", p + ".1.weight", p + ".1.bias", p + ".3.weight", p + ".3.bias"};
}
void append(std::vector<std::string>& a, const std::vector<std::string>& b) { a.insert(a.end(), b.begin(), b.end()); }

NetLayout build_layout(const fbhip_dims& d, int net) {
    const int o = d.obs_dim, a = d.action_dim, g = d.goal_dim, z = d.z_dim, H = d.hidden_dim, Fd = d.feature_dim,
              Hb = d.backward_hidden_dim;
    LayoutBuilder b;
    std::vector<std::string> order;
    const Geom gm = geom_of(d);
    if (net == FBHIP_NET_FORWARD) {               // ForwardMap, fb_modules.py:165-182
        if (gm.single) {                          // trunk = mlp(o + z + a, H, "ntanh", H, "irelu", H, "irelu")
            b.trunk("trunk", o + z + panel_action_cols(d), H, H);
            b.mat("trunk.5.weight", H, H); b.vec("trunk.5.bias", H);
        } else {
            b.trunk("obs_action_net", o + a, H, Fd);
            b.trunk("obs_z_net", o + z, H, Fd);
            if (d.add_trunk) { b.mat("trunk.0.weight", H, 2 * Fd); b.vec("trunk.0.bias", H); }
        }
        // F1/F2 first layers are stored back to back so both heads run as ONE [2H x feat] GEMM
        b.mat("F1.0.weight", H, gm.feat); b.mat("F2.0.weight", H, gm.feat);
        b.vec("F1.0.bias", H); b.vec("F2.0.bias", H);
        b.mat("F1.2.weight", fhead_out(d), H); b.vec("F1.2.bias", fhead_out(d));
        b.mat("F2.2.weight", fhead_out(d), H); b.vec("F2.2.bias", fhead_out(d));
        if (gm.single) {
            append(order, trunk_names("trunk")); append(order, {"trunk.5.weight", "trunk.5.bias"});
        } else {
            append(order, trunk_names("obs_action_net")); append(order, trunk_names("obs_z_net"));
            if (d.add_trunk) append(order, {"trunk.0.weight", "trunk.0.bias"});
        }
        append(order, {"F1.0.weight", "F1.0.bias", "F1.2.weight", "F1.2.bias",
                       "F2.0.weight", "F2.0.bias", "F2.2.weight", "F2.2.bias"});
    } else if (net == FBHIP_NET_BACKWARD) {       // BackwardMap, fb_modules.py:220
        const int HbP = pad64(Hb);
        // SFAgent (dims.sf): the same architecture is feature_learner.feature_net (sf.py:84-88; the projection is its last module)
        const std::string q = d.sf ? "feature_net." : "B.";
        b.mat(q + "0.weight", Hb, g, pad32(g), HbP); b.vec(q + "0.bias", Hb, HbP); b.vec(q + "1.weight", Hb, HbP);
        b.vec(q + "1.bias", Hb, HbP);
        b.mat(q + "3.weight", Hb, Hb, HbP, HbP); b.vec(q + "3.bias", Hb, HbP);
        b.mat(q + "5.weight", z, Hb, HbP); b.vec(q + "5.bias", z);
        order = {q + "0.weight", q + "0.bias", q + "1.weight", q + "1.bias", q + "3.weight", q + "3.bias", q + "5.weight", q + "5.bias"};
        if (d.sf == 1) {      // ICM: inverse_dynamic_net = mlp(2 z, Hb, 'irelu', Hb, 'irelu', a, 'tanh')  (sf.py:198)
            const std::string i = "inverse_dynamic_net.";
            b.mat(i + "0.weight", Hb, 2 * z, pad32(2 * z), HbP); b.vec(i + "0.bias", Hb, HbP);
            b.mat(i + "2.weight", Hb, Hb, HbP, HbP); b.vec(i + "2.bias", Hb, HbP);
            b.mat(i + "4.weight", a, Hb, HbP); b.vec(i + "4.bias", a);
            append(order, {i + "0.weight", i + "0.bias", i + "2.weight", i + "2.bias", i + "4.weight", i + "4.bias"});
        }
    } else if (d.discrete) {                      // DiscreteFBAgent has no actor: empty layout
    } else if (d.boltzmann) {                     // DiagGaussianActor.policy = mlp(o + z, H, "ntanh", H, "relu", 2a)
        b.trunk("policy", o + z, H, H);
        b.mat("policy.5.weight", 2 * a, H); b.vec("policy.5.bias", 2 * a);
        append(order, trunk_names("policy")); append(order, {"policy.5.weight", "policy.5.bias"});
    } else {                                      // Actor, fb_modules.py:91-105
        if (gm.single) {                          // trunk = mlp(o + z, H, "ntanh", H, "irelu", H, "irelu")
            b.trunk("trunk", o + z, H, H);
            b.mat("trunk.5.weight", H, H); b.vec("trunk.5.bias", H);
            append(order, trunk_names("trunk")); append(order, {"trunk.5.weight", "trunk.5.bias"});
        } else {
            b.trunk("obs_net", o, H, Fd);
            b.trunk("obs_z_net", o + z, H, Fd);
            if (d.add_trunk) { b.mat("trunk.0.weight", H, 2 * Fd); b.vec("trunk.0.bias", H); }
            append(order, trunk_names("obs_net")); append(order, trunk_names("obs_z_net"));
            if (d.add_trunk) append(order, {"trunk.0.weight", "trunk.0.bias"});
        }
        b.mat("policy.0.weight", H, gm.feat); b.vec("policy.0.bias", H);
        b.mat("policy.2.weight", a, H); b.vec("policy.2.bias", a);
        append(order, {"policy.0.weight", "policy.0.bias", "policy.2.weight", "policy.2.bias"});
    }
    return b.finish(order);
}

int check_dims(const fbhip_dims* d) {
    if (!d) return FBHIP_E_INVALID;
    if (d->struct_size != sizeof(fbhip_dims)) {
        g_err = "fbhip: fbhip_dims.struct_size is " + std::to_string(d->struct_size) + ", this library expects " +
                std::to_string(sizeof(fbhip_dims)) + " (caller built against another include/fbhip.h?)";
        return FBHIP_E_INVALID;
    }
    if (d->batch < 2 || d->obs_dim < 1 || d->action_dim < 1 || d->goal_dim < 1 || d->z_dim < 1 ||
        d->hidden_dim < 4 || d->feature_dim < 4 || d->backward_hidden_dim < 1) { g_err = "fbhip: non-positive dimension"; return FBHIP_E_INVALID; }
    if ((d->hidden_dim & 3) || (d->feature_dim & 3)) { g_err = "fbhip: hidden_dim and feature_dim must be multiples of 4"; return FBHIP_E_INVALID; }
    if (d->hidden_dim > 2048 || d->backward_hidden_dim > 2048) { g_err = "fbhip: hidden dims > 2048 unsupported (LayerNorm row kernel)"; return FBHIP_E_INVALID; }
    if (d->z_dim > 128) { g_err = "fbhip: z_dim > 128 unsupported (pairwise kernel)"; return FBHIP_E_INVALID; }
    if (d->action_dim > 64) { g_err = "fbhip: action_dim > 64 unsupported"; return FBHIP_E_INVALID; }
    if (d->batch > 8192) { g_err = "fbhip: batch > 8192 per GPU unsupported (permutation sort)"; return FBHIP_E_INVALID; }
    if (d->discrete && d->preprocess) { g_err = "fbhip: discrete needs preprocess == 0 (the reference's discrete ForwardMap.forward only runs without the preprocess nets, discrete_fb.py:91-94)"; return FBHIP_E_INVALID; }
    if (d->discrete && (int64_t)d->z_dim * d->action_dim > 8192) { g_err = "fbhip: discrete: z_dim * actions > 8192 unsupported"; return FBHIP_E_INVALID; }
    if (d->sf < 0 || d->sf > 2) { g_err = "fbhip: dims.sf must be 0, 1 (icm) or 2 (lap)"; return FBHIP_E_INVALID; }
    if (d->sf && (d->discrete || d->boltzmann || !d->norm_z)) { g_err = "fbhip: dims.sf needs discrete = 0, boltzmann = 0, norm_z = 1"; return FBHIP_E_INVALID; }
    if (!d->use_goal && d->goal_dim != d->obs_dim) { g_err = "fbhip: goal_dim must equal obs_dim when use_goal == 0"; return FBHIP_E_INVALID; }
    return FBHIP_OK;
}

// ------------------------------------------------------------------------------------------------ workspace
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
    BSet bsS;                           // feature_net activations, 2B rows
    Buf dBm2, dy2, s_dr2, s_dt1;        // its gradient panels, 2B rows
    Buf icat, ih1, ih2, ipre, d_ipre, d_ih1, d_ih2;          // icm: inverse-dynamics activations / gradients
    Buf zeroF, lapS1, lapS2;                                 // lap: the zero F panel and two throw-away dF panels of the pairwise pass
    float* act_in = nullptr;            // batch-1 fast path: [obs | z | 0.. | noise] / [goal | 0.. | z] as staged by the host
    float* act_vec = nullptr;           // its activation vectors
    float* act_out = nullptr;           // action (a floats) or the z correlation (1 float)
    size_t total_bytes = 0;
};

struct Carver {
    char* base;
    size_t cur = 0;
    explicit Carver(void* b) : base((char*)b) {}
    void* take(size_t bytes) {
        cur = (cur + 255) & ~(size_t)255;
        void* p = base ? base + cur : nullptr;
        cur += bytes;
        return p;
    }
    float* f(size_t n) { return (float*)take(n * sizeof(float)); }
    Buf buf(int rows, int cols, int ld = 0) {
        Buf b;
        b.rows = rows; b.cols = cols; b.ld = ld > 0 ? ld : pad4(cols);
        b.p = f((size_t)rows * b.ld);
        return b;
    }
};

// staging layout of the batch-1 entry points (host pinned buffer and device copy are identical):
//   act:      [obs (o) | z (d) | zeros up to pad32(o+d) | noise (a)]      compute_z_correl: [goal (g) | zeros up to pad32(g) | z (d)]
size_t act_noise_off(const fbhip_dims& d) { return (size_t)pad32(d.obs_dim + d.z_dim); }
size_t act_z_off(const fbhip_dims& d) { return (size_t)pad32(d.goal_dim); }
size_t act_in_floats(const fbhip_dims& d) {
    const size_t a = act_noise_off(d) + 64, b = act_z_off(d) + (size_t)pad4(d.z_dim);
    return (a > b ? a : b) + 64;
}

Ws carve(const fbhip_dims& d, void* base) {
    Ws w;
    Carver c(base);
    const int B = d.batch, o = d.obs_dim, a = d.action_dim, g = d.goal_dim, z = d.z_dim, H = d.hidden_dim,
              Hb = d.backward_hidden_dim;
    w.st = (StepState*)c.take(sizeof(StepState));
    w.metrics = c.f(FBHIP_NUM_METRICS);
    w.so.ep_idx = (int32_t*)c.take((size_t)B * 4);
    w.so.step_idx = (int32_t*)c.take((size_t)B * 4);
    w.so.perm = (int32_t*)c.take((size_t)B * 4);
    w.so.mix_uniform = c.f(B);
    w.so.z_gauss = c.f((size_t)B * z);
    w.so.eps_next = c.f((size_t)B * a);
    w.so.eps_actor = c.f((size_t)B * a);
    w.so.future_idx = (int32_t*)c.take((size_t)B * 4);
    w.so.z_uniform = c.f((size_t)B * z);
    w.so.future_uniform = c.f(B);
    // input panels: widths padded to 32 (pad columns stay zero: the workspace is zero-initialised by the host and
    // no kernel writes them)
    const Geom gm = geom_of(d);
    if (gm.single) {
        // preprocess == 0: the ForwardMap panels are [obs | z | action].  The actor keeps its own [obs|z] panels: a GEMM
        // runs over the weight's padded width, so whatever follows z in a shared panel would leak into the weight
        // gradient's pad columns and, through Adam, into the pad weights.
        const int wd = o + z + panel_action_cols(d);
        w.Xoa = c.buf(B, wd, pad32(wd)); w.Xnoa = c.buf(B, wd, pad32(wd)); w.Xopi = c.buf(B, wd, pad32(wd));
        w.Xoz = c.buf(B, o + z, pad32(o + z)); w.Xnoz = c.buf(B, o + z, pad32(o + z));
    } else {
        w.Xoa = c.buf(B, o + a, pad32(o + a)); w.Xoz = c.buf(B, o + z, pad32(o + z)); w.Xnoz = c.buf(B, o + z, pad32(o + z));
        w.Xnoa = c.buf(B, o + a, pad32(o + a)); w.Xopi = c.buf(B, o + a, pad32(o + a));
    }
    w.Xo = c.buf(B, o, pad32(o));
    if (d.sf) {
        w.goal2 = c.buf(2 * B, g, pad32(g));
        w.bin = w.goal2; w.bin.rows = B;
        w.next_goal = w.bin; w.next_goal.p = base ? w.goal2.p + (size_t)B * w.goal2.ld : nullptr;
        w.fgoal = c.buf(1, g, pad32(g));
    } else {
        w.next_goal = c.buf(B, g, pad32(g)); w.bin = c.buf(B, g, pad32(g)); w.fgoal = c.buf(B, g, pad32(g));
    }
    w.z = c.buf(B, z); w.zrand = c.buf(B, z);
    w.disc = c.f(B);
    for (BSet* s : {&w.bsA, &w.bsO, &w.bsM, &w.bsF}) {
        s->pre1 = c.buf(B, Hb, pad64(Hb)); s->t1 = c.buf(B, Hb, pad64(Hb)); s->r2 = c.buf(B, Hb, pad64(Hb));
        s->y = c.buf(B, z); s->Bm = c.buf(B, z);
        s->stats = c.f(2 * (size_t)B); s->norms = c.f(B);
    }
    for (FSet* s : {&w.fsT, &w.fsO}) {
        s->pre1a = c.buf(B, H); s->t1a = c.buf(B, H); s->pre1z = c.buf(B, H); s->t1z = c.buf(B, H);
        s->h = c.buf(B, gm.hw); s->tr = c.buf(gm.trunk ? B : 1, H); s->p = c.buf(B, 2 * H); s->F1 = c.buf(B, z); s->F2 = c.buf(B, z);
        s->Fall1 = c.buf(d.discrete ? B : 1, fhead_out(d)); s->Fall2 = c.buf(d.discrete ? B : 1, fhead_out(d));
        s->statsA = c.f(2 * (size_t)B); s->statsZ = c.f(2 * (size_t)B);
    }
    w.as.pre1o = c.buf(B, H); w.as.t1o = c.buf(B, H); w.as.pre1z = c.buf(B, H); w.as.t1z = c.buf(B, H);
    const Geom ga = actor_geom_of(d);
    const int Na = head_width(d);
    w.as.h = c.buf(B, ga.hw); w.as.tr = c.buf(ga.trunk ? B : 1, H); w.as.p = c.buf(B, H); w.as.premu = c.buf(B, Na); w.as.mu = c.buf(B, a);
    w.as.statsO = c.f(2 * (size_t)B); w.as.statsZ = c.f(2 * (size_t)B);
    w.asT.pre1o = c.buf(B, H); w.asT.t1o = c.buf(B, H); w.asT.pre1z = c.buf(B, H); w.asT.t1z = c.buf(B, H);
    w.asT.h = c.buf(B, ga.hw); w.asT.tr = c.buf(ga.trunk ? B : 1, H); w.asT.p = c.buf(B, H); w.asT.premu = c.buf(B, Na); w.asT.mu = c.buf(B, a);
    w.asT.statsO = c.f(2 * (size_t)B); w.asT.statsZ = c.f(2 * (size_t)B);
    w.dF1 = c.buf(B, z); w.dF2 = c.buf(B, z); w.dBm = c.buf(B, z); w.dy = c.buf(B, z);
    w.dFall1 = c.buf(d.discrete ? B : 1, fhead_out(d)); w.dFall2 = c.buf(d.discrete ? B : 1, fhead_out(d));
    w.act_idx = c.f(B); w.nextq = c.f(B); w.greedy = (int32_t*)c.take((size_t)B * 4);
    w.dp = c.buf(B, 2 * H); w.dtr = c.buf(gm.trunk ? B : 1, H); w.dh = c.buf(B, gm.hw > ga.hw ? gm.hw : ga.hw); w.dt1a = c.buf(B, H); w.dt1z = c.buf(B, H);
    w.b_dr2 = c.buf(B, Hb, pad64(Hb)); w.b_dt1 = c.buf(B, Hb, pad64(Hb)); w.a_dpremu = c.buf(B, Na); w.a_dp = c.buf(B, H);
    w.a_dact = c.buf(B, a);
    w.cov = c.buf(z, z); w.inv_cov = c.buf(z, z); w.BinvC = c.buf(B, z);
    const int nmax = H > Hb ? H : Hb;
    w.ln_partials = c.f((size_t)2 * ((B + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK) * 2 * nmax);   // two trunks
    w.ln_partials_b = c.f((size_t)(((d.sf ? 2 : 1) * B + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK) * 2 * nmax);
    w.pw_scratch = c.f(pairwise_scratch_floats(B, z));
    w.splitk = c.f((size_t)6 << 20);
    w.rw = c.f((size_t)B * B); w.rw_u = c.f(B); w.ymixw = c.buf(B, z);
    if (d.sf) {
        const int Lb = pad64(Hb), Lz = pad4(z), La = pad4(a);
        w.bsS.pre1 = c.buf(2 * B, Hb, Lb); w.bsS.t1 = c.buf(2 * B, Hb, Lb); w.bsS.r2 = c.buf(2 * B, Hb, Lb);
        w.bsS.y = c.buf(2 * B, z); w.bsS.Bm = c.buf(2 * B, z);
        w.bsS.stats = c.f(4 * (size_t)B); w.bsS.norms = c.f(2 * (size_t)B);
        w.dBm2 = c.buf(2 * B, z); w.dy2 = c.buf(2 * B, z); w.s_dr2 = c.buf(2 * B, Hb, Lb); w.s_dt1 = c.buf(2 * B, Hb, Lb);
        if (d.sf == 2) { w.zeroF = c.buf(B, z); w.lapS1 = c.buf(B, z); w.lapS2 = c.buf(B, z); }
        if (d.sf == 1) {
            w.icat = c.buf(B, 2 * z, pad32(2 * z)); w.ih1 = c.buf(B, Hb, Lb); w.ih2 = c.buf(B, Hb, Lb);
            w.ipre = c.buf(B, a, La); w.d_ipre = c.buf(B, a, La); w.d_ih1 = c.buf(B, Hb, Lb); w.d_ih2 = c.buf(B, Hb, Lb);
        }
        (void)Lz;
    }
    w.act_in = c.f(act_in_floats(d));
    w.act_vec = c.f((size_t)5 * 2048 + 256);
    w.act_out = c.f(64);
    w.total_bytes = (c.cur + 255) & ~(size_t)255;
    return w;
}

// ------------------------------------------------------------------------------------------------ weights
struct TrunkP { float *W1, *b1, *g1, *be1, *W2, *b2; int k1, ld1; };
struct FwdP { TrunkP oa, oz; float *Wt = nullptr, *bt = nullptr; float *W3s, *b3s, *W4[2], *b4[2]; };   // Wt: add_trunk
struct BwdP { float *W1, *b1, *g1, *be1, *W2, *b2, *W3, *b3; };
struct IcmP { float *W1 = nullptr, *b1, *W2, *b2, *W3, *b3; };     // SFAgent's inverse_dynamic_net (in the backward segment)
struct ActP { TrunkP o, oz; float *Wt = nullptr, *bt = nullptr; float *W3, *b3, *W4, *b4; };

TrunkP trunk_p(float* base, const NetLayout& L, const std::string& p) {
    const Slot& w1 = L.by_name.at(p + ".0.weight");
    TrunkP t;
    t.W1 = base + w1.off; t.k1 = w1.cols; t.ld1 = w1.ld;
    t.b1 = base + L.by_name.at(p + ".0.bias").off;
    t.g1 = base + L.by_name.at(p + ".1.weight").off;
    t.be1 = base + L.by_name.at(p + ".1.bias").off;
    t.W2 = base + L.by_name.at(p + ".3.weight").off;
    t.b2 = base + L.by_name.at(p + ".3.bias").off;
    return t;
}
FwdP fwd_p(float* base, const NetLayout& L) {
    FwdP f;
    if (L.by_name.count("trunk.5.weight")) {                 // preprocess == 0: the one branch + the trunk's last layer
        f.oa = trunk_p(base, L, "trunk"); f.oz = f.oa;
        f.Wt = base + L.by_name.at("trunk.5.weight").off; f.bt = base + L.by_name.at("trunk.5.bias").off;
    } else {
        f.oa = trunk_p(base, L, "obs_action_net"); f.oz = trunk_p(base, L, "obs_z_net");
        if (L.by_name.count("trunk.0.weight")) { f.Wt = base + L.by_name.at("trunk.0.weight").off; f.bt = base + L.by_name.at("trunk.0.bias").off; }
    }
    f.W3s = base + L.by_name.at("F1.0.weight").off; f.b3s = base + L.by_name.at("F1.0.bias").off;
    f.W4[0] = base + L.by_name.at("F1.2.weight").off; f.b4[0] = base + L.by_name.at("F1.2.bias").off;
    f.W4[1] = base + L.by_name.at("F2.2.weight").off; f.b4[1] = base + L.by_name.at("F2.2.bias").off;
    return f;
}
BwdP bwd_p(float* base, const NetLayout& L) {
    BwdP b;
    const std::string q = L.by_name.count("B.0.weight") ? "B." : "feature_net.";
    b.W1 = base + L.by_name.at(q + "0.weight").off; b.b1 = base + L.by_name.at(q + "0.bias").off;
    b.g1 = base + L.by_name.at(q + "1.weight").off; b.be1 = base + L.by_name.at(q + "1.bias").off;
    b.W2 = base + L.by_name.at(q + "3.weight").off; b.b2 = base + L.by_name.at(q + "3.bias").off;
    b.W3 = base + L.by_name.at(q + "5.weight").off; b.b3 = base + L.by_name.at(q + "5.bias").off;
    return b;
}
IcmP icm_p(float* base, const NetLayout& L) {
    IcmP i;
    if (!L.by_name.count("inverse_dynamic_net.0.weight")) return i;
    const std::string q = "inverse_dynamic_net.";
    i.W1 = base + L.by_name.at(q + "0.weight").off; i.b1 = base + L.by_name.at(q + "0.bias").off;
    i.W2 = base + L.by_name.at(q + "2.weight").off; i.b2 = base + L.by_name.at(q + "2.bias").off;
    i.W3 = base + L.by_name.at(q + "4.weight").off; i.b3 = base + L.by_name.at(q + "4.bias").off;
    return i;
}
ActP act_p(float* base, const NetLayout& L) {
    ActP a;
    if (L.by_name.count("policy.5.weight")) {                // DiagGaussianActor
        a.o = trunk_p(base, L, "policy"); a.oz = a.o;
        a.W3 = a.b3 = nullptr;
        a.W4 = base + L.by_name.at("policy.5.weight").off; a.b4 = base + L.by_name.at("policy.5.bias").off;
        return a;
    }
    if (L.by_name.count("trunk.5.weight")) {
        a.o = trunk_p(base, L, "trunk"); a.oz = a.o;
        a.Wt = base + L.by_name.at("trunk.5.weight").off; a.bt = base + L.by_name.at("trunk.5.bias").off;
    } else {
        a.o = trunk_p(base, L, "obs_net"); a.oz = trunk_p(base, L, "obs_z_net");
        if (L.by_name.count("trunk.0.weight")) { a.Wt = base + L.by_name.at("trunk.0.weight").off; a.bt = base + L.by_name.at("trunk.0.bias").off; }
    }
    a.W3 = base + L.by_name.at("policy.0.weight").off; a.b3 = base + L.by_name.at("policy.0.bias").off;
    a.W4 = base + L.by_name.at("policy.2.weight").off; a.b4 = base + L.by_name.at("policy.2.bias").off;
    return a;
}

struct GraphEntry { int mask; fbhip_hparams hp; bool has_inj; fbhip_inject inj; hipGraphExec_t exec; int n_steps; int set; };
struct InferGraph { int kind; int eval_mode; int has_noise; float stddev; hipGraphExec_t exec; };

}  // namespace

struct fbhip_ctx {
    fbhip_dims d;
    NetLayout L[3];
    float *fb_p = nullptr, *fb_g = nullptr, *fb_m = nullptr, *fb_v = nullptr, *fb_t = nullptr;
    float *a_p = nullptr, *a_g = nullptr, *a_m = nullptr, *a_v = nullptr;
    bool bound = false, replay_bound = false;
    Ws sets[2];                              // two complete workspace sets: fbhip_update_many alternates them so that step
    int cur = 0;                             // t+1's sampling and online forward passes can run beside step t's actor phase
    Ws& W() { return sets[cur]; }            // the set kernels are currently enqueued on
    const char* ws_lo = nullptr;
    size_t ws_bytes = 0;
    hipStream_t side = nullptr;              // second capture branch of fbhip_update_many
    std::vector<hipEvent_t> events;
    ReplayView rv{};
    uint64_t seed = 0;
    uint32_t rank = 0;
    FwdP F_p, F_g, F_t;
    BwdP K_p, K_g, K_t;
    IcmP I_p, I_g;                           // dims.sf == 1
    ActP A_p, A_g;
    std::vector<GraphEntry> graphs;
    std::vector<InferGraph> infer_graphs;    // batch-1 fast path (fbhip_act / fbhip_z_correl)
    float* h_in = nullptr;                   // pinned host staging, same layout as w.act_in
    float* h_out = nullptr;                  // pinned: action / correlation
    const float* gb_panels = nullptr;        // global-batch data parallel (fbhip_bind_global_batch): [6][gb_rows][Lz]
    const float* gb_discount = nullptr;      //   F1, F2, B, tF1, tF2, tB of ALL ranks' rows, and their discounts [gb_rows]
    int gb_rows = 0, gb_off = 0;             //   this rank owns rows [gb_off, gb_off + batch)
    PeerComm peers{};                        // fbhip_dp_bind_peers (world >= 2: bound)
    Squash sq{0, 1.f, -5.f, 2.f};            // boltzmann: temp, log_std_bounds (fb_ddpg.py:70-71); fbhip_set_policy_squash
    std::function<int(const PolicyHeadJobs&, hipStream_t)> run_policy_heads;   // set by the update that declares Ops::ph
    ColReduceJobs cr_pending{};              // LayerNorm column reduces waiting for the next split-K reduce launch (flush_round)
    std::string err;
};

namespace {

#define HIPCK(ctx, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) {                                                                           \
            char buf__[512];                                                                               \
            snprintf(buf__, sizeof(buf__), "fbhip: %s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                     __FILE__, __LINE__);                                                                  \
            g_err = buf__;                                                                                 \
            if (ctx) (ctx)->err = buf__;                                                                   \
            return FBHIP_E_HIP;                                                                            \
        }                                                                                                  \
    } while (0)

#define RC(expr)                          \
    do {                                  \
        int rc__ = (expr);                \
        if (rc__ != FBHIP_OK) return rc__; \
    } while (0)

GemmProblem P(const float* A, int lda, int akc, const float* B, int ldb, int bkc, float* C, int ldc, int M, int N,
              int K, const float* bias = nullptr, int epi = EPI_NONE, const float* aux = nullptr, int ldaux = 0,
              float* colsum = nullptr) {
    GemmProblem p{};
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.aux = aux; p.colsum = colsum;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
    p.a_kcontig = akc; p.b_kcontig = bkc; p.epi = epi;
    return p;
}

// scratch for split-K partials (launches of one update are stream-ordered, so one slab serves them all); standalone
// fbhip_gemm (ctx == nullptr) never splits
constexpr size_t SPLITK_SLAB_FLOATS = (size_t)6 << 20;     // 24 MiB

float* splitk_slab(fbhip_ctx* c) { return (c && c->W().splitk) ? c->W().splitk : nullptr; }

int run_gemms(fbhip_ctx* ctx, std::vector<GemmProblem> v, hipStream_t s) {
    long tiles32 = 0;
    int kmax = 0, nmax = 0, mmax = 0;
    for (auto& p : v) {
        tiles32 += (long)((p.M + 31) / 32) * ((p.N + 31) / 32);
        kmax = p.K > kmax ? p.K : kmax; nmax = p.N > nmax ? p.N : nmax; mmax = p.M > mmax ? p.M : mmax;
    }
    bool dma_all = true;
    long tiles128 = 0;
    for (auto& p : v) {
        dma_all = dma_all && gemm_problem_dma_ok(p);
        tiles128 += (long)((p.M + 127) / 128) * ((p.N + 63) / 64);
    }
    int cfg;
    // >= 8 128x64 tiles per CU (2048; measured: at 4-6 per CU, quadruped B = 2048, it still loses 1 % to the 64x64 kernel):
    // the LDS-DMA kernel with two accumulators per wave (25 % fewer operand bytes per FLOP
    // through the per-CU global->LDS path; 132 vs 122 TFLOP/s at 4096^3).  The step's own launches have 1-2 tiles per CU
    // and measure faster on the register-staged 64x64 kernel (see gemm.hip).
    static const long dma128_min = [] { const char* e = getenv("FBHIP_DMA128_MIN_TILES"); return e ? atol(e) : 2048L; }();
    if (dma_all && kmax > 64 && tiles128 >= dma128_min) cfg = CFG_DMA128;
    else if (kmax <= 64) cfg = (nmax <= 32) ? CFG_4x1x1 : CFG_2x2x1;
    // aim for >= 2 workgroups per CU (>= 512): a lone wave per SIMD cannot hide LDS / L2 latency behind its one
    // dependent MFMA chain, so medium outputs split K inside the workgroup instead of using bigger tiles
    else if (tiles32 >= 2048) cfg = CFG_2x2x1;
    else if (tiles32 >= 1024) cfg = (nmax > mmax) ? CFG_1x2x2 : CFG_2x1x2;
    else cfg = CFG_1x1x4;
    const int bkt = gemm_cfg_bkt(cfg);
    float* slab = splitk_slab(ctx);
    // per-workgroup cost of a problem in K-chunk units; unaligned operands take the predicated loader (~3x per chunk)
    auto cost_of = [&](const GemmProblem& p) {
        const bool vec = (((uintptr_t)p.A & 15) == 0 && (p.lda & 3) == 0) && (((uintptr_t)p.B & 15) == 0 && (p.ldb & 3) == 0);
        return (long)((p.K + bkt - 1) / bkt) * (vec ? 1 : 3);
    };
    // longest workgroups first, so that the stragglers of a heterogeneous group start at t = 0 and hide under the
    // bulk (the hardware dispatches workgroups in launch order)
    std::stable_sort(v.begin(), v.end(), [&](const GemmProblem& a, const GemmProblem& b) { return cost_of(a) > cost_of(b); });
    size_t i = 0;
    while (i < v.size()) {
        GemmGroup g{};
        int start = 0, red = 0;
        size_t slab_used = 0;
        // workgroups / total work of this launch without K slicing
        long base_blocks = 0, work = 0;
        for (size_t j = i; j < v.size() && j < i + MAX_GROUP; ++j) {
            GemmProblem q = v[j];
            q.kslices = 1;
            gemm_problem_finalize(q, cfg);
            base_blocks += (long)q.tiles_m * q.tiles_n;
            work += (long)q.tiles_m * q.tiles_n * cost_of(q);
        }
        const long ideal = (work + 511) / 512;          // chunk-units per slot with ~2 workgroups on every CU
        while (i < v.size() && g.n < MAX_GROUP) {
            GemmProblem p = v[i++];
            p.kslices = 1;
            gemm_problem_finalize(p, cfg);
            const int kchunks = (p.K + bkt - 1) / bkt;
            int want = 1;
            static const long small_split_max = [] { const char* e = getenv("FBHIP_SMALL_SPLIT_MAX_BLOCKS"); return e ? atol(e) : 160L; }();
            if (base_blocks <= small_split_max && kchunks >= 4) {
                // small launch: slice K across workgroups until it has ~3 workgroups per CU
                want = (int)((640 + base_blocks - 1) / base_blocks);
                if (want > kchunks / 2) want = kchunks / 2;
            } else if (base_blocks > small_split_max && cost_of(p) > std::max(ideal, 6L) && kchunks >= 4) {
                // straggler of a heterogeneous group: slice until one workgroup costs about half the ideal makespan
                const long pen = cost_of(p) / kchunks;
                long kper = std::max(2L, (ideal / 2 + pen - 1) / pen);
                want = (int)((kchunks + kper - 1) / kper);
            }
            if (slab && want > 1) {
                const int kper = (kchunks + want - 1) / want;
                const int ks = (kchunks + kper - 1) / kper;
                const size_t need = (size_t)ks * p.M * p.N + (size_t)ks * p.M;
                if (ks > 1 && slab_used + need <= SPLITK_SLAB_FLOATS) {
                    p.kslices = ks; p.kper = kper; p.partial = slab + slab_used; p.red_start = red;
                    slab_used += (need + 3) & ~(size_t)3;
                    red += p.M * p.N + p.M;
                }
            }
            p.tile_start = start;
            start += p.tiles_m * p.tiles_n * p.kslices;
            g.p[g.n++] = p;
        }
        g.total_tiles = start;
        static const bool log_launches = [] { const char* e = getenv("FBHIP_GEMM_LOG"); return e && e[0] == '1'; }();
        if (log_launches) {                      // tools/gemm_launch_report.py joins these lines with a kernel trace
            double fl = 0;
            for (int q = 0; q < g.n; ++q) fl += 2.0 * g.p[q].M * g.p[q].N * g.p[q].K;
            fprintf(stderr, "GEMMLOG cfg=%d wgs=%d gflop=%.4f reduce=%d :", cfg, start, fl * 1e-9, red > 0 ? 1 : 0);
            for (int q = 0; q < g.n; ++q) fprintf(stderr, " %dx%dx%d/%d", g.p[q].M, g.p[q].N, g.p[q].K, g.p[q].kslices);
            fprintf(stderr, "\n");
        }
        HIPCK(ctx, launch_gemm_group(g, cfg, s));
        if (red > 0) {
            const bool take = ctx != nullptr && ctx->cr_pending.count > 0;
            HIPCK(ctx, launch_splitk_reduce(g, red, s, take ? &ctx->cr_pending : nullptr));
            if (take) ctx->cr_pending.count = 0;
        }
    }
    return FBHIP_OK;
}

// ---- round-based merged scheduling -------------------------------------------------------------------------
// A pass (one net forward or backward) is a CHAIN of stages; a stage only DECLARES what it needs at its dependency
// level: GEMM problems, LayerNorm problems, and "post" launches that must follow them.  Independent chains advance in
// lock-step rounds and everything declared in a round goes out as ONE grouped GEMM launch (+ one grouped LayerNorm
// launch): e.g. the second layers of actor(next_obs), forward_net(obs) and both backward nets share a launch, and tiny
// heads ride along the big hidden-layer GEMMs instead of paying a ~6 us launch + pipeline-fill floor each.
// (Measured on MI355X: running independent chains as parallel hipGraph branches instead buys nothing -- the step
// costs the SUM of its kernels' standalone times -- so everything is enqueued on the caller's stream.)
struct Ops {
    std::vector<GemmProblem> gemms;
    std::vector<LnFwdProblem> lnf;
    std::vector<LnBwdProblem> lnb;
    std::vector<L2Problem> l2n;               // run after this round's GEMMs
    std::vector<PolicyHeadJob> ph;            // fused policy heads of this round (one launch for all of them)
    std::vector<DiscreteHeadJob> dh;          // discrete: the selection / gather row jobs of this round (one launch)
    int dh_rows = 0, dh_ldz = 0;
    std::vector<std::function<int(hipStream_t)>> post;
};
using Stage = std::function<void(Ops&)>;
using Chain = std::vector<Stage>;

// the LayerNorm column reduces deferred by the previous round go out now if no split-K reduce launch took them
int flush_colreduce(fbhip_ctx* c, hipStream_t s) {
    if (c->cr_pending.count > 0) {
        GemmGroup none{};
        HIPCK(c, launch_splitk_reduce(none, 0, s, &c->cr_pending));
        c->cr_pending.count = 0;
    }
    return FBHIP_OK;
}

int flush_round(fbhip_ctx* c, Ops& o, hipStream_t s) {
    if (!o.gemms.empty()) RC(run_gemms(c, o.gemms, s));
    RC(flush_colreduce(c, s));
    for (size_t i = 0; i < o.lnf.size(); i += LN_MAX_GROUP) {
        LnFwdGroup g{};
        for (size_t j = i; j < o.lnf.size() && j < i + LN_MAX_GROUP; ++j) g.p[g.n++] = o.lnf[j];
        HIPCK(c, launch_ln_tanh_fwd_group(g, s));
    }
    for (size_t i = 0; i < o.lnb.size(); i += LN_MAX_GROUP) {
        LnBwdGroup g{};
        for (size_t j = i; j < o.lnb.size() && j < i + LN_MAX_GROUP; ++j) g.p[g.n++] = o.lnb[j];
        static const bool defer_cr = [] { const char* e = getenv("FBHIP_DEFER_COLREDUCE"); return !(e && e[0] == '0'); }();
        HIPCK(c, launch_ln_tanh_bwd_group(g, s, defer_cr ? &c->cr_pending : nullptr));
    }
    for (size_t i = 0; i < o.l2n.size(); i += LN_MAX_GROUP) {
        L2Group g{};
        for (size_t j = i; j < o.l2n.size() && j < i + LN_MAX_GROUP; ++j) g.p[g.n++] = o.l2n[j];
        HIPCK(c, launch_l2norm_fwd_group(g, s));
    }
    for (size_t i = 0; i < o.ph.size(); i += PH_MAX_JOBS) {
        PolicyHeadJobs jobs{};
        for (size_t j = i; j < o.ph.size() && j < i + PH_MAX_JOBS; ++j) jobs.j[jobs.n++] = o.ph[j];
        RC(c->run_policy_heads(jobs, s));
    }
    if (!o.dh.empty()) {
        const fbhip_dims& d = c->d;
        for (size_t i = 0; i < o.dh.size(); i += 2) {
            DiscreteHeadJobs jobs{};
            for (size_t j = i; j < o.dh.size() && j < i + 2; ++j) jobs.j[jobs.n++] = o.dh[j];
            HIPCK(c, launch_discrete_heads(jobs, pad4(fhead_out(d)), o.dh_ldz, pad4(d.z_dim), o.dh_rows, d.z_dim, d.action_dim,
                                           d.boltzmann, c->sq.temp, s));
        }
    }
    for (auto& f : o.post) RC(f(s));
    return FBHIP_OK;
}

int run_rounds(fbhip_ctx* c, std::vector<Chain>& chains, hipStream_t s) {
    for (size_t r = 0;; ++r) {
        Ops o;
        bool any = false;
        for (auto& ch : chains)
            if (r < ch.size()) { ch[r](o); any = true; }
        if (!any) return flush_colreduce(c, s);
        RC(flush_round(c, o, s));
    }
}
int run_chain(fbhip_ctx* c, Chain& ch, hipStream_t s) {
    std::vector<Chain> v{ch};
    return run_rounds(c, v, s);
}

// A Program is the round list itself: round r holds the stages of every chain that is r levels deep; build_update
// appends to one, run_program flushes it round by round onto a stream.
using Round = std::vector<Stage>;
using Program = std::vector<Round>;
void prog_parallel(Program& p, std::vector<Chain>& chains) {
    size_t n = 0;
    for (auto& ch : chains) n = ch.size() > n ? ch.size() : n;
    for (size_t r = 0; r < n; ++r) {
        Round rd;
        for (auto& ch : chains)
            if (r < ch.size()) rd.push_back(ch[r]);
        p.push_back(std::move(rd));
    }
}
void prog_chain(Program& p, Chain& ch) {
    for (auto& st : ch) p.push_back(Round{st});
}
void prog_post(Program& p, std::function<int(hipStream_t)> f) {
    p.push_back(Round{[f](Ops& o) { o.post.push_back(f); }});
}
int run_program(fbhip_ctx* c, Program& p, hipStream_t s) {
    for (auto& rd : p) {
        Ops o;
        for (auto& st : rd) st(o);
        RC(flush_round(c, o, s));
    }
    return flush_colreduce(c, s);
}

// ---- network passes as chains ---------------------------------------------------------------------------------
// ForwardMap.forward (fb_modules.py:186-199); Xa = [obs|action] panel, Xz = [obs|z] panel
// (discrete: disc_mode 1 = target-side selection with z = disc_z, 2 = online-side gather of the sampled actions)
void forward_map_fwd_chain(fbhip_ctx* c, const FwdP& W, const float* Xa, int lda, const float* Xz, int ldz, FSet& S,
                           int rows, Chain& out, bool with_heads = true, int disc_mode = 0, const float* disc_z = nullptr,
                           int disc_ldz = 0) {
    const fbhip_dims& d = c->d;
    const Geom gm = geom_of(d);
    const int H = d.hidden_dim, z = d.z_dim, Lz = pad4(z), Fo = gm.Fo, hw = gm.hw, feat = gm.feat;
    FSet* Sp = &S;
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(Xa, lda, 1, W.oa.W1, W.oa.ld1, 1, Sp->pre1a.p, H, rows, H, W.oa.ld1, W.oa.b1, EPI_BIAS));
        if (!gm.single) o.gemms.push_back(P(Xz, ldz, 1, W.oz.W1, W.oz.ld1, 1, Sp->pre1z.p, H, rows, H, W.oz.ld1, W.oz.b1, EPI_BIAS));
    });
    out.push_back([=](Ops& o) {
        o.lnf.push_back(LnFwdProblem{Sp->pre1a.p, H, W.oa.g1, W.oa.be1, Sp->t1a.p, H, Sp->statsA, rows, H, 0, 0, 0, H});
        if (!gm.single) o.lnf.push_back(LnFwdProblem{Sp->pre1z.p, H, W.oz.g1, W.oz.be1, Sp->t1z.p, H, Sp->statsZ, rows, H, 0, 0, 0, H});
    });
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(Sp->t1a.p, H, 1, W.oa.W2, H, 1, Sp->h.p, hw, rows, Fo, H, W.oa.b2, EPI_BIAS_RELU));
        if (!gm.single) o.gemms.push_back(P(Sp->t1z.p, H, 1, W.oz.W2, H, 1, Sp->h.p + Fo, hw, rows, Fo, H, W.oz.b2, EPI_BIAS_RELU));
    });
    // what feeds the heads: h [hw], or with a trunk layer relu(trunk(h)) [H]   (fb_modules.py:194-195)
    if (gm.trunk)
        out.push_back([=](Ops& o) {
            o.gemms.push_back(P(Sp->h.p, hw, 1, W.Wt, hw, 1, Sp->tr.p, H, rows, H, hw, W.bt, EPI_BIAS_RELU));
        });
    out.push_back([=](Ops& o) {
        const float* x = gm.trunk ? Sp->tr.p : Sp->h.p;
        o.gemms.push_back(P(x, feat, 1, W.W3s, feat, 1, Sp->p.p, 2 * H, rows, 2 * H, feat, W.b3s, EPI_BIAS_RELU));
    });
    if (!with_heads) return;                     // the actor phase gets Q from p directly (actor_q_kernel)
    if (d.discrete) {                            // heads emit [rows, z * A]; the embedding the loss sees is picked by a row kernel
        const int zA = fhead_out(d), Lza = pad4(zA);
        Ws* w = &c->W();
        const bool target = disc_mode == 1;
        out.push_back([=](Ops& o) {
            o.gemms.push_back(P(Sp->p.p, 2 * H, 1, W.W4[0], H, 1, Sp->Fall1.p, Lza, rows, zA, H, W.b4[0], EPI_BIAS));
            o.gemms.push_back(P(Sp->p.p + H, 2 * H, 1, W.W4[1], H, 1, Sp->Fall2.p, Lza, rows, zA, H, W.b4[1], EPI_BIAS));
            // the row jobs of a round go out as ONE launch (flush_round): target-side selection (discrete_fb.py:289-303; also
            // act(): the arg-max index) and online-side gather (:309-311)
            o.dh_rows = rows;
            if (target) {
                o.dh_ldz = disc_ldz;
                o.dh.push_back(DiscreteHeadJob{Sp->Fall1.p, Sp->Fall2.p, disc_z, nullptr, Sp->F1.p, Sp->F2.p, w->nextq, w->greedy, 0});
            } else {
                o.dh.push_back(DiscreteHeadJob{Sp->Fall1.p, Sp->Fall2.p, nullptr, w->act_idx, Sp->F1.p, Sp->F2.p, nullptr, nullptr, 1});
            }
        });
        return;
    }
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(Sp->p.p, 2 * H, 1, W.W4[0], H, 1, Sp->F1.p, Lz, rows, z, H, W.b4[0], EPI_BIAS));
        o.gemms.push_back(P(Sp->p.p + H, 2 * H, 1, W.W4[1], H, 1, Sp->F2.p, Lz, rows, z, H, W.b4[1], EPI_BIAS));
    });
}

int forward_map_fwd(fbhip_ctx* c, const FwdP& W, const float* Xa, int lda, const float* Xz, int ldz, FSet& S,
                    int rows, hipStream_t s) {
    Chain ch;
    forward_map_fwd_chain(c, W, Xa, lda, Xz, ldz, S, rows, ch);
    return run_chain(c, ch, s);
}

// dgrad: dp = (dF_i . W4_i) * relu'(p)   (shared by the FB backward and the actor step)
// (runs when a round is flushed: the workspace set comes from the caller, not from the context's current one)
void heads_dgrad_ops(fbhip_ctx* c, Ws& w, const FwdP& W, FSet& S, int rows, Ops& o) {
    const fbhip_dims& d = c->d;
    const int H = d.hidden_dim, z = fhead_out(d), Lz = pad4(z);
    const float* g1 = d.discrete ? w.dFall1.p : w.dF1.p;
    const float* g2 = d.discrete ? w.dFall2.p : w.dF2.p;
    o.gemms.push_back(P(g1, Lz, 1, W.W4[0], H, 0, w.dp.p, 2 * H, rows, H, z, nullptr, EPI_MASK_RELU, S.p.p, 2 * H));
    o.gemms.push_back(P(g2, Lz, 1, W.W4[1], H, 0, w.dp.p + H, 2 * H, rows, H, z, nullptr, EPI_MASK_RELU, S.p.p + H, 2 * H));
}

// full backward of ForwardMap given dF1, dF2 (autograd of fb_ddpg.py:318, :383): each stage holds the weight
// gradient of layer l and the data gradient into layer l-1 (both depend only on the previous stage)
void forward_map_bwd_chain(fbhip_ctx* c, const FwdP& W, const FwdP& G, const float* Xa, int lda, const float* Xz,
                           int ldz, FSet& S, int rows, Chain& out) {
    const fbhip_dims& d = c->d;
    const int H = d.hidden_dim;
    Ws* w = &c->W();
    FSet* Sp = &S;
    // (the heads' output-layer WEIGHT gradients -- thin, 50 x H over K = rows -- wait for the chain's last round, where the other
    // thin-and-deep weight gradients are: next to the wide problems of this round they would need a cross-workgroup split-K
    // and a reduce launch on the critical path; only the optimiser reads them)
    out.push_back([=](Ops& o) { heads_dgrad_ops(c, *w, W, *Sp, rows, o); });
    const Geom gm = geom_of(d);
    const int Fo = gm.Fo, hw = gm.hw, feat = gm.feat;
    const bool trunk = gm.trunk;
    out.push_back([=](Ops& o) {
        const float* x = trunk ? Sp->tr.p : Sp->h.p;          // input of the heads' first layer and its relu mask
        float* dx = trunk ? w->dtr.p : w->dh.p;
        o.gemms.push_back(P(w->dp.p, 2 * H, 0, x, feat, 0, G.W3s, feat, 2 * H, feat, rows, nullptr, EPI_NONE, nullptr, 0, G.b3s));
        o.gemms.push_back(P(w->dp.p, 2 * H, 1, W.W3s, feat, 0, dx, feat, rows, feat, 2 * H, nullptr, EPI_MASK_RELU, x, feat));
    });
    if (trunk)
        out.push_back([=](Ops& o) {
            o.gemms.push_back(P(w->dtr.p, H, 0, Sp->h.p, hw, 0, G.Wt, hw, H, hw, rows, nullptr, EPI_NONE, nullptr, 0, G.bt));
            o.gemms.push_back(P(w->dtr.p, H, 1, W.Wt, hw, 0, w->dh.p, hw, rows, hw, H, nullptr, EPI_MASK_RELU, Sp->h.p, hw));
        });
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(w->dh.p, hw, 0, Sp->t1a.p, H, 0, G.oa.W2, H, Fo, H, rows, nullptr, EPI_NONE, nullptr, 0, G.oa.b2));
        o.gemms.push_back(P(w->dh.p, hw, 1, W.oa.W2, H, 0, w->dt1a.p, H, rows, H, Fo));
        if (!gm.single) {
            o.gemms.push_back(P(w->dh.p + Fo, hw, 0, Sp->t1z.p, H, 0, G.oz.W2, H, Fo, H, rows, nullptr, EPI_NONE, nullptr, 0, G.oz.b2));
            o.gemms.push_back(P(w->dh.p + Fo, hw, 1, W.oz.W2, H, 0, w->dt1z.p, H, rows, H, Fo));
        }
    });
    out.push_back([=](Ops& o) {
        const size_t half = (size_t)((rows + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK) * 2 * H;
        o.lnb.push_back(LnBwdProblem{w->dt1a.p, H, Sp->t1a.p, H, Sp->pre1a.p, H, Sp->statsA, W.oa.g1, w->dt1a.p, H, G.oa.g1,
                                     G.oa.be1, w->ln_partials, rows, H, 0, 0, 0, 0, 0, H});
        if (!gm.single)
            o.lnb.push_back(LnBwdProblem{w->dt1z.p, H, Sp->t1z.p, H, Sp->pre1z.p, H, Sp->statsZ, W.oz.g1, w->dt1z.p, H, G.oz.g1,
                                         G.oz.be1, w->ln_partials + half, rows, H, 0, 0, 0, 0, 0, H});
    });
    out.push_back([=](Ops& o) {
        const int zo = fhead_out(c->d), Lzo = pad4(zo);
        const float* g1 = c->d.discrete ? w->dFall1.p : w->dF1.p;
        const float* g2 = c->d.discrete ? w->dFall2.p : w->dF2.p;
        o.gemms.push_back(P(g1, Lzo, 0, Sp->p.p, 2 * H, 0, G.W4[0], H, zo, H, rows, nullptr, EPI_NONE, nullptr, 0, G.b4[0]));
        o.gemms.push_back(P(g2, Lzo, 0, Sp->p.p + H, 2 * H, 0, G.W4[1], H, zo, H, rows, nullptr, EPI_NONE, nullptr, 0, G.b4[1]));
        o.gemms.push_back(P(w->dt1a.p, H, 0, Xa, lda, 0, G.oa.W1, G.oa.ld1, H, G.oa.ld1, rows, nullptr, EPI_NONE, nullptr, 0, G.oa.b1));
        if (!gm.single)
            o.gemms.push_back(P(w->dt1z.p, H, 0, Xz, ldz, 0, G.oz.W1, G.oz.ld1, H, G.oz.ld1, rows, nullptr, EPI_NONE, nullptr, 0, G.oz.b1));
    });
}

// BackwardMap.forward (fb_modules.py:223-230)
void backward_map_fwd_chain(fbhip_ctx* c, const BwdP& W, const float* X, int ldx, BSet& S, int rows, Chain& out,
                            bool with_projection = true) {
    const fbhip_dims& d = c->d;
    // GEMMs run on the padded width Lb = pad64(Hb) (zero weight rows / columns), LayerNorm on the logical Hb
    const int g = d.goal_dim, Hb = d.backward_hidden_dim, Lb = pad64(Hb), z = d.z_dim, Lz = pad4(z);
    // workspace panels have >= pad32(g) finite columns per row (the weight's pad columns are zero, so whatever sits
    // there contributes nothing); arbitrary caller tensors are read on their logical width
    const bool in_ws = (const char*)X >= c->ws_lo && (const char*)X < c->ws_lo + c->ws_bytes;
    const int Kg = (in_ws && ldx >= pad32(g)) ? pad32(g) : g;
    BSet* Sp = &S;
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(X, ldx, 1, W.W1, pad32(g), 1, Sp->pre1.p, Lb, rows, Lb, Kg, W.b1, EPI_BIAS));
    });
    out.push_back([=](Ops& o) {
        o.lnf.push_back(LnFwdProblem{Sp->pre1.p, Lb, W.g1, W.be1, Sp->t1.p, Lb, Sp->stats, rows, Hb, 0, 0, 0, pad4(Hb)});
    });
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(Sp->t1.p, Lb, 1, W.W2, Lb, 1, Sp->r2.p, Lb, rows, Lb, Lb, W.b2, EPI_BIAS_RELU));
    });
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(Sp->r2.p, Lb, 1, W.W3, Lb, 1, Sp->y.p, Lz, rows, z, Lb, W.b3, EPI_BIAS));
        // cfg.norm_z == False: the map's output IS y (fb_modules.py:228-229); callers read ``bm_of(set)``
        if (with_projection && d.norm_z) o.l2n.push_back(L2Problem{Sp->y.p, Lz, Sp->Bm.p, Lz, Sp->norms, rows, z, sqrtf((float)z)});
    });
}

int backward_map_fwd(fbhip_ctx* c, const BwdP& W, const float* X, int ldx, BSet& S, int rows, hipStream_t s) {
    Chain ch;
    backward_map_fwd_chain(c, W, X, ldx, S, rows, ch);
    return run_chain(c, ch, s);
}

// backward of BackwardMap from dB (gradient wrt the projected embedding)
// gradient panels of one BackwardMap backward (default: the workspace's B-row set; SFAgent's 2B-row feature pass brings its own)
struct BGrad { const float* dBm; float* dy; float* dr2; float* dt1; };
void backward_map_bwd_chain(fbhip_ctx* c, const BwdP& W, const BwdP& G, const float* X, int ldx, BSet& S, int rows,
                            Chain& out, bool dy_done = false, const BGrad* bufs = nullptr) {
    const fbhip_dims& d = c->d;
    const int g = d.goal_dim, Hb = d.backward_hidden_dim, Lb = pad64(Hb), z = d.z_dim, Lz = pad4(z);
    // weight gradient of the first layer: X must be a zero-padded panel to use the padded width
    const bool padded_x = (X == c->W().next_goal.p) || (X == c->W().bin.p);     // zero-padded panels only
    const int Ng = padded_x ? pad32(g) : g;
    Ws* w = &c->W();
    BSet* Sp = &S;
    const BGrad B_ = bufs ? *bufs : BGrad{w->dBm.p, w->dy.p, w->b_dr2.p, w->b_dt1.p};
    const float* dy = d.norm_z ? B_.dy : B_.dBm;        // no projection: the gradient wrt y is dB itself
    out.push_back([=](Ops& o) {                 // dy = d/dy of sqrt(d) normalize(y)   (F.normalize backward)
        if (!c->d.norm_z || dy_done) return;     // (the stage stays, empty: the chain's thin last round must meet forward_net's)
        o.post.push_back([=](hipStream_t s) -> int {
            HIPCK(c, launch_l2norm_bwd(B_.dBm, Lz, Sp->y.p, Lz, Sp->norms, B_.dy, Lz, rows, z, s));
            return (int)FBHIP_OK;
        });
    });
    out.push_back([=](Ops& o) {                 // (the output layer's thin weight gradient waits for the last round, see forward_map_bwd_chain)
        o.gemms.push_back(P(dy, Lz, 1, W.W3, Lb, 0, B_.dr2, Lb, rows, Lb, z, nullptr, EPI_MASK_RELU, Sp->r2.p, Lb));
    });
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(B_.dr2, Lb, 0, Sp->t1.p, Lb, 0, G.W2, Lb, Lb, Lb, rows, nullptr, EPI_NONE, nullptr, 0, G.b2));
        o.gemms.push_back(P(B_.dr2, Lb, 1, W.W2, Lb, 0, B_.dt1, Lb, rows, Lb, Lb));
    });
    out.push_back([=](Ops& o) {
        o.lnb.push_back(LnBwdProblem{B_.dt1, Lb, Sp->t1.p, Lb, Sp->pre1.p, Lb, Sp->stats, W.g1, B_.dt1, Lb, G.g1,
                                     G.be1, w->ln_partials_b, rows, Hb, 0, 0, 0, 0, 0, pad4(Hb)});
    });
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(dy, Lz, 0, Sp->r2.p, Lb, 0, G.W3, Lb, z, Lb, rows, nullptr, EPI_NONE, nullptr, 0, G.b3));
        o.gemms.push_back(P(B_.dt1, Lb, 0, X, ldx, 0, G.W1, pad32(g), Lb, Ng, rows, nullptr, EPI_NONE, nullptr, 0, G.b1));
    });
}

// Actor.forward up to the pre-tanh policy output (fb_modules.py:107-121); Xo supplies obs (first o cols)
void actor_fwd_chain(fbhip_ctx* c, const ActP& W, const float* Xo, int ldo, const float* Xz, int ldz, ASet& S, int rows,
                     Chain& out, bool with_head = true) {
    const fbhip_dims& d = c->d;
    const Geom gm = actor_geom_of(d);
    const int H = d.hidden_dim, a = head_width(d), La = pad4(a), Fo = gm.Fo, hw = gm.hw, feat = gm.feat;
    // preprocess == 0 / boltzmann: the one branch reads [obs|z] (the Xz panel)
    const float* X1 = gm.single ? Xz : Xo;
    const int ld1 = gm.single ? ldz : ldo;
    ASet* Sp = &S;
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(X1, ld1, 1, W.o.W1, W.o.ld1, 1, Sp->pre1o.p, H, rows, H, W.o.ld1, W.o.b1, EPI_BIAS));
        if (!gm.single) o.gemms.push_back(P(Xz, ldz, 1, W.oz.W1, W.oz.ld1, 1, Sp->pre1z.p, H, rows, H, W.oz.ld1, W.oz.b1, EPI_BIAS));
    });
    out.push_back([=](Ops& o) {
        o.lnf.push_back(LnFwdProblem{Sp->pre1o.p, H, W.o.g1, W.o.be1, Sp->t1o.p, H, Sp->statsO, rows, H, 0, 0, 0, H});
        if (!gm.single) o.lnf.push_back(LnFwdProblem{Sp->pre1z.p, H, W.oz.g1, W.oz.be1, Sp->t1z.p, H, Sp->statsZ, rows, H, 0, 0, 0, H});
    });
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(Sp->t1o.p, H, 1, W.o.W2, H, 1, Sp->h.p, hw, rows, Fo, H, W.o.b2, EPI_BIAS_RELU));
        if (!gm.single) o.gemms.push_back(P(Sp->t1z.p, H, 1, W.oz.W2, H, 1, Sp->h.p + Fo, hw, rows, Fo, H, W.oz.b2, EPI_BIAS_RELU));
    });
    if (gm.trunk)                                            // fb_modules.py:116-117
        out.push_back([=](Ops& o) {
            o.gemms.push_back(P(Sp->h.p, hw, 1, W.Wt, hw, 1, Sp->tr.p, H, rows, H, hw, W.bt, EPI_BIAS_RELU));
        });
    if (!gm.boltz)
        out.push_back([=](Ops& o) {
            o.gemms.push_back(P(gm.trunk ? Sp->tr.p : Sp->h.p, feat, 1, W.W3, feat, 1, Sp->p.p, H, rows, H, feat, W.b3, EPI_BIAS_RELU));
        });
    if (!with_head) return;                      // the caller runs the head + the sample as one row kernel (policy_head_kernel)
    out.push_back([=](Ops& o) {                              // head: mu (a wide) / [loc | raw log-std] (2a wide, from h directly)
        o.gemms.push_back(P(gm.boltz ? Sp->h.p : Sp->p.p, H, 1, W.W4, H, 1, Sp->premu.p, La, rows, a, H, W.b4, EPI_BIAS));
    });
}

int actor_fwd(fbhip_ctx* c, const ActP& W, const float* Xo, int ldo, const float* Xz, int ldz, ASet& S, int rows,
              hipStream_t s) {
    Chain ch;
    actor_fwd_chain(c, W, Xo, ldo, Xz, ldz, S, rows, ch);
    return run_chain(c, ch, s);
}

// head_dgrad_done: a_dp (the policy hidden layer's gradient) was already produced by actor_head_bwd_kernel; the head's
// weight gradient then joins the LAST round (the other thin-and-deep weight gradients) instead of having a launch of its own
void actor_bwd_chain(fbhip_ctx* c, const ActP& W, const ActP& G, const float* Xo, int ldo, const float* Xz, int ldz,
                     ASet& S, int rows, Chain& out, bool head_dgrad_done = false) {
    const fbhip_dims& d = c->d;
    const int H = d.hidden_dim, a = head_width(d), La = pad4(a);
    Ws* w = &c->W();
    ASet* Sp = &S;
    const Geom gm = actor_geom_of(d);
    const int Fo = gm.Fo, hw = gm.hw, feat = gm.feat;
    const bool trunk = gm.trunk;
    if (!head_dgrad_done)
        out.push_back([=](Ops& o) {                          // head; boltzmann: straight into d h (there is no policy hidden layer)
            const float* x = gm.boltz ? Sp->h.p : Sp->p.p;
            o.gemms.push_back(P(w->a_dpremu.p, La, 0, x, H, 0, G.W4, H, a, H, rows, nullptr, EPI_NONE, nullptr, 0, G.b4));
            o.gemms.push_back(P(w->a_dpremu.p, La, 1, W.W4, H, 0, gm.boltz ? w->dh.p : w->a_dp.p, H, rows, H, a, nullptr, EPI_MASK_RELU, x, H));
        });
    const float* X1 = gm.single ? Xz : Xo;                   // preprocess == 0: the one branch reads [obs|z]
    const int ld1 = gm.single ? ldz : ldo;
    if (!gm.boltz)
        out.push_back([=](Ops& o) {
            const float* x = trunk ? Sp->tr.p : Sp->h.p;
            float* dx = trunk ? w->dtr.p : w->dh.p;
            o.gemms.push_back(P(w->a_dp.p, H, 0, x, feat, 0, G.W3, feat, H, feat, rows, nullptr, EPI_NONE, nullptr, 0, G.b3));
            o.gemms.push_back(P(w->a_dp.p, H, 1, W.W3, feat, 0, dx, feat, rows, feat, H, nullptr, EPI_MASK_RELU, x, feat));
        });
    if (trunk)
        out.push_back([=](Ops& o) {
            o.gemms.push_back(P(w->dtr.p, H, 0, Sp->h.p, hw, 0, G.Wt, hw, H, hw, rows, nullptr, EPI_NONE, nullptr, 0, G.bt));
            o.gemms.push_back(P(w->dtr.p, H, 1, W.Wt, hw, 0, w->dh.p, hw, rows, hw, H, nullptr, EPI_MASK_RELU, Sp->h.p, hw));
        });
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(w->dh.p, hw, 0, Sp->t1o.p, H, 0, G.o.W2, H, Fo, H, rows, nullptr, EPI_NONE, nullptr, 0, G.o.b2));
        o.gemms.push_back(P(w->dh.p, hw, 1, W.o.W2, H, 0, w->dt1a.p, H, rows, H, Fo));
        if (!gm.single) {
            o.gemms.push_back(P(w->dh.p + Fo, hw, 0, Sp->t1z.p, H, 0, G.oz.W2, H, Fo, H, rows, nullptr, EPI_NONE, nullptr, 0, G.oz.b2));
            o.gemms.push_back(P(w->dh.p + Fo, hw, 1, W.oz.W2, H, 0, w->dt1z.p, H, rows, H, Fo));
        }
    });
    out.push_back([=](Ops& o) {
        const size_t half = (size_t)((rows + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK) * 2 * H;
        o.lnb.push_back(LnBwdProblem{w->dt1a.p, H, Sp->t1o.p, H, Sp->pre1o.p, H, Sp->statsO, W.o.g1, w->dt1a.p, H, G.o.g1,
                                     G.o.be1, w->ln_partials, rows, H, 0, 0, 0, 0, 0, H});
        if (!gm.single)
            o.lnb.push_back(LnBwdProblem{w->dt1z.p, H, Sp->t1z.p, H, Sp->pre1z.p, H, Sp->statsZ, W.oz.g1, w->dt1z.p, H, G.oz.g1,
                                         G.oz.be1, w->ln_partials + half, rows, H, 0, 0, 0, 0, 0, H});
    });
    out.push_back([=](Ops& o) {
        // (the head's weight gradient, thin and deep like the first-layer ones, when actor_head_bwd_kernel made its launch redundant)
        if (head_dgrad_done)
            o.gemms.push_back(P(w->a_dpremu.p, La, 0, Sp->p.p, H, 0, G.W4, H, a, H, rows, nullptr, EPI_NONE, nullptr, 0, G.b4));
        o.gemms.push_back(P(w->dt1a.p, H, 0, X1, ld1, 0, G.o.W1, G.o.ld1, H, G.o.ld1, rows, nullptr, EPI_NONE, nullptr, 0, G.o.b1));
        if (!gm.single)
            o.gemms.push_back(P(w->dt1z.p, H, 0, Xz, ldz, 0, G.oz.W1, G.oz.ld1, H, G.oz.ld1, rows, nullptr, EPI_NONE, nullptr, 0, G.oz.b1));
    });
}

// ---- one update(): fb_ddpg.py:427-520 ----------------------------------------------------------------------
#define POST_BEGIN prog_post(prog, [=, &w](hipStream_t s) -> int {
#define POST_END return (int)FBHIP_OK; });
int build_update(fbhip_ctx* c, const fbhip_hparams& hp, const fbhip_inject* inj, int mask, Program& prog) {
    const fbhip_dims& d = c->d;
    Ws& w = c->W();
    const int B = d.batch, o = d.obs_dim, a = d.action_dim, g = d.goal_dim, z = d.z_dim, H = d.hidden_dim,
              Lz = pad4(z), La = pad4(a);
    const Geom gm = geom_of(d);
    const int Fo = gm.Fo, hw = gm.hw;
    if (d.discrete) mask &= ~(FBHIP_PHASE_ACTOR_GRAD | FBHIP_PHASE_ACTOR_STEP | FBHIP_PHASE_ACTOR_FWD);   // DiscreteFBAgent has no actor
    const int aoff = gm.single ? o + z : o;      // column of the action inside the ForwardMap input panels
    const int Lh = pad4(head_width(d));          // leading dimension of the policy head's output
    // next_goal = batch.next_goal if goal_space else batch.next_obs (fb_ddpg.py:440-443); always its own zero-padded panel
    const float* next_goal = w.next_goal.p;
    const int ld_ng = w.next_goal.ld;

    if (mask & FBHIP_PHASE_SAMPLE) {            // (the RNG counter is advanced by mix_z_kernel at the end of the phase)
        // every NULL field of ``inj`` is drawn on device; injected fields (parity mode / externally sampled
        // batches) overwrite the draw
        const bool hindsight = hp.future_ratio > 0.f;
        if (hindsight && !(hp.future < 1.f)) { c->err = g_err = "fbhip: future_ratio > 0 needs a replay buffer with future < 1"; return FBHIP_E_INVALID; }
        const bool all_injected = inj && inj->ep_idx && inj->step_idx && inj->perm && inj->mix_uniform &&
                                  inj->z_gauss && inj->eps_next && inj->eps_actor &&
                                  (!hindsight || (inj->future_idx && inj->future_uniform)) && (d.norm_z || inj->z_uniform);
        const bool randw = hp.rand_weight != 0 && hp.mix_ratio > 0.f;
        const bool randw_injected = randw && inj && inj->rand_weight && inj->rand_weight_u;
        POST_BEGIN
        if (!all_injected) HIPCK(c, launch_draw(c->rv, w.so, B, z, a, c->seed, c->rank, w.st, hindsight ? hp.future : -1.f, d.norm_z, s));
        if (inj != nullptr) {
#define INJ(field, bytes) if (inj->field) HIPCK(c, hipMemcpyAsync(w.so.field, inj->field, (size_t)(bytes), hipMemcpyDeviceToDevice, s))
            INJ(ep_idx, B * 4); INJ(step_idx, B * 4); INJ(perm, B * 4); INJ(mix_uniform, B * 4);
            INJ(z_gauss, (size_t)B * z * 4); INJ(eps_next, (size_t)B * a * 4); INJ(eps_actor, (size_t)B * a * 4);
            if (hindsight) { INJ(future_idx, B * 4); INJ(future_uniform, B * 4); }
            if (!d.norm_z) INJ(z_uniform, (size_t)B * z * 4);
            if (randw_injected) {
                HIPCK(c, hipMemcpyAsync(w.rw, inj->rand_weight, (size_t)B * B * 4, hipMemcpyDeviceToDevice, s));
                HIPCK(c, hipMemcpyAsync(w.rw_u, inj->rand_weight_u, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
            }
#undef INJ
        }
        GatherArgs ga{};
        ga.rv = c->rv; ga.ep_idx = w.so.ep_idx; ga.step_idx = w.so.step_idx; ga.perm = w.so.perm;
        ga.Xoa = w.Xoa.p; ga.ld_oa = w.Xoa.ld; ga.Xoz = w.Xoz.p; ga.ld_oz = w.Xoz.ld; ga.Xnoz = w.Xnoz.p; ga.ld_noz = w.Xnoz.ld;
        ga.Xnoa = w.Xnoa.p; ga.ld_noa = w.Xnoa.ld; ga.Xopi = w.Xopi.p; ga.ld_opi = w.Xopi.ld;
        ga.next_goal = w.next_goal.p; ga.ld_ng = w.next_goal.ld; ga.bin = w.bin.p; ga.ld_bin = w.bin.ld; ga.disc = w.disc;
        ga.Xo = w.Xo.p; ga.ld_o = w.Xo.ld;
        ga.future_idx = hindsight ? w.so.future_idx : nullptr; ga.fgoal = w.fgoal.p; ga.ld_fg = w.fgoal.ld;
        ga.B = B; ga.o = o; ga.a = d.discrete ? 1 : a; ga.g = g; ga.use_goal = d.use_goal; ga.gamma = hp.discount; ga.aoff = aoff;
        ga.act_idx = d.discrete ? w.act_idx : nullptr;
        HIPCK(c, launch_gather(ga, s));
        POST_END
        // sample_z (fb_ddpg.py:224-228) + z-mix (fb_ddpg.py:470-485: z[mix] = sqrt(d) normalize(B(backward_input[perm])))
        // in one row kernel; the BackwardMap pass stops at its raw mlp output y (the kernel applies both projections)
        // When the FB step follows in the same call, the target / online BackwardMap passes on next_goal (fb_ddpg.py:312,
        // :319) share these launches: they only need the gathered batch.
        {
            std::vector<Chain> ch;
            if (hp.mix_ratio > 0.f) {           // rand_weight mixes COMPLETE BackwardMap outputs (projection included)
                ch.emplace_back();
                backward_map_fwd_chain(c, c->K_p, w.bin.p, w.bin.ld, w.bsM, B, ch.back(), /*with_projection=*/randw);
            }
            if (hindsight) {                    // B(future_goal), fb_ddpg.py:491 (its projection happens in mix_z_kernel)
                ch.emplace_back();
                backward_map_fwd_chain(c, c->K_p, w.fgoal.p, w.fgoal.ld, w.bsF, B, ch.back(), /*with_projection=*/false);
            }
            if (mask & FBHIP_PHASE_FB_FWD_ONLINE) {
                ch.emplace_back();
                backward_map_fwd_chain(c, c->K_t, next_goal, ld_ng, w.bsA, B, ch.back());
                ch.emplace_back();
                backward_map_fwd_chain(c, c->K_p, next_goal, ld_ng, w.bsO, B, ch.back());
            }
            prog_parallel(prog, ch);
        }
        POST_BEGIN
        const float* ymix = w.bsM.y.p;
        if (randw) {
            // mix_z = (u * normalize(rand[., B])) @ backward_net(backward_input[perm])   (fb_ddpg.py:475-482), all rows
            HIPCK(c, launch_rand_weight(w.rw, w.rw_u, B, randw_injected ? 0 : 1, c->seed, c->rank, w.st, s));
            const Buf& bm = d.norm_z ? w.bsM.Bm : w.bsM.y;
            RC(run_gemms(c, {P(w.rw, B, 1, bm.p, bm.ld, 0, w.ymixw.p, Lz, B, z, B)}, s));
            ymix = w.ymixw.p;
        }
        ZPanels zx{};                            // preprocess == 0: z also sits inside the three ForwardMap panels
        if (gm.single) zx = ZPanels{{w.Xoa.p, w.Xnoa.p, w.Xopi.p}, {w.Xoa.ld, w.Xnoa.ld, w.Xopi.ld}};
        HIPCK(c, launch_mix_z(w.so.z_gauss, z, ymix, Lz, w.so.mix_uniform, hp.mix_ratio, w.z.p, Lz, w.Xoz.p, w.Xoz.ld,
                              w.Xnoz.p, w.Xnoz.ld, o, B, z, w.st, hindsight ? w.bsF.y.p : nullptr, w.so.future_uniform,
                              hp.future_ratio, d.norm_z ? nullptr : w.so.z_uniform, randw ? 1 : 2,
                              zx, s));
        POST_END
    }

    // the actor's own forward pass of update_actor (fb_ddpg.py:395-397) reads only the actor weights and (obs, z), not
    // forward_net or the new FB weights: in a call that also runs the FB backward it shares that backward's launches
    // (FB_BWD is two bits, see below: pass ACTOR_FWD with both or with neither)
    // ... and when the call also runs the target chain, whose first half is the SAME actor on next_obs, it rides there instead:
    // layer by layer the two passes are two problems of the same launches (own activation sets: w.as / w.asT)
    static const bool awt_env = [] { const char* e = getenv("FBHIP_ACTOR_WITH_TARGET"); return !(e && e[0] == '0'); }();
    const bool actor_with_target = awt_env && (mask & FBHIP_PHASE_FB_FWD_TARGET) && (mask & FBHIP_PHASE_ACTOR_FWD);
    // The optimiser counters are advanced by the first kernel of the call that precedes the optimiser pass anyway:
    // pairwise_reduce_kernel for fb_opt (3 = both optimisers when the call also holds the actor step), actor_q_kernel for a
    // lone actor step; calls that hold only the STEP phase launch step_advance_kernel.
    static const bool adv_env = [] { const char* e = getenv("FBHIP_FUSED_STEP_ADVANCE"); return !(e && e[0] == '0'); }();
    const int fb_adv_which = (mask & FBHIP_PHASE_ACTOR_STEP) ? 3 : 0;
    const bool fb_adv = adv_env && (mask & FBHIP_PHASE_FB_BWD_A) && (mask & FBHIP_PHASE_FB_STEP);
    const bool actor_adv = adv_env && (mask & FBHIP_PHASE_ACTOR_GRAD) && (mask & FBHIP_PHASE_ACTOR_STEP) &&
                           !(mask & FBHIP_PHASE_FB_STEP);
    // d/dy of B = sqrt(d) y/|y| is a row operation on the loss kernel's own output dB: pairwise_reduce_kernel does it when the
    // call continues with the backward (the BackwardMap chain then skips its l2norm_bwd launch)
    static const bool dy_env = [] { const char* e = getenv("FBHIP_FUSED_L2NORM_BWD"); return !(e && e[0] == '0'); }();
    const bool fused_dy = dy_env && d.norm_z && (mask & FBHIP_PHASE_FB_BWD_A) && pad4(z) == w.bsO.y.ld && z <= 128;
    const bool early_actor = !actor_with_target && (mask & FBHIP_PHASE_FB_BWD) && (mask & FBHIP_PHASE_ACTOR_FWD);
    // policy head + sample: one row kernel when the head's width has an instantiation and its weight fits 48 KB of LDS,
    // else head GEMM (in the chain) + sample
    static const bool head_env = [] { const char* e = getenv("FBHIP_FUSED_POLICY_HEAD"); return !(e && e[0] == '0'); }();
    const bool fused_policy = head_env && policy_head_ok(H, head_width(d));
    if (fused_policy) {
        const float stddev = hp.stddev, clip = hp.stddev_clip;
        c->run_policy_heads = [=](const PolicyHeadJobs& jobs, hipStream_t q) -> int {
            HIPCK(c, launch_policy_head(jobs, c->A_p.W4, H, c->A_p.b4, Lh, a, stddev, clip, La, B, H, a, head_width(d), c->sq, q));
            return (int)FBHIP_OK;
        };
    }
    auto policy_stage = [=, &w](const float* noise, float* mu, float* action_dst, int ld_dst, ASet* set = nullptr) {
        ASet* S = set ? set : &w.as;
        return [=](Ops& o2) {
            if (fused_policy) {                  // all heads of a round go out as one launch (flush_round)
                o2.ph.push_back(PolicyHeadJob{d.boltzmann ? S->h.p : S->p.p, H, S->premu.p, noise, mu, action_dst, ld_dst});
                return;
            }
            o2.post.push_back([=](hipStream_t q) -> int {
                HIPCK(c, launch_policy_sample(S->premu.p, Lh, noise, a, hp.stddev, hp.stddev_clip, mu, La, action_dst,
                                              ld_dst, B, a, c->sq, q));
                return (int)FBHIP_OK;
            });
        };
    };

    if (mask & FBHIP_PHASE_FB_FWD) {
        {
            // chain A: targets, no grad (fb_ddpg.py:303-315): actor(next_obs) -> next_action -> forward_target
            // chain B: online F (fb_ddpg.py:318)    chains C, D: target B (:312) and online B (:319)
            // (C, D already ran with the sampler's z-mix pass when this call also covered the SAMPLE phase)
            std::vector<Chain> ch(2);
            if ((mask & FBHIP_PHASE_FB_FWD_TARGET) && d.discrete)      // discrete_fb.py:289-303: no actor, the greedy / softmax column
                forward_map_fwd_chain(c, c->F_t, w.Xnoz.p, w.Xnoz.ld, w.Xnoz.p, w.Xnoz.ld, w.fsT, B, ch[0], true, 1, w.z.p, Lz);
            else if (mask & FBHIP_PHASE_FB_FWD_TARGET) {
                actor_fwd_chain(c, c->A_p, w.Xnoz.p, w.Xnoz.ld, w.Xnoz.p, w.Xnoz.ld, w.asT, B, ch[0], !fused_policy);
                ch[0].push_back(policy_stage(w.so.eps_next, nullptr, w.Xnoa.p + aoff, w.Xnoa.ld, &w.asT));
                forward_map_fwd_chain(c, c->F_t, w.Xnoa.p, w.Xnoa.ld, w.Xnoz.p, w.Xnoz.ld, w.fsT, B, ch[0]);
            }
            if ((mask & FBHIP_PHASE_FB_FWD_ONLINE) && d.discrete)      // discrete_fb.py:309-311
                forward_map_fwd_chain(c, c->F_p, w.Xoz.p, w.Xoz.ld, w.Xoz.p, w.Xoz.ld, w.fsO, B, ch[1], true, 2);
            else if (mask & FBHIP_PHASE_FB_FWD_ONLINE)
                forward_map_fwd_chain(c, c->F_p, w.Xoa.p, w.Xoa.ld, w.Xoz.p, w.Xoz.ld, w.fsO, B, ch[1]);
            if (actor_with_target) {             // update_actor's own actor pass (fb_ddpg.py:395-397), see above
                ch.emplace_back();
                actor_fwd_chain(c, c->A_p, w.Xo.p, w.Xo.ld, w.Xoz.p, w.Xoz.ld, w.as, B, ch.back(), !fused_policy);
                ch.back().push_back(policy_stage(w.so.eps_actor, w.as.mu.p, w.Xopi.p + aoff, w.Xopi.ld));
            }
            if ((mask & FBHIP_PHASE_FB_FWD_ONLINE) && !(mask & FBHIP_PHASE_SAMPLE)) {
                ch.emplace_back();
                backward_map_fwd_chain(c, c->K_t, next_goal, ld_ng, w.bsA, B, ch.back());
                ch.emplace_back();
                backward_map_fwd_chain(c, c->K_p, next_goal, ld_ng, w.bsO, B, ch.back());
            }
            // both ForwardMap chains in one call: the online heads' thin output layer (+ its split-K reduce) waits for the round
            // of the target chain's, five rounds later -- one launch pair instead of two, nothing needs it earlier
            static const bool align_env = [] { const char* e = getenv("FBHIP_ALIGN_HEADS"); return !(e && e[0] == '0'); }();
            if (align_env && ch[0].size() > ch[1].size() && !ch[1].empty()) {
                Stage heads = ch[1].back();
                ch[1].pop_back();
                while (ch[1].size() + 1 < ch[0].size()) ch[1].push_back([](Ops&) {});
                ch[1].push_back(heads);
            }
            prog_parallel(prog, ch);
        }
    }
    if (mask & FBHIP_PHASE_FB_BWD_A) {
        // --- pairwise loss + dF1, dF2, dB (fb_ddpg.py:320-348, :383)
        const float* BmO = d.norm_z ? w.bsO.Bm.p : w.bsO.y.p;      // online / target B(next_goal) as the loss sees them
        const float* BmT = d.norm_z ? w.bsA.Bm.p : w.bsA.y.p;
        const int Bg = c->gb_rows;              // > 0: the loss couples the rows of ALL ranks (global-batch data parallel)
        POST_BEGIN
        if (Bg > 0) {
            const float* G = c->gb_panels;
            const size_t ps = (size_t)Bg * Lz;
            HIPCK(c, launch_pairwise_fb_block(G, G + ps, G + 2 * ps, G + 3 * ps, G + 4 * ps, G + 5 * ps, c->gb_discount, Bg, z,
                                              Lz, hp.ortho_coef, c->gb_off, B, w.dF1.p, w.dF2.p, w.dBm.p, w.metrics,
                                              w.pw_scratch, s, fb_adv ? w.st : nullptr, fb_adv_which,
                                              fused_dy ? w.bsO.y.p : nullptr, w.bsO.norms, w.dy.p));
            if (hp.want_metrics || hp.q_loss)   // B^T B over the global rows (identical on every rank)
                RC(run_gemms(c, {P(G + 2 * ps, Lz, 0, G + 2 * ps, Lz, 0, w.cov.p, w.cov.ld, z, z, Bg)}, s));
        } else {
            HIPCK(c, launch_pairwise_fb(w.fsO.F1.p, w.fsO.F2.p, BmO, w.fsT.F1.p, w.fsT.F2.p, BmT, w.disc, B, z,
                                        Lz, hp.ortho_coef, w.dF1.p, w.dF2.p, w.dBm.p, w.metrics, w.pw_scratch, s,
                                        fb_adv ? w.st : nullptr, fb_adv_which, fused_dy ? w.bsO.y.p : nullptr, w.bsO.norms, w.dy.p));
            if (hp.want_metrics || hp.q_loss)   // B^T B: metrics (fb_ddpg.py:371) and the q_loss covariance (:334)
                RC(run_gemms(c, {P(BmO, Lz, 0, BmO, Lz, 0, w.cov.p, w.cov.ld, z, z, B)}, s));
        }
        if (hp.q_loss) {                        // fb_ddpg.py:330-340
            HIPCK(c, launch_inverse(w.cov.p, w.cov.ld, z, 1.0f / (float)(Bg > 0 ? Bg : B), w.inv_cov.p, w.inv_cov.ld, s));
            RC(run_gemms(c, {P(BmO, Lz, 1, w.inv_cov.p, w.inv_cov.ld, 0, w.BinvC.p, Lz, B, z, z)}, s));
            HIPCK(c, launch_qloss(w.fsO.F1.p, w.fsO.F2.p, w.fsT.F1.p, w.fsT.F2.p, w.BinvC.p, w.z.p, Lz, w.disc,
                                  hp.q_loss_coef, w.dF1.p, w.dF2.p, w.metrics, w.pw_scratch, B, z, s, Bg,
                                  d.discrete ? w.nextq : nullptr));      // discrete_fb.py:297, :302, :329
        }
        if (hp.want_metrics) {                  // fb_ddpg.py:356-377
            HIPCK(c, launch_extra_metrics(w.fsO.F1.p, BmO, w.z.p, Lz, B, z, w.cov.p, w.cov.ld, w.metrics, s, Bg));
        }
        if (d.discrete && (mask & FBHIP_PHASE_FB_BWD_A))      // backward of the action gather (discrete_fb.py:310)
            HIPCK(c, launch_discrete_scatter(w.dF1.p, w.dF2.p, Lz, w.act_idx, w.dFall1.p, w.dFall2.p, pad4(fhead_out(d)), B, z,
                                             d.action_dim, s));
        POST_END
    }
    if (mask & FBHIP_PHASE_FB_BWD) {
        {
            // --- backward (fb_ddpg.py:383): forward_net, backward_net and (early) the actor's own forward pass.
            // FB_BWD_A stops after the two rounds that finish the gradients of the ForwardMap heads' hidden layers (F{1,2}.0:
            // 57 % of the FB bucket at walker dims, fbhip_fb_early_grad_range), FB_BWD_B runs the rest: a data-parallel host
            // starts the all-reduce of that range in between and hides it under FB_BWD_B.
            std::vector<Chain> ch(3);
            const Buf& Xa = d.discrete ? w.Xoz : w.Xoa;
            forward_map_bwd_chain(c, c->F_p, c->F_g, Xa.p, Xa.ld, w.Xoz.p, w.Xoz.ld, w.fsO, B, ch[0]);
            backward_map_bwd_chain(c, c->K_p, c->K_g, next_goal, ld_ng, w.bsO, B, ch[1], fused_dy);
            if (early_actor) {
                actor_fwd_chain(c, c->A_p, w.Xo.p, w.Xo.ld, w.Xoz.p, w.Xoz.ld, w.as, B, ch[2], !fused_policy);
                ch[2].push_back(policy_stage(w.so.eps_actor, w.as.mu.p, w.Xopi.p + aoff, w.Xopi.ld));
            }
            Program bw;
            prog_parallel(bw, ch);
            const size_t cut = bw.size() < 2 ? bw.size() : 2;
            if (mask & FBHIP_PHASE_FB_BWD_A) prog.insert(prog.end(), bw.begin(), bw.begin() + cut);
            if (mask & FBHIP_PHASE_FB_BWD_B) prog.insert(prog.end(), bw.begin() + cut, bw.end());
        }
    }

    if (mask & FBHIP_PHASE_FB_STEP) {           // fb_opt.step() (:384) + soft_update_params x2 (:500-503)
        POST_BEGIN
        if (!fb_adv) HIPCK(c, launch_step_advance(w.st, fb_adv_which, s));   // (else pairwise_reduce_kernel did it)
        const int64_t nf = c->L[FBHIP_NET_FORWARD].numel, nb = c->L[FBHIP_NET_BACKWARD].numel;
        HIPCK(c, launch_adam_ema(c->fb_p, c->fb_g, c->fb_m, c->fb_v, c->fb_t, nf + nb, hp.lr, hp.lr_coef * hp.lr, nf,
                                 hp.grad_scale, hp.fb_target_tau, w.st, 0, 0, s));
        POST_END
    }

    if (mask & FBHIP_PHASE_ACTOR_GRAD) {        // update_actor, fb_ddpg.py:389-410
        Chain ch;
        if ((mask & FBHIP_PHASE_ACTOR_FWD) && !early_actor && !actor_with_target) {
            actor_fwd_chain(c, c->A_p, w.Xo.p, w.Xo.ld, w.Xoz.p, w.Xoz.ld, w.as, B, ch, !fused_policy);
            ch.push_back(policy_stage(w.so.eps_actor, w.as.mu.p, w.Xopi.p + aoff, w.Xopi.ld));
        }
        // ForwardMap up to the heads' hidden activations p; the heads' outputs F1, F2 are never formed: with V = z . W4
        // (no dependence on this pass, so it joins the chain's first round) Q_i = p_i . V_i + b4_i . z and the heads'
        // data gradient is -(w_i / B) V_i * relu'(p_i) -- one row kernel instead of GEMM + reduce + loss + GEMM
        forward_map_fwd_chain(c, c->F_p, w.Xopi.p, w.Xopi.ld, w.Xoz.p, w.Xoz.ld, w.fsO, B, ch, /*with_heads=*/false);
        {
            Stage first = ch.front();
            ch.front() = [=, &w](Ops& o2) {
                first(o2);
                o2.gemms.push_back(P(w.z.p, Lz, 1, c->F_p.W4[0], H, 0, w.dp.p, 2 * H, B, H, z));
                o2.gemms.push_back(P(w.z.p, Lz, 1, c->F_p.W4[1], H, 0, w.dp.p + H, 2 * H, B, H, z));
            };
        }
        // (data-gradient only along the action path of forward_net: the reference also computes and discards every weight
        // gradient of forward_net here)
        ch.push_back([=, &w](Ops& o2) {
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, launch_actor_q(w.fsO.p.p, 2 * H, w.dp.p, 2 * H, w.z.p, Lz, c->F_p.b4[0], c->F_p.b4[1], w.as.mu.p, La,
                                        w.Xopi.p + aoff, w.Xopi.ld, hp.stddev, hp.want_metrics ? w.metrics : nullptr,
                                        w.pw_scratch, B, H, z, a, c->sq, w.as.premu.p, Lh, w.so.eps_actor, a, q,
                                        actor_adv ? w.st : nullptr, 1));
                return (int)FBHIP_OK;
            });
        });
        // (only the branch that sees the action matters: the first Fo columns of h)
        if (gm.trunk) {
            ch.push_back([=, &w](Ops& o2) {          // d relu(trunk(h)) ...
                o2.gemms.push_back(P(w.dp.p, 2 * H, 1, c->F_p.W3s, H, 0, w.dtr.p, H, B, H, 2 * H, nullptr, EPI_MASK_RELU, w.fsO.tr.p, H));
            });
            ch.push_back([=, &w](Ops& o2) {
                o2.gemms.push_back(P(w.dtr.p, H, 1, c->F_p.Wt, hw, 0, w.dh.p, hw, B, Fo, H, nullptr, EPI_MASK_RELU, w.fsO.h.p, hw));
            });
        } else {
            ch.push_back([=, &w](Ops& o2) {
                o2.gemms.push_back(P(w.dp.p, 2 * H, 1, c->F_p.W3s, hw, 0, w.dh.p, hw, B, Fo, 2 * H, nullptr, EPI_MASK_RELU, w.fsO.h.p, hw));
            });
        }
        ch.push_back([=, &w](Ops& o2) { o2.gemms.push_back(P(w.dh.p, hw, 1, c->F_p.oa.W2, H, 0, w.dt1a.p, H, B, H, Fo)); });
        static const bool fuse_env = [] { const char* e = getenv("FBHIP_FUSED_ACTOR_HEAD"); return !(e && e[0] == '0'); }();
        const bool fused_head = fuse_env && !d.boltzmann && actor_head_bwd_ok(H, a);
        static const bool ln_env = [] { const char* e = getenv("FBHIP_FUSED_ACTOR_HEAD_LN"); return !(e && e[0] == '0'); }();
        const bool fused_ln = fused_head && ln_env && H <= 2048;      // the LayerNorm+tanh backward of this row chain joins the kernel
        if (!fused_ln)
            ch.push_back([=, &w](Ops& o2) {
                o2.lnb.push_back(LnBwdProblem{w.dt1a.p, H, w.fsO.t1a.p, H, w.fsO.pre1a.p, H, w.fsO.statsA, c->F_p.oa.g1, w.dt1a.p, H,
                                              nullptr, nullptr, nullptr, B, H, 0, 0, 0, 0, 0, H});
            });
        // d action -> d mu (straight-through clamp, utils.py:171-174) -> d pre-tanh

        if (d.boltzmann)                         // ... or through the SquashedNormal's rsample and log_prob (fb_ddpg.py:393-406)
            ch.push_back([=, &w](Ops& o2) {
                o2.gemms.push_back(P(w.dt1a.p, H, 1, c->F_p.oa.W1 + aoff, c->F_p.oa.ld1, 0, w.a_dact.p, La, B, a, H));
                o2.post.push_back([=, &w](hipStream_t q) -> int {
                    HIPCK(c, launch_squash_head_bwd(w.a_dact.p, La, w.as.premu.p, Lh, w.so.eps_actor, a, w.a_dpremu.p, Lh, B, a,
                                                    c->sq, q));
                    return (int)FBHIP_OK;
                });
            });
        else if (fused_head)
            ch.push_back([=, &w](Ops& o2) {      // d action -> d premu -> d p in one row kernel (actor_head_bwd_kernel)
                o2.post.push_back([=, &w](hipStream_t q) -> int {
                    if (fused_ln)
                        HIPCK(c, launch_actor_head_bwd(w.dt1a.p, H, c->F_p.oa.W1 + aoff, c->F_p.oa.ld1, w.as.mu.p, La, c->A_p.W4, H,
                                                       w.as.p.p, H, w.a_dpremu.p, La, w.a_dp.p, H, B, H, a, q, w.fsO.t1a.p, H,
                                                       w.fsO.pre1a.p, H, w.fsO.statsA, c->F_p.oa.g1));
                    else
                        HIPCK(c, launch_actor_head_bwd(w.dt1a.p, H, c->F_p.oa.W1 + aoff, c->F_p.oa.ld1, w.as.mu.p, La, c->A_p.W4, H,
                                                       w.as.p.p, H, w.a_dpremu.p, La, w.a_dp.p, H, B, H, a, q));
                    return (int)FBHIP_OK;
                });
            });
        else
            ch.push_back([=, &w](Ops& o2) {
                o2.gemms.push_back(P(w.dt1a.p, H, 1, c->F_p.oa.W1 + aoff, c->F_p.oa.ld1, 0, w.a_dpremu.p, La, B, a, H, nullptr,
                                     EPI_TANH_BWD, w.as.mu.p, La));
            });
        actor_bwd_chain(c, c->A_p, c->A_g, w.Xo.p, w.Xo.ld, w.Xoz.p, w.Xoz.ld, w.as, B, ch, fused_head);
        prog_chain(prog, ch);
    }

    if (mask & FBHIP_PHASE_ACTOR_STEP) {        // actor_opt.step(), fb_ddpg.py:411
        POST_BEGIN
        if (!(mask & FBHIP_PHASE_FB_STEP) && !actor_adv) HIPCK(c, launch_step_advance(w.st, 1, s));   // (else actor_q_kernel did it)
        const int64_t na = c->L[FBHIP_NET_ACTOR].numel;
        HIPCK(c, launch_adam_ema(c->a_p, c->a_g, c->a_m, c->a_v, nullptr, na, hp.lr, hp.lr, na, hp.grad_scale, 0.f, w.st,
                                 1, 0, s));
        POST_END
    }
    return FBHIP_OK;
}
#undef POST_BEGIN
#undef POST_END

// ---- one SFAgent.update(): sf.py:700-768 (dims.sf) ---------------------------------------------------------------------
// Everything up to and including the two critic-side optimiser steps (sf_opt, phi_opt: the two lr groups of the FB flat
// buffer); the actor phase (sf.py:666-694) and the target EMA are FBDDPGAgent's and come from build_update.
#define POST_BEGIN prog_post(prog, [=, &w](hipStream_t s) -> int {
#define POST_END return (int)FBHIP_OK; });
int build_update_sf(fbhip_ctx* c, const fbhip_hparams& hp, const fbhip_inject* inj, Program& prog) {
    const fbhip_dims& d = c->d;
    Ws& w = c->W();
    const int B = d.batch, o = d.obs_dim, a = d.action_dim, g = d.goal_dim, z = d.z_dim, H = d.hidden_dim,
              Hb = d.backward_hidden_dim, Lb = pad64(Hb), Lz = pad4(z), La = pad4(a);
    const Geom gm = geom_of(d);
    const int aoff = gm.single ? o + z : o;
    if (hp.mix_ratio != 0.f || hp.future_ratio != 0.f || hp.rand_weight) {
        c->err = g_err = "fbhip: dims.sf supports the reference's default z sampling only (mix_ratio = 0, sf.py:728-743 not built)";
        return FBHIP_E_INVALID;
    }
    // ---- sample: the FB sampler with the identity permutation: goal2 = [goal ; next_goal] (sf.py:705-721), z = sample_z (:723)
    const bool all_injected = inj && inj->ep_idx && inj->step_idx && inj->z_gauss && inj->eps_next && inj->eps_actor;
    POST_BEGIN
    if (!all_injected) HIPCK(c, launch_draw(c->rv, w.so, B, z, a, c->seed, c->rank, w.st, -1.f, 1, s));
    if (inj != nullptr) {
#define INJ(field, bytes) if (inj->field) HIPCK(c, hipMemcpyAsync(w.so.field, inj->field, (size_t)(bytes), hipMemcpyDeviceToDevice, s))
        INJ(ep_idx, B * 4); INJ(step_idx, B * 4); INJ(z_gauss, (size_t)B * z * 4); INJ(eps_next, (size_t)B * a * 4);
        INJ(eps_actor, (size_t)B * a * 4);
#undef INJ
    }
    GatherArgs ga{};
    ga.rv = c->rv; ga.ep_idx = w.so.ep_idx; ga.step_idx = w.so.step_idx; ga.perm = nullptr;
    ga.Xoa = w.Xoa.p; ga.ld_oa = w.Xoa.ld; ga.Xoz = w.Xoz.p; ga.ld_oz = w.Xoz.ld; ga.Xnoz = w.Xnoz.p; ga.ld_noz = w.Xnoz.ld;
    ga.Xnoa = w.Xnoa.p; ga.ld_noa = w.Xnoa.ld; ga.Xopi = w.Xopi.p; ga.ld_opi = w.Xopi.ld;
    ga.next_goal = w.next_goal.p; ga.ld_ng = w.next_goal.ld; ga.bin = w.bin.p; ga.ld_bin = w.bin.ld; ga.disc = w.disc;
    ga.Xo = w.Xo.p; ga.ld_o = w.Xo.ld; ga.future_idx = nullptr; ga.fgoal = w.fgoal.p; ga.ld_fg = w.fgoal.ld;
    ga.B = B; ga.o = o; ga.a = a; ga.g = g; ga.use_goal = d.use_goal; ga.gamma = hp.discount; ga.aoff = aoff; ga.act_idx = nullptr;
    HIPCK(c, launch_gather(ga, s));
    ZPanels zx{};
    if (gm.single) zx = ZPanels{{w.Xoa.p, w.Xnoa.p, w.Xopi.p}, {w.Xoa.ld, w.Xnoa.ld, w.Xopi.ld}};
    HIPCK(c, launch_mix_z(w.so.z_gauss, z, nullptr, Lz, w.so.mix_uniform, 0.f, w.z.p, Lz, w.Xoz.p, w.Xoz.ld, w.Xnoz.p, w.Xnoz.ld,
                          o, B, z, w.st, nullptr, nullptr, 0.f, nullptr, 2, zx, s));
    POST_END

    // ---- forward passes: target chain (actor(next_obs) -> next_action -> successor_target), online successor_net, the
    // feature pass on [goal ; next_goal], and update_actor's own actor pass (it reads only the actor weights)
    static const bool head_env = [] { const char* e = getenv("FBHIP_FUSED_POLICY_HEAD"); return !(e && e[0] == '0'); }();
    const bool fused_policy = head_env && policy_head_ok(H, a);
    if (fused_policy) {
        const float stddev = hp.stddev, clip = hp.stddev_clip;
        c->run_policy_heads = [=](const PolicyHeadJobs& jobs, hipStream_t q) -> int {
            HIPCK(c, launch_policy_head(jobs, c->A_p.W4, H, c->A_p.b4, La, a, stddev, clip, La, B, H, a, a, c->sq, q));
            return (int)FBHIP_OK;
        };
    }
    auto policy_stage = [=](const float* noise, float* mu, float* action_dst, int ld_dst, ASet* S) {
        return [=](Ops& o2) {
            if (fused_policy) { o2.ph.push_back(PolicyHeadJob{S->p.p, H, S->premu.p, noise, mu, action_dst, ld_dst}); return; }
            o2.post.push_back([=](hipStream_t q) -> int {
                HIPCK(c, launch_policy_sample(S->premu.p, La, noise, a, hp.stddev, hp.stddev_clip, mu, La, action_dst, ld_dst, B, a, c->sq, q));
                return (int)FBHIP_OK;
            });
        };
    };
    {
        std::vector<Chain> ch(4);
        actor_fwd_chain(c, c->A_p, w.Xnoz.p, w.Xnoz.ld, w.Xnoz.p, w.Xnoz.ld, w.asT, B, ch[0], !fused_policy);
        ch[0].push_back(policy_stage(w.so.eps_next, nullptr, w.Xnoa.p + aoff, w.Xnoa.ld, &w.asT));
        forward_map_fwd_chain(c, c->F_t, w.Xnoa.p, w.Xnoa.ld, w.Xnoz.p, w.Xnoz.ld, w.fsT, B, ch[0]);
        forward_map_fwd_chain(c, c->F_p, w.Xoa.p, w.Xoa.ld, w.Xoz.p, w.Xoz.ld, w.fsO, B, ch[1]);
        backward_map_fwd_chain(c, c->K_p, w.goal2.p, w.goal2.ld, w.bsS, 2 * B, ch[2]);
        actor_fwd_chain(c, c->A_p, w.Xo.p, w.Xo.ld, w.Xoz.p, w.Xoz.ld, w.as, B, ch[3], !fused_policy);
        ch[3].push_back(policy_stage(w.so.eps_actor, w.as.mu.p, w.Xopi.p + aoff, w.Xopi.ld, &w.as));
        prog_parallel(prog, ch);
    }
    const float* phi = w.bsS.Bm.p;                                    // phi(goal)       rows [0, B)
    const float* nphi = w.bsS.Bm.p + (size_t)B * Lz;                  // phi(next_goal)  rows [B, 2B)
    float* dphi = w.dBm2.p;
    float* dnphi = w.dBm2.p + (size_t)B * Lz;
    // ---- critic loss (sf.py:607-626) -> dF1, dF2
    POST_BEGIN
    HIPCK(c, launch_sf_loss(w.fsO.F1.p, w.fsO.F2.p, w.fsT.F1.p, w.fsT.F2.p, nphi, w.z.p, Lz, w.disc, hp.q_loss, w.dF1.p, w.dF2.p,
                            w.metrics, w.pw_scratch, B, z, s));
    POST_END
    // ---- feature loss and its gradient wrt [phi ; next_phi]  (sf.py:628 -> ICM :203-213 / Laplacian :100-116), then both backward
    // passes side by side: successor_net from (dF1, dF2), feature_learner from d[phi ; next_phi]
    Chain feat;
    if (d.sf == 1) {
        const IcmP &I = c->I_p, &G = c->I_g;
        const int Kc = pad32(2 * z);
        feat.push_back([=, &w](Ops& o2) {
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, launch_concat2(w.icat.p, Kc, phi, Lz, z, nphi, Lz, z, B, q));
                return (int)FBHIP_OK;
            });
        });
        feat.push_back([=, &w](Ops& o2) { o2.gemms.push_back(P(w.icat.p, Kc, 1, I.W1, Kc, 1, w.ih1.p, Lb, B, Lb, Kc, I.b1, EPI_BIAS_RELU)); });
        feat.push_back([=, &w](Ops& o2) { o2.gemms.push_back(P(w.ih1.p, Lb, 1, I.W2, Lb, 1, w.ih2.p, Lb, B, Lb, Lb, I.b2, EPI_BIAS_RELU)); });
        feat.push_back([=, &w](Ops& o2) {
            o2.gemms.push_back(P(w.ih2.p, Lb, 1, I.W3, Lb, 1, w.ipre.p, La, B, a, Lb, I.b3, EPI_BIAS));
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, launch_icm_loss(w.ipre.p, La, w.Xoa.p + aoff, w.Xoa.ld, w.d_ipre.p, La, B, a, w.metrics, w.pw_scratch, q));
                return (int)FBHIP_OK;
            });
        });
        feat.push_back([=, &w](Ops& o2) {
            o2.gemms.push_back(P(w.d_ipre.p, La, 0, w.ih2.p, Lb, 0, G.W3, Lb, a, Lb, B, nullptr, EPI_NONE, nullptr, 0, G.b3));
            o2.gemms.push_back(P(w.d_ipre.p, La, 1, I.W3, Lb, 0, w.d_ih2.p, Lb, B, Lb, a, nullptr, EPI_MASK_RELU, w.ih2.p, Lb));
        });
        feat.push_back([=, &w](Ops& o2) {
            o2.gemms.push_back(P(w.d_ih2.p, Lb, 0, w.ih1.p, Lb, 0, G.W2, Lb, Lb, Lb, B, nullptr, EPI_NONE, nullptr, 0, G.b2));
            o2.gemms.push_back(P(w.d_ih2.p, Lb, 1, I.W2, Lb, 0, w.d_ih1.p, Lb, B, Lb, Lb, nullptr, EPI_MASK_RELU, w.ih1.p, Lb));
        });
        feat.push_back([=, &w](Ops& o2) {
            o2.gemms.push_back(P(w.d_ih1.p, Lb, 0, w.icat.p, Kc, 0, G.W1, Kc, Lb, Kc, B, nullptr, EPI_NONE, nullptr, 0, G.b1));
            o2.gemms.push_back(P(w.d_ih1.p, Lb, 1, I.W1, Kc, 0, dphi, Lz, B, z, Lb));            // d cat[:, :z]
            o2.gemms.push_back(P(w.d_ih1.p, Lb, 1, I.W1 + z, Kc, 0, dnphi, Lz, B, z, Lb));       // d cat[:, z:2z]
        });
    } else {
        feat.push_back([=, &w](Ops& o2) {
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                // orthonormality part: the pairwise kernel with zero F panels leaves 2 Hm . phi in d phi and orth_loss in the
                // metrics; lap_kernel adds the mean((phi - next_phi)^2) part and writes d next_phi
                HIPCK(c, launch_pairwise_fb(w.zeroF.p, w.zeroF.p, phi, w.zeroF.p, w.zeroF.p, phi, w.disc, B, z, Lz, 1.0f, w.lapS1.p,
                                            w.lapS2.p, dphi, w.metrics, w.pw_scratch, q));
                HIPCK(c, launch_lap(phi, nphi, Lz, dphi, dnphi, w.metrics, w.pw_scratch, B, z, q));
                return (int)FBHIP_OK;
            });
        });
    }
    BGrad bg{w.dBm2.p, w.dy2.p, w.s_dr2.p, w.s_dt1.p};
    backward_map_bwd_chain(c, c->K_p, c->K_g, w.goal2.p, w.goal2.ld, w.bsS, 2 * B, feat, false, &bg);
    Chain succ;
    forward_map_bwd_chain(c, c->F_p, c->F_g, w.Xoa.p, w.Xoa.ld, w.Xoz.p, w.Xoz.ld, w.fsO, B, succ);
    {
        std::vector<Chain> ch{succ, feat};
        prog_parallel(prog, ch);
    }
    // ---- sf_opt.step() + phi_opt.step() (sf.py:643-653): one pass over forward ++ backward, lr | lr_coef * lr; the EMA of
    // successor_target_net (sf.py:751-752) rides along like FBDDPGAgent's (nothing reads a target before the next update)
    POST_BEGIN
    HIPCK(c, launch_step_advance(w.st, 0, s));
    const int64_t nf = c->L[FBHIP_NET_FORWARD].numel, nb = c->L[FBHIP_NET_BACKWARD].numel;
    HIPCK(c, launch_adam_ema(c->fb_p, c->fb_g, c->fb_m, c->fb_v, c->fb_t, nf + nb, hp.lr, hp.lr_coef * hp.lr, nf, hp.grad_scale,
                             hp.fb_target_tau, w.st, 0, 0, s));
    POST_END
    return FBHIP_OK;
}
#undef POST_BEGIN
#undef POST_END

int enqueue_update(fbhip_ctx* c, const fbhip_hparams& hp, const fbhip_inject* inj, int mask, hipStream_t s) {
    Program prog;
    if (c->d.sf) {
        if (mask != FBHIP_PHASE_ALL) { c->err = g_err = "fbhip: dims.sf runs complete updates only (phase_mask = FBHIP_PHASE_ALL)"; return FBHIP_E_INVALID; }
        RC(build_update_sf(c, hp, inj, prog));
        RC(build_update(c, hp, nullptr, FBHIP_PHASE_ACTOR_GRAD | FBHIP_PHASE_ACTOR_STEP, prog));      // sf.py:666-694
        return run_program(c, prog, s);
    }
    RC(build_update(c, hp, inj, mask, prog));
    return run_program(c, prog, s);
}

int check_hparams(fbhip_ctx* c, const fbhip_hparams* hp) {
    if (!hp) { c->err = g_err = "fbhip: null hparams"; return FBHIP_E_INVALID; }
    if (hp->struct_size != sizeof(fbhip_hparams)) {
        c->err = g_err = "fbhip: fbhip_hparams.struct_size is " + std::to_string(hp->struct_size) + ", this library expects " +
                         std::to_string(sizeof(fbhip_hparams)) + " (caller built against another include/fbhip.h?)";
        return FBHIP_E_INVALID;
    }
    return FBHIP_OK;
}

int need_bound(fbhip_ctx* c, bool replay) {
    if (!c) { g_err = "fbhip: null context"; return FBHIP_E_INVALID; }
    if (!c->bound) { c->err = g_err = "fbhip: buffers not bound (fbhip_bind_buffers)"; return FBHIP_E_STATE; }
    if (replay && !c->replay_bound) { c->err = g_err = "fbhip: replay storage not bound (fbhip_replay_bind)"; return FBHIP_E_STATE; }
    return FBHIP_OK;
}

}  // namespace

// =================================================================================================== C ABI
extern "C" {

int fbhip_abi_version(void) { return FBHIP_ABI_VERSION; }

const char* fbhip_last_error(const fbhip_ctx* ctx) { return (ctx && !ctx->err.empty()) ? ctx->err.c_str() : g_err.c_str(); }

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
    return 2 * carve(*dims, nullptr).total_bytes;            // two complete sets (fbhip_update_many pipelines consecutive steps)
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
        hipHostMalloc((void**)&c->h_out, 64 * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();             // no usable device here (e.g. the CPU-only build check): fast path disabled
        c->h_in = c->h_out = nullptr;
    } else {
        memset(c->h_in, 0, act_in_floats(*dims) * sizeof(float));
    }
    *out = c;
    return FBHIP_OK;
}

int fbhip_destroy(fbhip_ctx* ctx) {
    if (!ctx) return FBHIP_OK;
    for (auto& g : ctx->graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto& g : ctx->infer_graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto e : ctx->events) (void)hipEventDestroy(e);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    delete ctx;
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
    if (workspace_bytes < need) { c->err = g_err = "fbhip: workspace too small"; return FBHIP_E_INVALID; }
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
    if (has_actor) { c->A_p = act_p(actor_params, c->L[2]); c->A_g = act_p(actor_grads, c->L[2]); }
    for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
    c->graphs.clear();
    for (auto& g : c->infer_graphs) (void)hipGraphExecDestroy(g.exec);
    c->infer_graphs.clear();
    HIPCK(c, pairwise_prepare(c->d.batch, c->d.z_dim));
    HIPCK(c, gemm_init());
    HIPCK(c, inverse_prepare());
    if (has_actor) HIPCK(c, actor_head_bwd_prepare(c->d.hidden_dim, c->d.action_dim));
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

int fbhip_update(fbhip_ctx* c, const fbhip_hparams* hp, const fbhip_inject* inject, int32_t phase_mask,
                 int32_t use_graph, void* stream) {
    RC(need_bound(c, (phase_mask & FBHIP_PHASE_SAMPLE) != 0));
    RC(check_hparams(c, hp));
    hipStream_t s = (hipStream_t)stream;
    if (!use_graph) return enqueue_update(c, *hp, inject, phase_mask, s);
    for (auto& g : c->graphs) {
        if (g.n_steps == 1 && g.set == c->cur && g.mask == phase_mask && memcmp(&g.hp, hp, sizeof(*hp)) == 0 && g.has_inj == (inject != nullptr) &&
            (!inject || memcmp(&g.inj, inject, sizeof(*inject)) == 0)) {
            HIPCK(c, hipGraphLaunch(g.exec, s));
            return FBHIP_OK;
        }
    }
    hipGraph_t graph = nullptr;
    HIPCK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_update(c, *hp, inject, phase_mask, s);
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != FBHIP_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    HIPCK(c, e);
    GraphEntry ge{};
    ge.mask = phase_mask; ge.hp = *hp; ge.has_inj = inject != nullptr; ge.n_steps = 1; ge.set = c->cur;
    if (inject) ge.inj = *inject;
    e = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIPCK(c, e);
    if (c->graphs.size() >= 16) { (void)hipGraphExecDestroy(c->graphs.front().exec); c->graphs.erase(c->graphs.begin()); }
    c->graphs.push_back(ge);
    HIPCK(c, hipGraphLaunch(ge.exec, s));
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
                            bool dp = false) {
    RC(need_bound(c, true));
    RC(check_hparams(c, hp));
    if (n_steps < 1 || n_steps > 64) { c->err = g_err = "fbhip_update_many: bad argument"; return FBHIP_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    for (auto& g : c->graphs) {
        if (g.n_steps == n_steps && g.set == c->cur && g.mask == (FBHIP_PHASE_ALL | (dp ? DP_GRAPH_BIT : 0)) && g.has_inj == (injs != nullptr) &&
            (!injs || memcmp(&g.inj, injs, sizeof(*injs)) == 0) && memcmp(&g.hp, hp, sizeof(*hp)) == 0) {
            HIPCK(c, hipGraphLaunch(g.exec, s));
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
    static const bool pipelined = [] { const char* e = getenv("FBHIP_UPDATE_PIPELINE"); return !(e && e[0] == '0'); }();
    const bool pipe = pipelined && n_steps > 1 && !c->d.discrete && !c->d.sf;   // (discrete: no actor phase to overlap with)
    if (pipe) {
        if (!c->side) HIPCK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        while ((int)c->events.size() < 2 * 64) {
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
    auto allreduce = [&](int which) -> int {                     // the peers' gradients, summed in place (peer.hip)
        HIPCK(c, launch_peer_allreduce(c->peers, which, which == 0 ? n_fb : n_ac, s));
        return (int)FBHIP_OK;
    };
    if (dp && !pipe) {
        for (int i = 0; i < n_steps && rc == FBHIP_OK; ++i) {
            rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_SAMPLE | FBHIP_PHASE_FB_GRAD | FBHIP_PHASE_ACTOR_FWD, s);
            if (rc == FBHIP_OK) rc = allreduce(0);
            if (rc == FBHIP_OK) rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_FB_STEP | FBHIP_PHASE_ACTOR_GRAD, s);
            if (rc == FBHIP_OK && has_actor) rc = allreduce(1);
            if (rc == FBHIP_OK) rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_ACTOR_STEP, s);
        }
    } else if (dp) {
        // the single-rank pipeline below with the optimiser steps cut off their phases and the two all-reduces in the cuts:
        //   [target chain | FB backward | actor fwd] -> AR(fb) -> FB step -> fork [next head] || [actor grad -> AR(actor) -> actor step] -> join
        const int HEAD = FBHIP_PHASE_SAMPLE | FBHIP_PHASE_FB_FWD_ONLINE;
        const int cur0 = c->cur;
        rc = enqueue_update(c, *hp, nullptr, HEAD, s);
        for (int i = 0; i < n_steps && rc == FBHIP_OK && he == hipSuccess; ++i) {
            rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_FB_FWD_TARGET | FBHIP_PHASE_FB_BWD | FBHIP_PHASE_ACTOR_FWD, s);
            if (rc == FBHIP_OK) rc = allreduce(0);
            if (rc == FBHIP_OK) rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_FB_STEP, s);
            if (rc != FBHIP_OK) break;
            const bool more = i + 1 < n_steps;
            if (more) {
                if ((he = hipEventRecord(c->events[2 * i], s)) != hipSuccess) break;
                if ((he = hipStreamWaitEvent(c->side, c->events[2 * i], 0)) != hipSuccess) break;
                c->cur ^= 1;
                rc = enqueue_update(c, *hp, nullptr, HEAD, c->side);
                c->cur ^= 1;
                if (rc != FBHIP_OK) break;
            }
            rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_ACTOR_GRAD, s);
            if (rc == FBHIP_OK) rc = allreduce(1);
            if (rc == FBHIP_OK) rc = enqueue_update(c, *hp, nullptr, FBHIP_PHASE_ACTOR_STEP, s);
            if (more) {
                if ((he = hipEventRecord(c->events[2 * i + 1], c->side)) != hipSuccess) break;
                if ((he = hipStreamWaitEvent(s, c->events[2 * i + 1], 0)) != hipSuccess) break;
                c->cur ^= 1;
            }
        }
        c->cur = cur0;
    } else if (!pipe) {
        for (int i = 0; i < n_steps && rc == FBHIP_OK; ++i) rc = enqueue_update(c, *hp, injs ? &injs[i] : nullptr, FBHIP_PHASE_ALL, s);
    } else {
        const int HEAD = FBHIP_PHASE_SAMPLE | FBHIP_PHASE_FB_FWD_ONLINE;
        const int MID = FBHIP_PHASE_FB_FWD_TARGET | FBHIP_PHASE_FB_BWD | FBHIP_PHASE_ACTOR_FWD | FBHIP_PHASE_FB_STEP;   // (both FB_BWD bits)
        const int TAIL = FBHIP_PHASE_ACTOR_GRAD | FBHIP_PHASE_ACTOR_STEP;
        const int cur0 = c->cur;
        rc = enqueue_update(c, *hp, injs ? &injs[0] : nullptr, HEAD, s);
        for (int i = 0; i < n_steps && rc == FBHIP_OK && he == hipSuccess; ++i) {
            rc = enqueue_update(c, *hp, nullptr, MID, s);
            if (rc != FBHIP_OK) break;
            const bool more = i + 1 < n_steps;
            if (more) {                          // fork: the next step's head on the twin workspace set
                if ((he = hipEventRecord(c->events[2 * i], s)) != hipSuccess) break;
                if ((he = hipStreamWaitEvent(c->side, c->events[2 * i], 0)) != hipSuccess) break;
                c->cur ^= 1;
                rc = enqueue_update(c, *hp, injs ? &injs[i + 1] : nullptr, HEAD, c->side);
                c->cur ^= 1;
                if (rc != FBHIP_OK) break;
            }
            rc = enqueue_update(c, *hp, nullptr, TAIL, s);
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
    GraphEntry ge{};
    ge.mask = FBHIP_PHASE_ALL | (dp ? DP_GRAPH_BIT : 0); ge.hp = *hp; ge.has_inj = injs != nullptr; ge.n_steps = n_steps; ge.set = c->cur;
    if (injs) ge.inj = injs[0];              // (cache key: a caller that reuses its per-step buffers replays the same graph)
    e = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIPCK(c, e);
    if (c->graphs.size() >= 16) { (void)hipGraphExecDestroy(c->graphs.front().exec); c->graphs.erase(c->graphs.begin()); }
    c->graphs.push_back(ge);
    HIPCK(c, hipGraphLaunch(ge.exec, s));
    return FBHIP_OK;
}

int fbhip_update_many(fbhip_ctx* c, const fbhip_hparams* hp, int32_t n_steps, void* stream) {
    return update_many_impl(c, hp, n_steps, nullptr, stream);
}

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
    if (c->d.sf) { c->err = g_err = "fbhip_dp_bind_peers: dims.sf runs on a single rank"; return FBHIP_E_INVALID; }
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
    if (c->peers.world < 2) { c->err = g_err = "fbhip_update_many_dp: no peers bound (fbhip_dp_bind_peers)"; return FBHIP_E_STATE; }
    return update_many_impl(c, hp, n_steps, nullptr, stream, /*dp=*/true);
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
    const size_t nin = act_noise_off(d) + (has_noise ? a : 0);
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
                             c->seed, c->rank, w.st, w.act_out, c->sq, s));
    HIPCK(c, hipMemcpyAsync(c->h_out, w.act_out, (size_t)a * sizeof(float), hipMemcpyDeviceToHost, s));
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
    HIPCK(c, launch_zcorrel(y, w.act_in + act_z_off(d), z, d.norm_z, w.act_out, s));
    HIPCK(c, hipMemcpyAsync(c->h_out, w.act_out, sizeof(float), hipMemcpyDeviceToHost, s));
    return FBHIP_OK;
}

// replay (or capture + replay) the graph of one fast-path call, then wait for its result
int run_infer_graph(fbhip_ctx* c, int kind, float stddev, int eval_mode, bool has_noise, hipStream_t s) {
    hipGraphExec_t exec = nullptr;
    for (auto& g : c->infer_graphs)
        if (g.kind == kind && g.eval_mode == eval_mode && g.has_noise == (int)has_noise && g.stddev == stddev) exec = g.exec;
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

int fbhip_actor_loss(const float* F1, const float* F2, int32_t ldf, const float* z, int32_t ldz, const float* mu,
                     int32_t ldmu, const float* action, int32_t lda, float stddev, float* dF1, float* dF2, float* metrics,
                     float* scratch, int32_t rows, int32_t d, int32_t a, void* stream) {
    fbhip_ctx* none = nullptr;
    HIPCK(none, launch_actor_loss(F1, F2, ldf, z, ldz, mu, ldmu, action, lda, stddev, dF1, dF2, metrics, scratch, rows, d, a,
                                  (hipStream_t)stream));
    return FBHIP_OK;
}

}  // extern "C"
