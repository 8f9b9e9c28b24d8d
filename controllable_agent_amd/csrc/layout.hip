// Flat-buffer layout of the nets (names and order of the reference's state_dict(), physical padding), dimension checks and
// the carving of the caller's workspace: the host-side bookkeeping of libfbhip.so that involves no launch.
#include "host.h"

namespace fbhip {
namespace host {

thread_local std::string g_err;



// ------------------------------------------------------------------------------------------------ layout


struct LayoutBuilder {
    NetLayout L;
    int64_t cur = 0;
    // logical [rows x cols]; physical leading dimension ld (>= cols) and phys_rows (>= rows) allocated
    // (weight matrices start on 128-byte boundaries: a matrix whose leading dimension is a multiple of 32 then consists of whole
    // P3 blocks, p3.h; ``follow``: directly behind the previous matrix -- F2.0.weight behind F1.0.weight, one stacked GEMM operand)
    void mat(const std::string& n, int rows, int cols, int ld = 0, int phys_rows = 0, bool follow = false) {
        if (!follow) cur = (cur + 31) & ~(int64_t)31;
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
        L.numel = (cur + 31) & ~(int64_t)31;      // whole 128-byte blocks: the next net of a shared flat buffer stays aligned
        return L;
    }
};

// Front-end geometry of ForwardMap / Actor (fb_modules.py:90-103, 165-178).  preprocess (default): TWO LayerNorm branches
// (in -> H -> Fd) whose outputs are concatenated (2 Fd wide), optionally followed by a Linear(2Fd, H) + ReLU trunk
// (add_trunk).  preprocess == 0: ONE LayerNorm branch on the concatenated input with Fd := H, always followed by the
// trunk's last Linear(H, H) + ReLU -- the same pipeline with one branch.
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
// DiscreteFBAgent (dims.discrete, discrete_fb.py:52-101): action_dim is A, the NUMBER of actions.  The ForwardMap has no action
// input (one trunk on [obs|z]) and its heads emit one embedding per action: z * A outputs, (k, a) at column k * A + a.

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
        b.mat("F1.0.weight", H, gm.feat); b.mat("F2.0.weight", H, gm.feat, 0, 0, /*follow=*/true);
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
        int hin = 0, hout = 0;
        const char* pfx = nullptr;
        if (sf_head_dims(d, &hin, &hout, &pfx)) {      // e.g. ICM: inverse_dynamic_net = mlp(2 z, Hb, 'irelu', Hb, 'irelu', a, 'tanh')  (sf.py:198)
            const std::string i = pfx;
            b.mat(i + "0.weight", Hb, hin, pad32(hin), HbP); b.vec(i + "0.bias", Hb, HbP);
            b.mat(i + "2.weight", Hb, Hb, HbP, HbP); b.vec(i + "2.bias", Hb, HbP);
            b.mat(i + "4.weight", hout, Hb, HbP); b.vec(i + "4.bias", hout);
            append(order, {i + "0.weight", i + "0.bias", i + "2.weight", i + "2.bias", i + "4.weight", i + "4.bias"});
        }
        if (d.sf == 6 || d.sf >= 8) {      // (contrastive / contrastivev2, sf.py:121, 162: the same modules + the projection) SVDP: mu_net = mlp(g + a, Hb, "ntanh", Hb, "relu", z)  (sf.py:340; BackwardMap.B's modules, no projection);
            const std::string m = "mu_net.";   // SVDSR: the same on the goal alone (sf.py:268)
            const int mi = d.sf == 6 ? g + a : g;
            b.mat(m + "0.weight", Hb, mi, pad32(mi), HbP); b.vec(m + "0.bias", Hb, HbP); b.vec(m + "1.weight", Hb, HbP);
            b.vec(m + "1.bias", Hb, HbP);
            b.mat(m + "3.weight", Hb, Hb, HbP, HbP); b.vec(m + "3.bias", Hb, HbP);
            b.mat(m + "5.weight", z, Hb, HbP); b.vec(m + "5.bias", z);
            append(order, {m + "0.weight", m + "0.bias", m + "1.weight", m + "1.bias", m + "3.weight", m + "3.bias", m + "5.weight", m + "5.bias"});
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
    if (d->sf < 0 || d->sf > 12) { g_err = "fbhip: dims.sf must be 0 or 1..12 (icm, lap, random, autoencoder, transition, svd_p, latent, svd_sr, svd_srv2, contrastive, contrastivev2, identity)"; return FBHIP_E_INVALID; }
    if (d->sf == 12 && d->z_dim != d->goal_dim) { g_err = "fbhip: dims.sf = 12 (identity features) needs z_dim == goal_dim"; return FBHIP_E_INVALID; }
    if (d->sf && (d->discrete || d->boltzmann || !d->norm_z)) { g_err = "fbhip: dims.sf needs discrete = 0, boltzmann = 0, norm_z = 1"; return FBHIP_E_INVALID; }
    if (d->backward_identity && (d->sf || d->z_dim != d->goal_dim)) { g_err = "fbhip: backward_identity (cfg.debug) needs sf = 0 and z_dim == goal_dim"; return FBHIP_E_INVALID; }
    if (!d->use_goal && d->goal_dim != d->obs_dim) { g_err = "fbhip: goal_dim must equal obs_dim when use_goal == 0"; return FBHIP_E_INVALID; }
    return FBHIP_OK;
}

// ------------------------------------------------------------------------------------------------ workspace


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
        // [goal ; next_goal] -- contrastivev2 also runs feature_net on the hindsight goals: a third block of rows
        w.goal2 = c.buf((d.sf == 11 ? 3 : 2) * B, g, pad32(g));
        w.bin = w.goal2; w.bin.rows = B;
        w.next_goal = w.bin; w.next_goal.p = base ? w.goal2.p + (size_t)B * w.goal2.ld : nullptr;
        if (d.sf == 11) { w.fgoal = w.bin; w.fgoal.p = base ? w.goal2.p + (size_t)2 * B * w.goal2.ld : nullptr; }
        else w.fgoal = c.buf(d.sf == 10 ? B : 1, g, pad32(g));      // (contrastive reads the hindsight goal of every row)
        w.pgoal = c.buf(B, g, pad32(g));                            // next_goal[perm]: the z-mix of sf.py:726-727
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
    w.ln_partials_b = c.f((size_t)(((d.sf == 11 ? 3 : d.sf ? 2 : 1) * B + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK) * 2 * nmax);
    w.pw_scratch = c.f(pairwise_scratch_floats(B, z));
    w.splitk = c.f((size_t)6 << 20);
    w.rw = c.f((size_t)B * B); w.rw_u = c.f(B); w.ymixw = c.buf(B, z);
    if (d.sf) {
        const int Lb = pad64(Hb), Lz = pad4(z), La = pad4(a);
        const int RB = (d.sf == 11 ? 3 : 2) * B;              // rows of the feature pass
        w.bsS.pre1 = c.buf(RB, Hb, Lb); w.bsS.t1 = c.buf(RB, Hb, Lb); w.bsS.r2 = c.buf(RB, Hb, Lb);
        w.bsS.y = c.buf(RB, z); w.bsS.Bm = c.buf(RB, z);
        w.bsS.stats = c.f(2 * (size_t)RB); w.bsS.norms = c.f((size_t)RB);
        w.dBm2 = c.buf(RB, z); w.dy2 = c.buf(RB, z); w.s_dr2 = c.buf(RB, Hb, Lb); w.s_dt1 = c.buf(RB, Hb, Lb);
        if (d.sf == 2 || d.sf == 6 || d.sf == 8 || d.sf == 9) { w.zeroF = c.buf(B, z); w.lapS1 = c.buf(B, z); w.lapS2 = c.buf(B, z); }
        if (d.sf == 8 || d.sf == 9) { w.c99 = c.f(B); w.dphi_o = c.buf(B, z); }
        if (d.sf == 10 || d.sf == 11) w.dmu_y = c.buf(B, z);
        if (d.sf == 6 || d.sf >= 8) {
            w.Xga = c.buf(B, g + a, pad32(g + a));
            w.dmu = c.buf(B, z); w.m_dr2 = c.buf(B, Hb, Lb); w.m_dt1 = c.buf(B, Hb, Lb);
            w.ln_partials_m = c.f((size_t)((B + LN_BWD_ROWS_PER_BLOCK - 1) / LN_BWD_ROWS_PER_BLOCK) * 2 * (size_t)(H > Hb ? H : Hb));
        }
        int hin = 0, hout = 0;
        const char* pfx = nullptr;
        if (sf_head_dims(d, &hin, &hout, &pfx)) {
            w.icat = c.buf(B, hin, pad32(hin)); w.ih1 = c.buf(B, Hb, Lb); w.ih2 = c.buf(B, Hb, Lb);
            w.ipre = c.buf(B, hout, pad4(hout)); w.d_ipre = c.buf(B, hout, pad4(hout)); w.d_ih1 = c.buf(B, Hb, Lb); w.d_ih2 = c.buf(B, Hb, Lb);
        }
        (void)Lz; (void)La;
    }
    w.act_in = c.f(act_in_floats(d));
    w.act_vec = c.f((size_t)5 * 2048 + 256);
    w.act_out = c.f(64);
    w.total_bytes = (c.cur + 255) & ~(size_t)255;
    return w;
}

// ------------------------------------------------------------------------------------------------ weights

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
bool sf_head_dims(const fbhip_dims& d, int* in, int* out, const char** prefix) {
    switch (d.sf) {
        case 1: *in = 2 * d.z_dim; *out = d.action_dim; *prefix = "inverse_dynamic_net."; return true;
        case 4: *in = d.z_dim; *out = d.goal_dim; *prefix = "decoder."; return true;
        case 5: *in = d.z_dim + d.action_dim; *out = d.goal_dim; *prefix = "forward_dynamic_net."; return true;
        case 7: *in = d.z_dim + d.action_dim; *out = d.z_dim; *prefix = "forward_dynamic_net."; return true;
        default: return false;
    }
}
BwdP mu_p(float* base, const NetLayout& L) {
    BwdP b{};
    if (!L.by_name.count("mu_net.0.weight")) return b;
    const std::string q = "mu_net.";
    b.W1 = base + L.by_name.at(q + "0.weight").off; b.b1 = base + L.by_name.at(q + "0.bias").off;
    b.g1 = base + L.by_name.at(q + "1.weight").off; b.be1 = base + L.by_name.at(q + "1.bias").off;
    b.W2 = base + L.by_name.at(q + "3.weight").off; b.b2 = base + L.by_name.at(q + "3.bias").off;
    b.W3 = base + L.by_name.at(q + "5.weight").off; b.b3 = base + L.by_name.at(q + "5.bias").off;
    return b;
}
IcmP icm_p(float* base, const NetLayout& L) {
    IcmP i;
    std::string q;
    for (const char* cand : {"inverse_dynamic_net.", "decoder.", "forward_dynamic_net."})
        if (L.by_name.count(std::string(cand) + "0.weight")) q = cand;
    if (q.empty()) return i;
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



}  // namespace host
}  // namespace fbhip
