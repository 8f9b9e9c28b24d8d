// The launch schedule of one update on gfx950: grouped-GEMM launch policy, round-based merged scheduling, the network passes as
// chains of stages, and the programs of FBDDPGAgent.update / DiscreteFBAgent.update (build_update) and SFAgent.update
// (build_update_sf).  Everything here only ENQUEUES kernels on the caller's stream (no allocation, no synchronisation), so a
// whole update is hipGraph-capturable (api.hip captures and replays).
#include "host.h"

namespace fbhip {
namespace host {

struct BGrad { const float* dBm; float* dy; float* dr2; float* dt1; float* ln_partials = nullptr; };




GemmProblem P(const float* A, int lda, int akc, const float* B, int ldb, int bkc, float* C, int ldc, int M, int N,
              int K, const float* bias, int epi, const float* aux, int ldaux, float* colsum) {
    GemmProblem p{};
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.aux = aux; p.colsum = colsum;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
    p.a_kcontig = akc; p.b_kcontig = bkc; p.epi = epi;
    return p;
}

// scratch for split-K partials (launches of one update are stream-ordered, so one slab serves them all); standalone
// fbhip_gemm (ctx == nullptr) never splits

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
    constexpr long dma128_min = 2048;
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
            constexpr long small_split_max = 160;       // (anywhere in 0..256 measured the same in round 1)
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
            if (p.kslices > 1) p.xgm = p.xgn = 0;            // (K-sliced problems keep the slice-major band order)
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
    for (size_t i = 0; i < o.heads.size(); i += HEAD_MAX_GROUP) {
        HeadGroup g{};
        for (size_t j = i; j < o.heads.size() && j < i + HEAD_MAX_GROUP; ++j) g.p[g.n++] = o.heads[j];
        HIPCK(c, launch_head_group(g, s));
    }
    RC(flush_colreduce(c, s));
    for (size_t i = 0; i < o.lnf.size(); i += LN_MAX_GROUP) {
        LnFwdGroup g{};
        for (size_t j = i; j < o.lnf.size() && j < i + LN_MAX_GROUP; ++j) g.p[g.n++] = o.lnf[j];
        HIPCK(c, launch_ln_tanh_fwd_group(g, s));
    }
    for (size_t i = 0; i < o.lnb.size(); i += LN_MAX_GROUP) {
        LnBwdGroup g{};
        for (size_t j = i; j < o.lnb.size() && j < i + LN_MAX_GROUP; ++j) g.p[g.n++] = o.lnb[j];
        HIPCK(c, launch_ln_tanh_bwd_group(g, s, &c->cr_pending));      // (d gamma, d beta) folds ride in the next reduce launch
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
                           int rows, Chain& out, bool with_heads, int disc_mode, const float* disc_z, int disc_ldz,
                           int oa_base_k, int part) {
    const fbhip_dims& d = c->d;
    const Geom gm = geom_of(d);
    const int H = d.hidden_dim, z = d.z_dim, Lz = pad4(z), Fo = gm.Fo, hw = gm.hw, feat = gm.feat;
    FSet* Sp = &S;
    if (part != 2) {
        out.push_back([=](Ops& o) {
            o.gemms.push_back(P(Xa, lda, 1, W.oa.W1, W.oa.ld1, 1, Sp->pre1a.p, H, rows, H, oa_base_k > 0 ? oa_base_k : W.oa.ld1, W.oa.b1,
                                EPI_BIAS));
            if (!gm.single) o.gemms.push_back(P(Xz, ldz, 1, W.oz.W1, W.oz.ld1, 1, Sp->pre1z.p, H, rows, H, W.oz.ld1, W.oz.b1, EPI_BIAS));
        });
        out.push_back([=](Ops& o) {
            if (oa_base_k <= 0) o.lnf.push_back(LnFwdProblem{Sp->pre1a.p, H, W.oa.g1, W.oa.be1, Sp->t1a.p, H, Sp->statsA, rows, H, 0, 0, 0, H});
            if (!gm.single) o.lnf.push_back(LnFwdProblem{Sp->pre1z.p, H, W.oz.g1, W.oz.be1, Sp->t1z.p, H, Sp->statsZ, rows, H, 0, 0, 0, H});
        });
    }
    if (part == 1) {
        if (!gm.single)
            out.push_back([=](Ops& o) {
                o.gemms.push_back(P(Sp->t1z.p, H, 1, W.oz.W2, H, 1, Sp->h.p + Fo, hw, rows, Fo, H, W.oz.b2, EPI_BIAS_RELU));
            });
        return;
    }
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(Sp->t1a.p, H, 1, W.oa.W2, H, 1, Sp->h.p, hw, rows, Fo, H, W.oa.b2, EPI_BIAS_RELU));
        if (!gm.single && part != 2) o.gemms.push_back(P(Sp->t1z.p, H, 1, W.oz.W2, H, 1, Sp->h.p + Fo, hw, rows, Fo, H, W.oz.b2, EPI_BIAS_RELU));
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
        // the embedding heads' output layer (K = H, N = z): one head_kernel launch for all of a round's heads when z <= 64
        // (no split-K slab, no reduce launch), else two problems of the round's grouped GEMM
        const HeadProblem h1{Sp->p.p, 2 * H, W.W4[0], H, W.b4[0], Sp->F1.p, Lz, nullptr, 0, nullptr, 0.f, rows, z, H};
        const HeadProblem h2{Sp->p.p + H, 2 * H, W.W4[1], H, W.b4[1], Sp->F2.p, Lz, nullptr, 0, nullptr, 0.f, rows, z, H};
        if (head_ok(h1) && head_ok(h2)) { o.heads.push_back(h1); o.heads.push_back(h2); return; }
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
                            bool with_projection, int in_dim) {
    const fbhip_dims& d = c->d;
    // GEMMs run on the padded width Lb = pad64(Hb) (zero weight rows / columns), LayerNorm on the logical Hb
    const int g = in_dim > 0 ? in_dim : d.goal_dim, Hb = d.backward_hidden_dim, Lb = pad64(Hb), z = d.z_dim, Lz = pad4(z);
    // workspace panels have >= pad32(g) finite columns per row (the weight's pad columns are zero, so whatever sits
    // there contributes nothing); arbitrary caller tensors are read on their logical width
    const bool in_ws = (const char*)X >= c->ws_lo && (const char*)X < c->ws_lo + c->ws_bytes;
    const int Kg = (in_ws && ldx >= pad32(g)) ? pad32(g) : g;
    BSet* Sp = &S;
    if (d.backward_identity) {                   // cfg.debug: IdentityMap.forward (fb_modules.py:207-208) -- the map's output, projected or
        out.push_back([=](Ops& o) {               // not, is its input (z_dim == goal_dim): both output panels of the set get the copy
            o.post.push_back([=](hipStream_t q) -> int {
                HIPCK(c, launch_concat2(Sp->y.p, Lz, X, ldx, z, nullptr, 0, 0, rows, q));
                HIPCK(c, launch_concat2(Sp->Bm.p, Lz, X, ldx, z, nullptr, 0, 0, rows, q));
                return (int)FBHIP_OK;
            });
        });
        return;
    }
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
        // cfg.norm_z == False: the map's output IS y (fb_modules.py:228-229); callers read ``bm_of(set)``
        const bool proj = with_projection && d.norm_z;
        // (head_kernel with the projection as its epilogue was built for this layer too, 8.8 us for the three passes of a step
        // against 10.0 + 7.0 + 7.5; not used: its summation order moves the q_loss of the tiny_goal trace -- B^T B inverted -- to
        // 2.1e-5 of the reference's value, against the stated 2e-5, and the passes are off the step's critical path: round 5 measured
        // the step with it, 1221.0 vs 1221.7 update-steps/s, profiles/r05l_bhead_ab.txt)
        o.gemms.push_back(P(Sp->r2.p, Lb, 1, W.W3, Lb, 1, Sp->y.p, Lz, rows, z, Lb, W.b3, EPI_BIAS));
        if (proj) o.l2n.push_back(L2Problem{Sp->y.p, Lz, Sp->Bm.p, Lz, Sp->norms, rows, z, sqrtf((float)z)});
    });
}

int backward_map_fwd(fbhip_ctx* c, const BwdP& W, const float* X, int ldx, BSet& S, int rows, hipStream_t s) {
    Chain ch;
    backward_map_fwd_chain(c, W, X, ldx, S, rows, ch);
    return run_chain(c, ch, s);
}

// backward of BackwardMap from dB (gradient wrt the projected embedding)
// gradient panels of one BackwardMap backward (default: the workspace's B-row set; SFAgent's 2B-row feature pass brings its own)
// (in_dim > 0: a net of the same structure on another input width and WITHOUT the projection -- svd_p's mu_net: the incoming
// gradient is wrt y itself and X is a zero-padded workspace panel)
void backward_map_bwd_chain(fbhip_ctx* c, const BwdP& W, const BwdP& G, const float* X, int ldx, BSet& S, int rows,
                            Chain& out, bool dy_done = false, const BGrad* bufs = nullptr, int in_dim = -1) {
    const fbhip_dims& d = c->d;
    const int g = in_dim > 0 ? in_dim : d.goal_dim, Hb = d.backward_hidden_dim, Lb = pad64(Hb), z = d.z_dim, Lz = pad4(z);
    // weight gradient of the first layer: X must be a zero-padded panel to use the padded width
    const bool padded_x = (X == c->W().next_goal.p) || (X == c->W().bin.p) || in_dim > 0;     // zero-padded panels only
    const int Ng = padded_x ? pad32(g) : g;
    Ws* w = &c->W();
    BSet* Sp = &S;
    const BGrad B_ = bufs ? *bufs : BGrad{w->dBm.p, w->dy.p, w->b_dr2.p, w->b_dt1.p};
    const bool projected = d.norm_z && in_dim <= 0;
    const float* dy = projected ? B_.dy : B_.dBm;       // no projection: the gradient wrt y is dB itself
    out.push_back([=](Ops& o) {                 // dy = d/dy of sqrt(d) normalize(y)   (F.normalize backward)
        if (!projected || dy_done) return;       // (the stage stays, empty: the chain's thin last round must meet forward_net's)
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
                                     G.be1, B_.ln_partials ? B_.ln_partials : w->ln_partials_b, rows, Hb, 0, 0, 0, 0, 0, pad4(Hb)});
    });
    out.push_back([=](Ops& o) {
        o.gemms.push_back(P(dy, Lz, 0, Sp->r2.p, Lb, 0, G.W3, Lb, z, Lb, rows, nullptr, EPI_NONE, nullptr, 0, G.b3));
        o.gemms.push_back(P(B_.dt1, Lb, 0, X, ldx, 0, G.W1, pad32(g), Lb, Ng, rows, nullptr, EPI_NONE, nullptr, 0, G.b1));
    });
}

// Actor.forward up to the pre-tanh policy output (fb_modules.py:107-121); Xo supplies obs (first o cols)
void actor_fwd_chain(fbhip_ctx* c, const ActP& W, const float* Xo, int ldo, const float* Xz, int ldz, ASet& S, int rows,
                     Chain& out, bool with_head) {
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
                              hp.future_ratio, d.norm_z ? nullptr : w.so.z_uniform, d.backward_identity ? -1 : (randw ? 1 : 2),
                              zx, s));
        POST_END
    }

    // the actor's own forward pass of update_actor (fb_ddpg.py:395-397) reads only the actor weights and (obs, z), not
    // forward_net or the new FB weights: in a call that also runs the FB backward it shares that backward's launches
    // (FB_BWD is two bits, see below: pass ACTOR_FWD with both or with neither)
    // ... and when the call also runs the target chain, whose first half is the SAME actor on next_obs, it rides there instead:
    // layer by layer the two passes are two problems of the same launches (own activation sets: w.as / w.asT)
    const bool actor_with_target = (mask & FBHIP_PHASE_FB_FWD_TARGET) && (mask & FBHIP_PHASE_ACTOR_FWD);
    // The optimiser counters are advanced by the first kernel of the call that precedes the optimiser pass anyway:
    // pairwise_reduce_kernel for fb_opt (3 = both optimisers when the call also holds the actor step), actor_q_kernel for a
    // lone actor step; calls that hold only the STEP phase launch step_advance_kernel.
    const int fb_adv_which = (mask & FBHIP_PHASE_ACTOR_STEP) ? 3 : 0;
    const bool fb_adv = (mask & FBHIP_PHASE_FB_BWD_A) && (mask & FBHIP_PHASE_FB_STEP);
    const bool actor_adv = (mask & FBHIP_PHASE_ACTOR_GRAD) && (mask & FBHIP_PHASE_ACTOR_STEP) &&
                           !(mask & FBHIP_PHASE_FB_STEP);
    // d/dy of B = sqrt(d) y/|y| is a row operation on the loss kernel's own output dB: pairwise_reduce_kernel does it when the
    // call continues with the backward (the BackwardMap chain then skips its l2norm_bwd launch)
    const bool fused_dy = d.norm_z && !d.backward_identity && (mask & FBHIP_PHASE_FB_BWD_A) && pad4(z) == w.bsO.y.ld && z <= 128;
    const bool early_actor = !actor_with_target && (mask & FBHIP_PHASE_FB_BWD) && (mask & FBHIP_PHASE_ACTOR_FWD);
    // policy head + sample: one row kernel when the head's width has an instantiation and its weight fits 48 KB of LDS,
    // else head GEMM (in the chain) + sample
    const bool fused_policy = policy_head_ok(H, head_width(d));
    if (fused_policy) {
        const float stddev = hp.stddev, clip = hp.stddev_clip;
        c->run_policy_heads = [=](const PolicyHeadJobs& jobs, hipStream_t q) -> int {
            HIPCK(c, launch_policy_head(jobs, c->A_p.W4, H, c->A_p.b4, Lh, a, stddev, clip, La, B, H, a, head_width(d), c->sq, q));
            return (int)FBHIP_OK;
        };
    }
    // fs / fw: the ForwardMap whose obs_action trunk consumes this action, when the head kernel is to finish that trunk's first layer
    // (the chain was built with oa_base_k = aoff); keep_pre: the pass has a backward (pre-activation and LayerNorm statistics kept)
    auto policy_stage = [=, &w](const float* noise, float* mu, float* action_dst, int ld_dst, ASet* set = nullptr,
                                FSet* fs = nullptr, const FwdP* fw = nullptr, bool keep_pre = false) {
        ASet* S = set ? set : &w.as;
        return [=](Ops& o2) {
            if (fused_policy) {                  // all heads of a round go out as one launch (flush_round)
                PolicyHeadJob jb{d.boltzmann ? S->h.p : S->p.p, H, S->premu.p, noise, mu, action_dst, ld_dst};
                if (fs != nullptr) {
                    jb.base = fs->pre1a.p; jb.ldb = H; jb.W1a = fw->oa.W1 + aoff; jb.ldw1 = fw->oa.ld1;
                    jb.gamma = fw->oa.g1; jb.beta = fw->oa.be1; jb.t1 = fs->t1a.p; jb.ldt1 = H;
                    jb.stats = keep_pre ? fs->statsA : nullptr;
                }
                o2.ph.push_back(jb);
                return;
            }
            o2.post.push_back([=](hipStream_t q) -> int {
                HIPCK(c, launch_policy_sample(S->premu.p, Lh, noise, a, hp.stddev, hp.stddev_clip, mu, La, action_dst,
                                              ld_dst, B, a, c->sq, q));
                return (int)FBHIP_OK;
            });
        };
    };
    // the policy-head kernel finishes the first layer of the trunk that consumes its action (see PolicyHeadJob)
    const bool fuse_first = fused_policy && !d.discrete && policy_first_ok(H, a, head_width(d));

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
                if (fuse_first) {
                    // what forward_target computes without the action -- the action-free part of the obs_action trunk's first
                    // layer and the whole obs_z trunk -- is part of the ONLINE phase (below): like the online passes it depends on
                    // the previous update only through its FB optimiser step.  Here: the head kernel finishes the obs_action
                    // trunk's first layer, and the chain continues with that trunk's second layer.
                    ch[0].push_back(policy_stage(w.so.eps_next, nullptr, w.Xnoa.p + aoff, w.Xnoa.ld, &w.asT, &w.fsT, &c->F_t));
                    forward_map_fwd_chain(c, c->F_t, w.Xnoa.p, w.Xnoa.ld, w.Xnoz.p, w.Xnoz.ld, w.fsT, B, ch[0], true, 0, nullptr, 0, aoff, 2);
                } else {
                    ch[0].push_back(policy_stage(w.so.eps_next, nullptr, w.Xnoa.p + aoff, w.Xnoa.ld, &w.asT));
                    forward_map_fwd_chain(c, c->F_t, w.Xnoa.p, w.Xnoa.ld, w.Xnoz.p, w.Xnoz.ld, w.fsT, B, ch[0]);
                }
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
            if ((mask & FBHIP_PHASE_FB_FWD_ONLINE) && fuse_first) {      // forward_target without the action (see above)
                ch.emplace_back();
                forward_map_fwd_chain(c, c->F_t, w.Xnoa.p, w.Xnoa.ld, w.Xnoz.p, w.Xnoz.ld, w.fsT, B, ch.back(), true, 0, nullptr, 0, aoff, 1);
            }
            if ((mask & FBHIP_PHASE_FB_FWD_ONLINE) && !(mask & FBHIP_PHASE_SAMPLE)) {
                ch.emplace_back();
                backward_map_fwd_chain(c, c->K_t, next_goal, ld_ng, w.bsA, B, ch.back());
                ch.emplace_back();
                backward_map_fwd_chain(c, c->K_p, next_goal, ld_ng, w.bsO, B, ch.back());
            }
            // both ForwardMap chains in one call: the online heads' thin output layer (+ its split-K reduce) waits for the round
            // of the target chain's, five rounds later -- one launch pair instead of two, nothing needs it earlier
            if (ch[0].size() > ch[1].size() && !ch[1].empty()) {
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
            HIPCK(c, launch_extra_metrics(w.fsO.F1.p, BmO, w.z.p, Lz, B, z, w.cov.p, w.cov.ld, w.metrics, s, Bg, c->d_xm_part,
                                          c->d_pubseq ? c->d_pubseq + 1 : nullptr));
            if (d.discrete && c->h_metrics) HIPCK(c, launch_metrics_publish(w.metrics, c->h_metrics, c->d_pubseq, s));   // (no actor phase: these are the last)
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
            if (!d.backward_identity)             // (IdentityMap: dB ends at the input, nothing behind it)
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
        // In the pipelined multi-step graph V is produced on the OTHER capture branch (enqueue_actor_v, ahead of the next step's
        // head): it depends on the FB optimiser step only, and as part of this chain's first round its two K = z_dim problems
        // cost ~20 us of the actor phase's critical path.  c->v_ready then carries the event this chain waits for before actor_q.
        hipEvent_t v_ready = c->v_ready;
        if (v_ready == nullptr) {
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
                if (v_ready != nullptr) HIPCK(c, hipStreamWaitEvent(q, v_ready, 0));
                HIPCK(c, launch_actor_q(w.fsO.p.p, 2 * H, w.dp.p, 2 * H, w.z.p, Lz, c->F_p.b4[0], c->F_p.b4[1], w.as.mu.p, La,
                                        w.Xopi.p + aoff, w.Xopi.ld, hp.stddev, hp.want_metrics ? w.metrics : nullptr,
                                        w.pw_scratch, B, H, z, a, c->sq, w.as.premu.p, Lh, w.so.eps_actor, a, q,
                                        actor_adv ? w.st : nullptr, 1,
                                        // every metric of this update is final with this launch: its finalize kernel hands them to
                                        // the host, BEFORE the actor's backward pass and optimiser step
                                        hp.want_metrics ? c->h_metrics : nullptr, hp.want_metrics && c->h_metrics ? c->d_pubseq : nullptr));
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
        const bool fused_head = !d.boltzmann && actor_head_bwd_ok(H, a);
        const bool fused_ln = fused_head && H <= 2048;      // the LayerNorm+tanh backward of this row chain joins the kernel
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
// head: sampling + the passes that depend on the previous step only through its optimiser steps (online successor_net,
// feature_net [, mu_net]) -- what the pipelined multi-step graph runs on its second branch beside the previous actor phase;
// mid: target chain, the actor's own pass, losses, backward passes; step: sf_opt / phi_opt steps + the target EMA (a data-parallel
// host all-reduces the gradient bucket in between).  All three: one complete update_sf.
int build_update_sf(fbhip_ctx* c, const fbhip_hparams& hp, const fbhip_inject* inj, Program& prog, bool head, bool mid, bool step) {
    const fbhip_dims& d = c->d;
    Ws& w = c->W();
    const int B = d.batch, o = d.obs_dim, a = d.action_dim, g = d.goal_dim, z = d.z_dim, H = d.hidden_dim,
              Hb = d.backward_hidden_dim, Lb = pad64(Hb), Lz = pad4(z), La = pad4(a);
    const Geom gm = geom_of(d);
    const int aoff = gm.single ? o + z : o;
    if (hp.future_ratio != 0.f || hp.rand_weight) {
        c->err = g_err = "fbhip: SFAgent has no future_ratio / rand_weight (sf.py:57-96)";
        return FBHIP_E_INVALID;
    }
    // mix_ratio > 0 (sf.py:725-739): z[mix] = sqrt(d) normalize(phi(next_goal[perm]) @ pinv(phi^T phi / B)).  The covariance is
    // inverted by inverse_kernel (Gauss-Jordan in fp64): equal to pinv while phi^T phi has full rank, i.e. B >= z_dim and no
    // feature column that is constant zero or a combination of others -- the only regime in which the whitening means anything.
    const bool mixz = hp.mix_ratio > 0.f;
    // ---- sample: the FB sampler with the identity permutation: goal2 = [goal ; next_goal] (sf.py:705-721), z = sample_z (:723)
    // contrastive reads batch.future_goal (sf.py:125, 713-719): the hindsight draw of in_memory_replay_buffer.py:157-161 without FB's z override
    const bool hind = d.sf == 10 || d.sf == 11;
    const int RB = (d.sf == 11 ? 3 : 2) * B;              // rows of the feature pass: [goal ; next_goal (; future_goal)]
    if (hind && !(hp.future < 1.f)) { c->err = g_err = "fbhip: the contrastive feature learner needs a replay buffer with future < 1"; return FBHIP_E_INVALID; }
    const bool all_injected = inj && inj->ep_idx && inj->step_idx && inj->z_gauss && inj->eps_next && inj->eps_actor && (!hind || inj->future_idx) &&
                              (!mixz || (inj->perm && inj->mix_uniform));
    if (head) {
    POST_BEGIN
    if (!all_injected) HIPCK(c, launch_draw(c->rv, w.so, B, z, a, c->seed, c->rank, w.st, hind ? hp.future : -1.f, 1, s));
    if (inj != nullptr) {
#define INJ(field, bytes) if (inj->field) HIPCK(c, hipMemcpyAsync(w.so.field, inj->field, (size_t)(bytes), hipMemcpyDeviceToDevice, s))
        INJ(ep_idx, B * 4); INJ(step_idx, B * 4); INJ(z_gauss, (size_t)B * z * 4); INJ(eps_next, (size_t)B * a * 4);
        INJ(eps_actor, (size_t)B * a * 4);
        if (hind) INJ(future_idx, B * 4);
        if (mixz) { INJ(perm, B * 4); INJ(mix_uniform, B * 4); }
#undef INJ
    }
    GatherArgs ga{};
    ga.rv = c->rv; ga.ep_idx = w.so.ep_idx; ga.step_idx = w.so.step_idx; ga.perm = nullptr;
    ga.Xoa = w.Xoa.p; ga.ld_oa = w.Xoa.ld; ga.Xoz = w.Xoz.p; ga.ld_oz = w.Xoz.ld; ga.Xnoz = w.Xnoz.p; ga.ld_noz = w.Xnoz.ld;
    ga.Xnoa = w.Xnoa.p; ga.ld_noa = w.Xnoa.ld; ga.Xopi = w.Xopi.p; ga.ld_opi = w.Xopi.ld;
    ga.next_goal = w.next_goal.p; ga.ld_ng = w.next_goal.ld; ga.bin = w.bin.p; ga.ld_bin = w.bin.ld; ga.disc = w.disc;
    ga.Xo = w.Xo.p; ga.ld_o = w.Xo.ld; ga.future_idx = hind ? w.so.future_idx : nullptr; ga.fgoal = w.fgoal.p; ga.ld_fg = w.fgoal.ld;
    ga.B = B; ga.o = o; ga.a = a; ga.g = g; ga.use_goal = d.use_goal; ga.gamma = hp.discount; ga.aoff = aoff; ga.act_idx = nullptr;
    if (mixz) { ga.pgoal = w.pgoal.p; ga.ld_pg = w.pgoal.ld; ga.pperm = w.so.perm; }
    HIPCK(c, launch_gather(ga, s));
    POST_END
    const float* phi_p = w.pgoal.p;            // identity: phi = the raw goal (sf.py:74-81; g == z)
    int ld_phi = w.pgoal.ld;
    if (mixz && d.sf != 12) {                  // feature_net on the permuted next goals, with the weights as they are now (no_grad)
        std::vector<Chain> ch(1);
        backward_map_fwd_chain(c, c->K_p, w.pgoal.p, w.pgoal.ld, w.bsF, B, ch[0]);
        prog_parallel(prog, ch);
        phi_p = w.bsF.Bm.p; ld_phi = Lz;
    }
    POST_BEGIN
    if (mixz) {
        RC(run_gemms(c, {P(phi_p, ld_phi, 0, phi_p, ld_phi, 0, w.cov.p, w.cov.ld, z, z, B)}, s));                 // phi^T phi
        HIPCK(c, launch_inverse(w.cov.p, w.cov.ld, z, 1.0f / (float)B, w.inv_cov.p, w.inv_cov.ld, s));
        RC(run_gemms(c, {P(phi_p, ld_phi, 1, w.inv_cov.p, w.inv_cov.ld, 0, w.ymixw.p, Lz, B, z, z)}, s));         // phi @ inv_cov
    }
    ZPanels zx{};
    if (gm.single) zx = ZPanels{{w.Xoa.p, w.Xnoa.p, w.Xopi.p}, {w.Xoa.ld, w.Xnoa.ld, w.Xopi.ld}};
    // (one projection for the mixed rows: sf.py:738)
    HIPCK(c, launch_mix_z(w.so.z_gauss, z, mixz ? w.ymixw.p : nullptr, Lz, w.so.mix_uniform, hp.mix_ratio, w.z.p, Lz, w.Xoz.p, w.Xoz.ld,
                          w.Xnoz.p, w.Xnoz.ld, o, B, z, w.st, nullptr, nullptr, 0.f, nullptr, 1, zx, s));
    POST_END
    }

    // ---- forward passes: target chain (actor(next_obs) -> next_action -> successor_target), online successor_net, the
    // feature pass on [goal ; next_goal], and update_actor's own actor pass (it reads only the actor weights)
    const bool fused_policy = policy_head_ok(H, a);
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
        std::vector<Chain> ch(7);                // (empty chains contribute nothing; the order keeps the one-call launches as they were)
        if (mid) {
            actor_fwd_chain(c, c->A_p, w.Xnoz.p, w.Xnoz.ld, w.Xnoz.p, w.Xnoz.ld, w.asT, B, ch[0], !fused_policy);
            ch[0].push_back(policy_stage(w.so.eps_next, nullptr, w.Xnoa.p + aoff, w.Xnoa.ld, &w.asT));
            forward_map_fwd_chain(c, c->F_t, w.Xnoa.p, w.Xnoa.ld, w.Xnoz.p, w.Xnoz.ld, w.fsT, B, ch[0]);
        }
        if (head) {
            forward_map_fwd_chain(c, c->F_p, w.Xoa.p, w.Xoa.ld, w.Xoz.p, w.Xoz.ld, w.fsO, B, ch[1]);
            if (d.sf != 12) backward_map_fwd_chain(c, c->K_p, w.goal2.p, w.goal2.ld, w.bsS, RB, ch[2]);
            else            // identity (sf.py:94-98): feature_net = nn.Identity(): phi(goal) is the goal itself (z_dim == goal_dim), a copy into the phi panel
                ch[2].push_back([=, &w](Ops& o2) {
                    o2.post.push_back([=, &w](hipStream_t q) -> int {
                        HIPCK(c, launch_concat2(w.bsS.Bm.p, Lz, w.goal2.p, w.goal2.ld, z, nullptr, 0, 0, RB, q));
                        return (int)FBHIP_OK;
                    });
                });
        }
        if (mid) {
            actor_fwd_chain(c, c->A_p, w.Xo.p, w.Xo.ld, w.Xoz.p, w.Xoz.ld, w.as, B, ch[3], !fused_policy);
            ch[3].push_back(policy_stage(w.so.eps_actor, w.as.mu.p, w.Xopi.p + aoff, w.Xopi.ld, &w.as));
        }
        if (d.sf == 7 && head)       // latent: next_phi of its loss = target_feature_net(next_goal) (sf.py:241-242), the TARGET copy of the feature block
            backward_map_fwd_chain(c, c->K_t, w.next_goal.p, w.next_goal.ld, w.bsA, B, ch[4]);
        if (d.sf == 8 && head) {
            // svd_sr (sf.py:264-275): mu = mu_net(next_goal) and the two target nets on next_goal -- the TARGET copies of the feature
            // block and of mu_net (own initial weights, followed at 0.01 like latent's)
            backward_map_fwd_chain(c, c->M_p, w.next_goal.p, w.next_goal.ld, w.bsM, B, ch[4], false, d.goal_dim);
            backward_map_fwd_chain(c, c->K_t, w.next_goal.p, w.next_goal.ld, w.bsA, B, ch[5]);
            backward_map_fwd_chain(c, c->M_t, w.next_goal.p, w.next_goal.ld, w.bsO, B, ch[6], false, d.goal_dim);
        }
        if (d.sf == 10 && head)      // contrastive (sf.py:136): future_mu = mu_net(future_goal), the BackwardMap chain WITH its projection
            backward_map_fwd_chain(c, c->M_p, w.fgoal.p, w.fgoal.ld, w.bsM, B, ch[4]);
        if (d.sf == 11 && head)      // contrastivev2 (sf.py:175-176): mu = mu_net(goal); future_phi is the third block of the feature pass
            backward_map_fwd_chain(c, c->M_p, w.bin.p, w.bin.ld, w.bsM, B, ch[4]);
        if (d.sf == 9 && head) {
            // svd_srv2 (sf.py:303-318): mu = mu_net(goal); the two target nets on next_goal as for svd_sr
            backward_map_fwd_chain(c, c->M_p, w.bin.p, w.bin.ld, w.bsM, B, ch[4], false, d.goal_dim);
            backward_map_fwd_chain(c, c->K_t, w.next_goal.p, w.next_goal.ld, w.bsA, B, ch[5]);
            backward_map_fwd_chain(c, c->M_t, w.next_goal.p, w.next_goal.ld, w.bsO, B, ch[6], false, d.goal_dim);
        }
        if (d.sf == 6 && head) {      // svd_p: mu = mu_net(cat[goal, action]) (sf.py:347), the BackwardMap module chain on another input, unprojected
            const int g = d.goal_dim, act = d.action_dim;
            ch[4].push_back([=, &w](Ops& o2) {       // (a stage of its own: the panel is built behind the first round, mu_net starts in the second)
                o2.post.push_back([=, &w](hipStream_t q) -> int {
                    HIPCK(c, launch_concat2(w.Xga.p, w.Xga.ld, w.goal2.p, w.goal2.ld, g, w.Xoa.p + aoff, w.Xoa.ld, act, B, q));
                    return (int)FBHIP_OK;
                });
            });
            backward_map_fwd_chain(c, c->M_p, w.Xga.p, w.Xga.ld, w.bsM, B, ch[4], false, g + act);
        }
        prog_parallel(prog, ch);
    }
    if (mid) {
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
    int hin = 0, hout = 0;
    const char* hpfx = nullptr;
    if (sf_head_dims(d, &hin, &hout, &hpfx)) {
        // the feature learner's head mlp on [phi | ...] regressed on a target (icm: tanh head vs the action, sf.py:203-213;
        // autoencoder: vs the goal itself, :249-262; transition: [phi | action] vs the next goal, :215-227)
        const IcmP &I = c->I_p, &G = c->I_g;
        const int Kc = pad32(hin), a = hout, La = pad4(hout), sfm = d.sf, act = d.action_dim;
        const bool with_action = sfm == 5 || sfm == 7;
        const float* second = sfm == 1 ? nphi : with_action ? w.Xoa.p + aoff : nullptr;
        const int ld2 = sfm == 1 ? Lz : w.Xoa.ld, n2 = sfm == 1 ? z : with_action ? act : 0;
        const float* target = sfm == 1 ? w.Xoa.p + aoff : sfm == 4 ? w.goal2.p : sfm == 7 ? w.bsA.Bm.p : w.goal2.p + (size_t)B * w.goal2.ld;
        const int ldt = sfm == 1 ? w.Xoa.ld : sfm == 7 ? Lz : w.goal2.ld;
        feat.push_back([=, &w](Ops& o2) {
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, launch_concat2(w.icat.p, Kc, phi, Lz, z, second, ld2, n2, B, q));
                // only icm's head sees next_phi: the feature pass runs on [goal ; next_goal], so its half of the gradient panel is zero
                if (sfm != 1) HIPCK(c, hipMemsetAsync(dnphi, 0, (size_t)B * Lz * sizeof(float), q));
                return (int)FBHIP_OK;
            });
        });
        feat.push_back([=, &w](Ops& o2) { o2.gemms.push_back(P(w.icat.p, Kc, 1, I.W1, Kc, 1, w.ih1.p, Lb, B, Lb, Kc, I.b1, EPI_BIAS_RELU)); });
        feat.push_back([=, &w](Ops& o2) { o2.gemms.push_back(P(w.ih1.p, Lb, 1, I.W2, Lb, 1, w.ih2.p, Lb, B, Lb, Lb, I.b2, EPI_BIAS_RELU)); });
        feat.push_back([=, &w](Ops& o2) {
            o2.gemms.push_back(P(w.ih2.p, Lb, 1, I.W3, Lb, 1, w.ipre.p, La, B, a, Lb, I.b3, EPI_BIAS));
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, launch_icm_loss(w.ipre.p, La, target, ldt, w.d_ipre.p, La, B, a, sfm == 1 ? 1 : 0, w.metrics, w.pw_scratch, q));
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
            if (sfm == 1) o2.gemms.push_back(P(w.d_ih1.p, Lb, 1, I.W1 + z, Kc, 0, dnphi, Lz, B, z, Lb));       // d cat[:, z:2z]
        });
    } else if (d.sf == 6) {
        // svd_p (sf.py:344-362): P = mu . phi'^T,  loss = -2 mean diag P + mean offdiag P^2 + orthonormality(phi')  with phi' = phi(next_goal)
        //   = 2 x [ the FB loss of (F1 = mu, F2 = 0, B = phi', no target) with ortho_coef = 1/2 ]: the pairwise kernel as it is, every
        // gradient doubled; dF1 is d mu, dB is d phi' (rows [B, 2B) of the feature pass; the goal rows get no gradient)
        feat.push_back([=, &w](Ops& o2) {
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, hipMemsetAsync(dphi, 0, (size_t)B * Lz * sizeof(float), q));
                HIPCK(c, launch_pairwise_fb(w.bsM.y.p, w.zeroF.p, nphi, w.zeroF.p, w.zeroF.p, nphi, w.disc, B, z, Lz, 0.5f, w.dmu.p,
                                            w.lapS2.p, dnphi, w.metrics, w.pw_scratch, q, nullptr, 0, nullptr, nullptr, nullptr, 2.0f));
                HIPCK(c, launch_scale_metric(w.metrics, FBHIP_M_FB_LOSS, FBHIP_M_PHI_LOSS, 2.0f, q));
                return (int)FBHIP_OK;
            });
        });
    } else if (d.sf == 8) {
        // svd_sr (sf.py:271-292): SR = phi . mu^T, target_SR = target_phi . target_mu^T (both on next_goal),
        //   loss = -2 mean diag SR + mean offdiag (SR - 0.99 target_SR)^2 + orthonormality(phi)
        // The first two terms are the FB loss of two IDENTICAL heads F1 = F2 = phi against B = mu with targets tF1 = tF2 = target_phi,
        // tB = target_mu and a constant discount 0.99 (min of two equal target products; 1/2 + 1/2 of the squares; diag + diag); the
        // third is the pairwise kernel's covariance role on phi alone, as for lap.  d phi = dF1 + dF2 + the orthonormality share.
        feat.push_back([=, &w](Ops& o2) {
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, launch_fill_add(w.c99, nullptr, 0.99f, B, q));
                HIPCK(c, hipMemsetAsync(dnphi, 0, (size_t)B * Lz * sizeof(float), q));
                HIPCK(c, launch_pairwise_fb(phi, phi, w.bsM.y.p, w.bsA.Bm.p, w.bsA.Bm.p, w.bsO.y.p, w.c99, B, z, Lz, 0.0f, dphi, w.lapS2.p, w.dmu.p,
                                            w.metrics, w.pw_scratch, q));
                HIPCK(c, launch_scale_metric(w.metrics, FBHIP_M_FB_LOSS, FBHIP_M_PHI_LOSS, 1.0f, q));
                HIPCK(c, launch_fill_add(dphi, w.lapS2.p, 0.f, (int64_t)B * Lz, q));
                HIPCK(c, launch_pairwise_fb(w.zeroF.p, w.zeroF.p, phi, w.zeroF.p, w.zeroF.p, phi, w.disc, B, z, Lz, 1.0f, w.lapS1.p, w.lapS2.p,
                                            w.dphi_o.p, w.metrics, w.pw_scratch, q));
                HIPCK(c, launch_scale_metric(w.metrics, FBHIP_M_FB_LOSS, FBHIP_M_PHI_LOSS, 1.0f, q, 1));
                HIPCK(c, launch_fill_add(dphi, w.dphi_o.p, 0.f, (int64_t)B * Lz, q));
                return (int)FBHIP_OK;
            });
        });
    } else if (d.sf == 11) {
        // contrastivev2 (sf.py:173-186): the same loss with the roles swapped -- logits[s, t] = cos(mu_net(goal)_s, feature_net(future_goal)_t)
        const float* mu = w.bsM.Bm.p;
        const float* fphi = w.bsS.Bm.p + (size_t)2 * B * Lz;
        float* dfphi = w.dBm2.p + (size_t)2 * B * Lz;
        feat.push_back([=, &w](Ops& o2) { o2.gemms.push_back(P(mu, Lz, 1, fphi, Lz, 1, w.rw, B, B, B, z)); });
        feat.push_back([=, &w](Ops& o2) {
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, launch_contrastive_rows(w.rw, B, B, z, w.metrics, w.pw_scratch, q));
                HIPCK(c, hipMemsetAsync(dphi, 0, (size_t)2 * B * Lz * sizeof(float), q));      // goal and next_goal rows: no gradient
                return (int)FBHIP_OK;
            });
        });
        feat.push_back([=, &w](Ops& o2) {
            o2.gemms.push_back(P(w.rw, B, 1, fphi, Lz, 0, w.dmu.p, Lz, B, z, B));
            o2.gemms.push_back(P(w.rw, B, 0, mu, Lz, 0, dfphi, Lz, B, z, B));
        });
    } else if (d.sf == 10) {
        // contrastive (sf.py:134-142): logits = normalize(phi) . normalize(future_mu)^T = phi . mu^T / d (both already have norm sqrt(d)),
        // loss = mean_s(-logits_ss + logsumexp_{t != s} logits_st).  Three small GEMMs around one row kernel on the [B, B] scratch
        // matrix (the rand_weight panel, unused here): L = phi . mu^T;  L <- dloss/dL;  d phi = L . mu,  d mu = L^T . phi.  The
        // projections' own backward passes (feature_net's and mu_net's L2 stages) remove the radial parts, as F.normalize does twice.
        const float* mu = w.bsM.Bm.p;
        feat.push_back([=, &w](Ops& o2) { o2.gemms.push_back(P(phi, Lz, 1, mu, Lz, 1, w.rw, B, B, B, z)); });
        feat.push_back([=, &w](Ops& o2) {
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, launch_contrastive_rows(w.rw, B, B, z, w.metrics, w.pw_scratch, q));
                HIPCK(c, hipMemsetAsync(dnphi, 0, (size_t)B * Lz * sizeof(float), q));
                return (int)FBHIP_OK;
            });
        });
        feat.push_back([=, &w](Ops& o2) {
            o2.gemms.push_back(P(w.rw, B, 1, mu, Lz, 0, dphi, Lz, B, z, B));
            o2.gemms.push_back(P(w.rw, B, 0, phi, Lz, 0, w.dmu.p, Lz, B, z, B));
        });
    } else if (d.sf == 9) {
        // svd_srv2 (sf.py:311-329): SR = mu(goal) . phi(next_goal)^T against 0.98 x target_mu . target_phi^T (both on next_goal), plus
        // the orthonormality of phi(next_goal): the FB loss itself -- two identical heads F = mu, B = phi', its own orthonormality
        // term on B with ortho_coef = 1 -- in ONE launch;  d mu = dF1 + dF2, d phi' = dB, the goal rows of the feature pass get none
        feat.push_back([=, &w](Ops& o2) {
            o2.post.push_back([=, &w](hipStream_t q) -> int {
                HIPCK(c, launch_fill_add(w.c99, nullptr, 0.98f, B, q));
                HIPCK(c, hipMemsetAsync(dphi, 0, (size_t)B * Lz * sizeof(float), q));
                HIPCK(c, launch_pairwise_fb(w.bsM.y.p, w.bsM.y.p, nphi, w.bsO.y.p, w.bsO.y.p, w.bsA.Bm.p, w.c99, B, z, Lz, 1.0f, w.dmu.p, w.lapS2.p,
                                            dnphi, w.metrics, w.pw_scratch, q));
                HIPCK(c, launch_scale_metric(w.metrics, FBHIP_M_FB_LOSS, FBHIP_M_PHI_LOSS, 1.0f, q));
                HIPCK(c, launch_fill_add(w.dmu.p, w.lapS2.p, 0.f, (int64_t)B * Lz, q));
                return (int)FBHIP_OK;
            });
        });
    } else if (d.sf == 2) {
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
    // (random, sf.py:430: feature_net keeps its initial weights -- no loss, no phi_opt; its gradient block stays zero and the
    // Adam pass below leaves it where it is)
    BGrad bg{w.dBm2.p, w.dy2.p, w.s_dr2.p, w.s_dt1.p};
    const bool frozen = d.sf == 3 || d.sf == 12;          // random / FB, identity: nothing to train behind phi
    if (!frozen) backward_map_bwd_chain(c, c->K_p, c->K_g, w.goal2.p, w.goal2.ld, w.bsS, RB, feat, false, &bg);
    Chain succ;
    forward_map_bwd_chain(c, c->F_p, c->F_g, w.Xoa.p, w.Xoa.ld, w.Xoz.p, w.Xoz.ld, w.fsO, B, succ);
    {
        std::vector<Chain> ch{succ};
        if (!frozen) ch.push_back(feat);
        if (d.sf == 6 || d.sf >= 8) {      // mu_net's backward from d mu; its first stage is empty (no projection), which keeps it one round behind the loss
            Chain mu;
            mu.push_back([](Ops&) {});
            BGrad mg{w.dmu.p, w.dmu.p, w.m_dr2.p, w.m_dt1.p, w.ln_partials_m};
            if (d.sf == 6) backward_map_bwd_chain(c, c->M_p, c->M_g, w.Xga.p, w.Xga.ld, w.bsM, B, mu, true, &mg, d.goal_dim + d.action_dim);
            else if (d.sf == 8) backward_map_bwd_chain(c, c->M_p, c->M_g, w.next_goal.p, w.next_goal.ld, w.bsM, B, mu, true, &mg, d.goal_dim);
            else if (d.sf == 10 || d.sf == 11) {        // projected: d mu -> d y first; two more empty stages: its loss has three
                mu.push_back([](Ops&) {});
                mu.push_back([](Ops&) {});
                BGrad cg{w.dmu.p, w.dmu_y.p, w.m_dr2.p, w.m_dt1.p, w.ln_partials_m};
                const Buf& xin = d.sf == 10 ? w.fgoal : w.bin;
                backward_map_bwd_chain(c, c->M_p, c->M_g, xin.p, xin.ld, w.bsM, B, mu, false, &cg);
            }
            else backward_map_bwd_chain(c, c->M_p, c->M_g, w.bin.p, w.bin.ld, w.bsM, B, mu, true, &mg, d.goal_dim);
            ch.push_back(mu);
        }
        prog_parallel(prog, ch);
    }
    }
    if (!step) return FBHIP_OK;
    // ---- sf_opt.step() + phi_opt.step() (sf.py:643-653): one pass over forward ++ backward, lr | lr_coef * lr; the EMA of
    // successor_target_net (sf.py:751-752) rides along like FBDDPGAgent's (nothing reads a target before the next update)
    POST_BEGIN
    HIPCK(c, launch_step_advance(w.st, 0, s));
    const int64_t nf = c->L[FBHIP_NET_FORWARD].numel, nb = c->L[FBHIP_NET_BACKWARD].numel;
    // (latent, sf.py:245: utils.soft_update_params(feature_net, target_feature_net, 0.01) runs inside the learner's forward(), i.e.
    // towards the feature parameters as they were BEFORE phi_opt.step(); the other learners never read that part of the target buffer)
    HIPCK(c, launch_adam_ema(c->fb_p, c->fb_g, c->fb_m, c->fb_v, c->fb_t, nf + nb, hp.lr, hp.lr_coef * hp.lr, nf, hp.grad_scale,
                             hp.fb_target_tau, w.st, 0, 0, s, (d.sf >= 7 && d.sf <= 9) ? 0.01f : -1.f, (d.sf >= 7 && d.sf <= 9) ? 1 : 0));
    POST_END
    return FBHIP_OK;
}
#undef POST_BEGIN
#undef POST_END

// V_i = z . W4_i of the CURRENT workspace set (update_actor's Q and head gradient read it, see build_update) as a launch of its
// own: the pipelined multi-step graph issues it on the second capture branch right after the FB optimiser step
int enqueue_actor_v(fbhip_ctx* c, hipStream_t s) {
    const fbhip_dims& d = c->d;
    Ws& w = c->W();
    const int B = d.batch, z = d.z_dim, H = d.hidden_dim, Lz = pad4(z);
    return run_gemms(c, {P(w.z.p, Lz, 1, c->F_p.W4[0], H, 0, w.dp.p, 2 * H, B, H, z),
                         P(w.z.p, Lz, 1, c->F_p.W4[1], H, 0, w.dp.p + H, 2 * H, B, H, z)}, s);
}

int enqueue_update(fbhip_ctx* c, const fbhip_hparams& hp, const fbhip_inject* inj, int mask, hipStream_t s) {
    Program prog;
    if (c->d.sf) {
        // a complete update, or one of the three groups the pipelined multi-step graph cuts it into
        // a complete update, the three groups the pipelined multi-step graph cuts it into, or the data-parallel cuts: the
        // gradient passes and the two optimiser steps are separable (head | gradients | sf/phi step | actor gradient | actor step,
        // in this order; each group whole)
        const int HEAD = FBHIP_PHASE_SAMPLE | FBHIP_PHASE_FB_FWD_ONLINE;
        const int GRAD = FBHIP_PHASE_FB_FWD_TARGET | FBHIP_PHASE_FB_BWD | FBHIP_PHASE_ACTOR_FWD;
        const bool head = (mask & HEAD) == HEAD, grad = (mask & GRAD) == GRAD, step = (mask & FBHIP_PHASE_FB_STEP) != 0;
        const int tail = mask & (FBHIP_PHASE_ACTOR_GRAD | FBHIP_PHASE_ACTOR_STEP);
        if ((mask & HEAD) != (head ? HEAD : 0) || (mask & GRAD) != (grad ? GRAD : 0) || mask == 0 || (mask & ~FBHIP_PHASE_ALL)) {
            c->err = g_err = "fbhip: dims.sf runs whole phase groups only: SAMPLE|FB_FWD_ONLINE, FB_FWD_TARGET|FB_BWD|ACTOR_FWD, FB_STEP, ACTOR_GRAD, ACTOR_STEP";
            return FBHIP_E_INVALID;
        }
        if (head || grad || step) RC(build_update_sf(c, hp, inj, prog, head, grad, step));
        if (tail) RC(build_update(c, hp, nullptr, tail, prog));                                         // sf.py:666-694
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


}  // namespace host
}  // namespace fbhip
