// Internal declarations shared by the gfx950 kernels and the C-ABI translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace fbhip {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// epilogue kinds of the grouped GEMM (mirrors include/fbhip.h::fbhip_gemm)
enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_MASK_RELU = 3, EPI_TANH_BWD = 4 };

// One GEMM of a grouped launch: C[M,N] = epi(sum_k A(m,k) * B(n,k)).
struct GemmProblem {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* aux;
    float* colsum;
    int M, N, K;
    int lda, ldb, ldc, ldaux;
    int a_kcontig, b_kcontig;
    int a_vec, b_vec;        // 16-byte aligned base + ld % 4 == 0 -> float4 global loads
    int epi;
    int tiles_m, tiles_n, tile_start;
    int xgm, xgn;            // > 0: the 8 XCDs own an xgm x xgn grid of BLOCKS of this problem's tiles (gemm_problem_finalize); 0: bands of rows
    // grid-level split-K (small outputs): slice s handles K chunks [s*kper, (s+1)*kper) and writes raw partial
    // tiles to partial[s][M][N] (+ partial column sums after them); splitk_reduce folds them in slice order and
    // applies the epilogue -- deterministic, no atomics.
    int kslices, kper;       // kper in units of the config's K chunk
    float* partial;
    int red_start;           // first element of this problem in the reduce launch
};

constexpr int MAX_GROUP = 8;
struct GemmGroup {
    GemmProblem p[MAX_GROUP];
    int n;
    int total_tiles;
};

// tile configurations: <waves along M, waves along N, waves along K>, each wave owns one 32x32 MFMA tile
enum GemmCfg { CFG_2x2x1 = 0, CFG_2x1x2 = 1, CFG_1x2x2 = 2, CFG_1x1x4 = 3, CFG_4x1x1 = 4, CFG_DMA128 = 5,
               CFG_COUNT };   // the last: the LDS-DMA kernel (128x64 tiles), needs gemm_problem_dma_ok()

hipError_t launch_gemm_group(const GemmGroup& g, int cfg, hipStream_t stream);
// LayerNorm-backward column reduces (d gamma, d beta = sums of per-block partials) that ride in a split-K reduce launch instead
// of paying for a launch of their own: both are tiny fold-the-partials jobs whose results only the optimiser reads
constexpr int CR_MAX = 6;
struct ColReduceJobs {
    const float* partials[CR_MAX]; float* dgamma[CR_MAX]; float* dbeta[CR_MAX];
    int rows[CR_MAX], n[CR_MAX], block_start[CR_MAX + 1];
    int count;
};
hipError_t launch_splitk_reduce(const GemmGroup& g, int total_elems, hipStream_t stream, const ColReduceJobs* extra = nullptr);
int gemm_cfg_bkt(int cfg);    // K extent of one chunk of a tile configuration
int gemm_cfg_bm(int cfg);     // tile rows / columns
int gemm_cfg_bn(int cfg);
hipError_t gemm_init();        // one-time kernel attribute setup (outside graph capture)
int pick_gemm_cfg(int M, int N, int K);
bool gemm_problem_dma_ok(const GemmProblem& p);        // eligible for the LDS-DMA kernels (CFG_DMA128 requires it)
void gemm_problem_finalize(GemmProblem& p, int cfg);   // fills a_vec/b_vec/tiles_*

// ---- embedding heads in one launch (fused.hip) ----------------------------------------------------------------------------
// C = X[rows, K] . W[N, K]^T + bias, N <= 64 (an embedding head); out2 != nullptr: also out2 = scale * C / max(|C|_row, 1e-12)
// and norms[row] = |C|_row (sqrt(d) F.normalize, fb_modules.py:229).  K % 4 == 0, 16-byte aligned rows, ldc / ldo >= pad4(N)
// (the pad columns are written as 0), bias readable to pad4(N).
struct HeadProblem { const float* X; int ldx; const float* W; int ldw; const float* bias; float* C; int ldc;
                     float* out2; int ldo; float* norms; float scale; int rows, N, K; };
constexpr int HEAD_MAX_GROUP = 6;
struct HeadGroup { HeadProblem p[HEAD_MAX_GROUP]; int n; };
bool head_ok(const HeadProblem& p);
hipError_t launch_head_group(const HeadGroup& g, hipStream_t s);

// ---- row-wise ops ------------------------------------------------------------------------------------
hipError_t launch_ln_tanh_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy,
                              float* stats, int rows, int n, hipStream_t s);
hipError_t launch_ln_tanh_bwd(const float* dy, int lddy, const float* y, int ldy, const float* x, int ldx,
                              const float* stats, const float* gamma, float* dx, int lddx,
                              float* dgamma, float* dbeta, float* partials, int rows, int n, hipStream_t s);
constexpr int LN_BWD_ROWS_PER_BLOCK = 8;
// grouped variants: the two trunks of a net (independent rows, different parameters) share one launch
// np (optional, 0 = unknown): a multiple of 4 with n <= np <= every leading dimension and np readable elements in
// gamma / beta -- lets the kernels use branch-free clamped float4 loads (all of a row's loads in flight at once)
struct LnFwdProblem { const float* x; int ldx; const float* gamma; const float* beta; float* y; int ldy; float* stats;
                      int rows, n, vx, vy, vp, np; };
constexpr int LN_MAX_GROUP = 6;
struct LnFwdGroup { LnFwdProblem p[LN_MAX_GROUP]; int n; };
hipError_t launch_ln_tanh_fwd_group(LnFwdGroup g, hipStream_t s);
struct LnBwdProblem { const float* dy; int lddy; const float* y; int ldy; const float* x; int ldx; const float* stats;
                      const float* gamma; float* dx; int lddx; float* dgamma; float* dbeta; float* partials;
                      int rows, n, vdy, vy, vx, vdx, vp, np; };
struct LnBwdGroup { LnBwdProblem p[LN_MAX_GROUP]; int n; };
// defer != nullptr: the column reduce of (d gamma, d beta) is NOT launched; its jobs are appended to *defer for
// launch_splitk_reduce (bit-identical result: same partials, same fold order)
hipError_t launch_ln_tanh_bwd_group(LnBwdGroup g, hipStream_t s, ColReduceJobs* defer = nullptr);
// out = scale * y / max(||y||, 1e-12) per row (F.normalize, fb_modules.py:229); grouped like the LayerNorm launches
struct L2Problem { const float* y; int ldy; float* out; int ldo; float* norms; int rows, d; float scale; };
struct L2Group { L2Problem p[LN_MAX_GROUP]; int n; };
hipError_t launch_l2norm_fwd_group(const L2Group& g, hipStream_t s);
hipError_t launch_l2norm_fwd(const float* y, int ldy, float* out, int ldo, float* norms, int rows, int d,
                             float scale, hipStream_t s);
// policy head: mu = tanh(pre); action = clampST(mu + clip(noise*std))   (utils.py:171-185)
// boltzmann (fb_modules.py:129-151, utils.py:188-232): the policy head emits [loc | raw log-std] (2a wide),
//   log_std = lo + (hi - lo)/2 (tanh(raw) + 1),  action = tanh(loc + exp(log_std) eps)   (SquashedNormal, no clamp)
struct Squash { int on; float temp, lo, hi; };
hipError_t launch_policy_sample(const float* pre, int ldp, const float* noise, int ldn, float stddev, float clip,
                                float* mu, int ldmu, float* action, int lda, int rows, int a, Squash sq, hipStream_t s);
// boltzmann actor loss backward at the policy head (fb_ddpg.py:393, 399, 406): d action [rows,a] -> d [loc | raw] [rows,2a]
hipError_t launch_squash_head_bwd(const float* dact, int ldd, const float* pre, int ldp, const float* noise, int ldn,
                                  float* dpre, int ldo, int rows, int a, Squash sq, hipStream_t s);
// actor loss (fb_ddpg.py:400-406): Q = min(F1.z, F2.z); loss = -mean Q; dF_i = -z/B * w_i
// metrics == nullptr skips the (metric-only) finalize launch
hipError_t launch_actor_loss(const float* F1, const float* F2, int ldf, const float* z, int ldz,
                             const float* mu, int ldmu, const float* action, int lda, float stddev,
                             float* dF1, float* dF2, float* metrics, float* scratch /* >= 3*ceil(rows/4) floats */,
                             int rows, int d, int a, hipStream_t s, Squash sq = Squash{0, 1.f, -5.f, 2.f},
                             const float* pre = nullptr, int ldp = 0, const float* noise = nullptr, int ldn = 0);

// metrics -> pinned host memory + sequence number, from inside the step (metrics_publish_kernel)
hipError_t launch_metrics_publish(const float* metrics, float* host /* pinned: NUM_METRICS floats + 1 uint */, unsigned int* dseq,
                                  hipStream_t s);

// the update's actor phase: Q and d p straight from the heads' hidden activations p [rows, 2H] and V = z . W4 [rows, 2H]
// (overwritten with d p); see actor_q_kernel
struct StepState;
hipError_t launch_actor_q(const float* P, int ldp_, float* V, int ldv, const float* z, int ldz, const float* b41,
                          const float* b42, const float* mu, int ldmu, const float* action, int lda, float stddev,
                          float* metrics, float* scratch /* >= 3*ceil(rows/4) floats */, int rows, int H, int d, int a,
                          Squash sq, const float* pre, int ldp, const float* noise, int ldn, hipStream_t s,
                          StepState* adv = nullptr, int adv_which = 0 /* also does step_advance(adv, adv_which), see there */,
                          float* pub_host = nullptr, unsigned int* pub_seq = nullptr /* both set (and metrics): the finalize launch
                          also publishes every metric to the host, like launch_metrics_publish */);

// policy head (premu = p . W4^T + b4, na = a or 2a outputs) + policy_sample in one row kernel (policy_head_kernel)
bool policy_head_ok(int H, int na);
constexpr int PH_MAX_JOBS = 2;
struct PolicyHeadJob { const float* P; int ldp; float* premu; const float* noise; float* mu; float* action; int lda;
    // optional (base != nullptr): the FIRST layer of the ForwardMap trunk that consumes the action, finished in the same kernel.
    // ``base`` [rows, H] holds W1[:, :aoff] . x + b1 (a GEMM that does not need the action: off the dependency chain); the kernel
    // adds W1[:, aoff + j] action[j], applies LayerNorm + tanh (exactly ln_tanh_fwd_kernel's arithmetic) and writes t1; with
    // ``stats`` also the full pre-activation (in place, over base) and (mean, rstd) for a later LayerNorm backward.
    const float* base; int ldb; const float* W1a; int ldw1; const float* gamma; const float* beta; float* t1; int ldt1; float* stats; };
// the fused first layer needs H = 512 / 1024 / 2048 (and what the head itself needs: na * H floats of LDS within 48 KB)
bool policy_first_ok(int H, int a, int na);
hipError_t policy_head_prepare(int H, int a, int na);   // raises the dynamic-LDS limit (not inside a stream capture)
struct PolicyHeadJobs { PolicyHeadJob j[PH_MAX_JOBS]; int n; };
hipError_t launch_policy_head(const PolicyHeadJobs& jobs, const float* W4, int ldw4, const float* b4, int ldpre, int ldn,
                              float stddev, float clip, int ldmu, int rows, int H, int a, int na, Squash sq, hipStream_t s);
// the same two seams as 16-row MFMA tiles (headtiles.hip): no weight image in LDS; TruncatedNormal actor, a <= 16, and for the fused
// first layer / LayerNorm backward H = 512 / 1024.  FBHIP_HEAD_TILES=0 keeps the row kernels.  The launchers below dispatch.
bool policy_head_tiles_ok(const PolicyHeadJobs& jobs, int ldw4, int rows, int H, int a, int na, const Squash& sq);
hipError_t launch_policy_head_tiles(const PolicyHeadJobs& jobs, const float* W4, int ldw4, const float* b4, int ldpre, int ldn,
                                    float stddev, float clip, int ldmu, int rows, int H, int a, hipStream_t s);
bool actor_head_bwd_tiles_ok(int ldt, int ldy, int ldx, int ldp_, int lddp, int rows, int H, int a, const void* p0, const void* p1,
                             const void* p2, const void* p3, const void* p4, const void* p5);
hipError_t launch_actor_head_bwd_tiles(const float* dt1, int ldt, const float* W1a, int ldw1, const float* mu, int ldmu, const float* W4,
                                       int ldw4, const float* P, int ldp_, float* dpremu, int ldd, float* dp, int lddp, int rows, int H,
                                       int a, hipStream_t s, const float* lnY, int ldy, const float* lnX, int ldx, const float* lnStats,
                                       const float* lnGamma);
// d action -> d premu -> d p of the actor's policy hidden layer in one row kernel (actor_head_bwd_kernel; a <= 16)
bool actor_head_bwd_ok(int H, int a);
hipError_t actor_head_bwd_prepare(int H, int a);   // raises the kernel's dynamic-LDS limit (not inside a stream capture)
// lnY != nullptr: dt1 is dL/d(tanh output) and the LayerNorm+tanh backward (y = lnY, x = lnX, (mean, rstd) = lnStats, gamma;
// no parameter gradients) is done by the kernel as well; H <= 2048
hipError_t launch_actor_head_bwd(const float* dt1, int ldt, const float* W1a, int ldw1, const float* mu, int ldmu,
                                 const float* W4, int ldw4, const float* P, int ldp_, float* dpremu, int ldd, float* dp,
                                 int lddp, int rows, int H, int a, hipStream_t s, const float* lnY = nullptr, int ldy = 0,
                                 const float* lnX = nullptr, int ldx = 0, const float* lnStats = nullptr,
                                 const float* lnGamma = nullptr);

// ---- pairwise FB loss ----------------------------------------------------------------------------------
size_t pairwise_scratch_floats(int B, int d);
hipError_t launch_pairwise_fb(const float* F1, const float* F2, const float* Bm, const float* tF1,
                              const float* tF2, const float* tB, const float* discount, int B, int d, int ld,
                              float ortho_coef, float* dF1, float* dF2, float* dB, float* metrics,
                              float* scratch, hipStream_t s, StepState* adv = nullptr, int adv_which = 0,
                              const float* y = nullptr /* + norms, dy: also the backward of B = sqrt(d) y/|y| (rows [.,ld]) */,
                              const float* norms = nullptr, float* dy = nullptr,
                              float out_scale = 1.f /* every gradient (and dy) times this: losses that are a multiple of the FB form */);
// rows [row_off, row_off + rows) of the same loss on B-row panels (global-batch data parallel); outputs are [rows, ld]
hipError_t launch_pairwise_fb_block(const float* F1, const float* F2, const float* Bm, const float* tF1,
                                    const float* tF2, const float* tB, const float* discount, int B, int d, int ld,
                                    float ortho_coef, int row_off, int rows, float* dF1, float* dF2, float* dB,
                                    float* metrics, float* scratch, hipStream_t s, StepState* adv = nullptr, int adv_which = 0, const float* y = nullptr,
                                    const float* norms = nullptr, float* dy = nullptr, float out_scale = 1.f);
// dy = (sqrt(d)/||y||) (dB - yhat (yhat . dB))      (F.normalize backward)
hipError_t launch_l2norm_bwd(const float* dB, int lddb, const float* y, int ldy, const float* norms,
                             float* dy, int lddy, int rows, int d, hipStream_t s);

// ---- optimiser -------------------------------------------------------------------------------------------
struct StepState {          // device-resident, advanced in-graph
    int fb_t;               // Adam step counts (1-based after the first step)
    int actor_t;
    unsigned int update_count;   // RNG counter: number of update() calls so far
    unsigned int act_count;      // RNG counter of the batch-1 act() fast path (exploration noise drawn on device)
    double fb_bc1;          // 1 - beta1^t          (fp64 like torch's python-side scalars)
    double fb_bc2_sqrt;     // sqrt(1 - beta2^t)
    double actor_bc1;
    double actor_bc2_sqrt;
};
constexpr int EXTRA_METRICS_MAX_BLOCKS = 128;   // part: [EXTRA_METRICS_MAX_BLOCKS][4] doubles (64 rows per workgroup: batch <= 8192)
hipError_t launch_extra_metrics(const float* F1, const float* Bm, const float* z, int ld, int rows, int d,
                                const float* cov, int ldc, float* metrics, hipStream_t s, int cov_rows = 0 /* 0: rows */,
                                double* part = nullptr, unsigned int* ticket = nullptr /* both set: the many-workgroup form */);
// DiscreteFBAgent heads, [rows, d * A] with (k, a) at column k * A + a (discrete_fb.py:289-311): target-side selection
// (greedy column or softmax mix, + next_Q and the greedy index), online-side gather of the taken action's column, and the
// gather's backward
struct DiscreteHeadJob { const float* Fall1; const float* Fall2; const float* z; const float* act_idx; float* out1; float* out2;
                         float* nextq; int32_t* act_out; int gather; };
struct DiscreteHeadJobs { DiscreteHeadJob j[2]; int n; };
hipError_t launch_discrete_heads(const DiscreteHeadJobs& jobs, int ldfa, int ldz, int ldo, int rows, int d, int A, int boltz,
                                 float temp, hipStream_t s);
hipError_t launch_discrete_select(const float* Fall1, const float* Fall2, int ldfa, const float* z, int ldz, float* out1,
                                  float* out2, int ldo, float* nextq, int32_t* act_out, int rows, int d, int A, int boltz,
                                  float temp, hipStream_t s);
hipError_t launch_discrete_gather(const float* Fall1, const float* Fall2, int ldfa, const float* act_idx, float* out1,
                                  float* out2, int ldo, int rows, int d, int A, hipStream_t s);
hipError_t launch_discrete_scatter(const float* dF1, const float* dF2, int ldf, const float* act_idx, float* dFall1,
                                   float* dFall2, int ldfa, int rows, int d, int A, hipStream_t s);
// ---- SFAgent (sf.hip) ------------------------------------------------------------------------------------------------
// TD regression on successor features (sf.py:594-626): writes dF1, dF2 [rows, d] (ld) and SF_LOSS, SF_TARGET_F, F1, SF_PHI,
// SF_PHI_NORM, Z_NORM into metrics; scratch >= 6 * ceil(rows / 4) floats
hipError_t launch_sf_loss(const float* F1, const float* F2, const float* nF1, const float* nF2, const float* phi_next,
                          const float* z, int ld, const float* discount, int q_loss, float* dF1, float* dF2, float* metrics,
                          float* scratch, int rows, int d, hipStream_t s);
// inverse-dynamics loss of the ICM feature learner (sf.py:207-210): pred = tanh(pre), PHI_LOSS = mean((action - pred)^2), d pre;
// scratch >= ceil(rows * a / 256) floats
hipError_t launch_scale_metric(float* metrics, int src, int dst, float scale, hipStream_t s, int accumulate = 0);   // metrics[dst] (+)= scale * metrics[src]
// contrastive (sf.py:134-142): L[s, t] = phi_s . mu_t (raw), rows = cols = B.  In place: L <- d loss / d L; loss = mean_s(-l_ss + logsumexp_{t != s} l_st),
// l = L / d (both embeddings have norm sqrt(d): the cosine) -> metrics[FBHIP_M_PHI_LOSS]
hipError_t launch_contrastive_rows(float* L, int ld, int B, int d, float* metrics, float* scratch, hipStream_t s);
hipError_t launch_fill_add(float* dst, const float* add, float fill, int64_t n, hipStream_t s);   // add ? dst[i] += add[i] : dst[i] = fill
hipError_t launch_icm_loss(const float* pre, int ldp, const float* action, int lda, float* dpre, int ldd, int rows, int a,
                           int squash, float* metrics, float* scratch, hipStream_t s);
// Laplacian feature learner (sf.py:104-114) on top of pairwise_kernel's orthonormality pass: dphi += d mse, dnext_phi = d mse,
// PHI_LOSS = mse + metrics[ORTH_LOSS]; scratch >= ceil(rows / 4) floats
hipError_t launch_lap(const float* phi, const float* next_phi, int ld, float* dphi, float* dnext_phi, float* metrics,
                      float* scratch, int rows, int d, hipStream_t s);
hipError_t launch_inverse(const float* A, int lda, int d, float scale, float* out, int ldo, hipStream_t s);
hipError_t inverse_prepare();
hipError_t launch_qloss(const float* F1, const float* F2, const float* tF1, const float* tF2, const float* BinvC,
                        const float* z, int ld, const float* discount, float coef, float* dF1, float* dF2,
                        float* metrics, float* scratch /* >= ceil(rows/4) floats */, int rows, int d, hipStream_t s,
                        int norm_rows = 0 /* the mean's row count when ``rows`` is a block of a larger batch */,
                        const float* nextq = nullptr /* [rows] next_Q given (DiscreteFBAgent) instead of min(tF1.z, tF2.z) */);
// Adam step counters + fp64 bias corrections (which: 0 fb, 1 actor, 3 both, 2 rng counter).  A launch of its own
// (launch_step_advance) or the first thread of a kernel that precedes the optimiser pass in the same stream anyway
// (pairwise_reduce_kernel, actor_q_kernel: ``adv`` arguments) -- one dispatch less on the dependency chain.
__device__ inline void step_advance_device(StepState* st, int which) {
    if (which == 2) { st->update_count += 1u; return; }
    if (which == 0 || which == 3) {
        const int t = ++st->fb_t;
        st->fb_bc1 = 1.0 - pow(0.9, (double)t);
        st->fb_bc2_sqrt = sqrt(1.0 - pow(0.999, (double)t));
    }
    if (which == 1 || which == 3) {
        const int t = ++st->actor_t;
        st->actor_bc1 = 1.0 - pow(0.9, (double)t);
        st->actor_bc2_sqrt = sqrt(1.0 - pow(0.999, (double)t));
    }
}
hipError_t launch_step_advance(StepState* st, int which /*0 fb, 1 actor, 2 rng*/, hipStream_t s);
hipError_t launch_adam_ema(float* p, const float* g, float* m, float* v, float* target, int64_t numel,
                           float lr, float lr2, int64_t split, float grad_scale, float tau,
                           const StepState* st, int which, int t_explicit, hipStream_t s,
                           float tau2 = -1.f /* the target rate of the elements behind ``split`` (< 0: tau) */,
                           int ema_before2 = 0 /* there, the target follows the parameter as it was BEFORE this step */);

// ---- peer-access all-reduce (peer.hip): the data-parallel gradient exchange as graph-capturable kernels -------------------
constexpr int PEER_MAX_WORLD = 8;
constexpr int PEER_TIMEOUT = 1;
struct PeerState { int epoch[3]; int status; };             // device-resident, this rank's (zero-initialised by the host)
struct PeerComm {
    int world, rank;
    float* bucket[2][PEER_MAX_WORLD];                       // [fb | actor][rank]: every rank's gradient bucket, mapped into this process
    int* flags[PEER_MAX_WORLD];                             // [rank]: int32[3][PEER_MAX_WORLD] barrier slots of every rank
    PeerState* state;
};
hipError_t launch_peer_allreduce(const PeerComm& pc, int which, int64_t numel, hipStream_t s);

// ---- batch-1 inference (infer.hip) -----------------------------------------------------------------------------
// y[n] = (relu)( W[n, :K] . f(x) + bias[n] ), f = identity or tanh(LayerNorm(x[:n_ln])) (entries >= n_ln read as 0)
struct GemvProblem { const float* x; const float* W; const float* bias; float* y; const float* ln_g; const float* ln_b;
                     int N, K, ldw, n_ln, relu, block_start; };
constexpr int GEMV_MAX_GROUP = 4;
struct GemvGroup { GemvProblem p[GEMV_MAX_GROUP]; int n; };
hipError_t launch_gemv_group(GemvGroup g, hipStream_t s);
hipError_t launch_act_head(const float* x, const float* W, int ldw, const float* bias, int a, int K, float stddev,
                           int eval_mode, const float* noise, uint64_t seed, uint32_t rank, StepState* st, float* out,
                           Squash sq, hipStream_t s, const float* stddev_dev = nullptr /* device-resident stddev (replayable graphs) */,
                           float* host_out = nullptr, unsigned int* dseq = nullptr);
hipError_t launch_zcorrel(const float* y, const float* z, int d, int project, float* out, hipStream_t s, float* host_out = nullptr,
                          unsigned int* dseq = nullptr);
// host_out + dseq (both set): the result ALSO goes straight into pinned host memory, then a sequence number into slot INFER_SEQ_SLOT
// behind a system-scope fence: the batch-1 caller spins on the number instead of a D2H copy node + hipStreamSynchronize (api.hip)
constexpr int INFER_SEQ_SLOT = 63;

// ---- sampler -----------------------------------------------------------------------------------------------
struct ReplayView {
    const float* observation; const float* action; const float* discount; const float* goal;
    const int32_t* episode_len; const int64_t* cum_len;
    int n_episodes, t1, fixed_length;
};
struct SampleOut {          // all device pointers into the workspace
    int32_t* ep_idx; int32_t* step_idx; int32_t* perm; float* mix_uniform;
    float* z_gauss; float* eps_next; float* eps_actor;
    int32_t* future_idx; float* future_uniform;       // hindsight replay (only drawn when future_ratio > 0)
    float* z_uniform;                                 // [B,d] uniform factor of sample_z (only drawn when norm_z == 0)
};
// future < 0: no hindsight draws; else future_idx = clip(step_idx + Geometric(1 - future), 0, len) and a uniform
hipError_t launch_draw(const ReplayView& rv, const SampleOut& so, int B, int d, int a,
                       uint64_t seed, uint32_t rank, const StepState* st, float future, int norm_z,
                       hipStream_t s);
struct GatherArgs {
    ReplayView rv;
    const int32_t* ep_idx; const int32_t* step_idx; const int32_t* perm;      // perm == nullptr: identity
    float* Xoa; int ld_oa;      // [obs | action]
    float* Xoz; int ld_oz;      // [obs | z]          (z filled later)
    float* Xnoz; int ld_noz;    // [next_obs | z]
    float* Xnoa; int ld_noa;    // [next_obs | next_action]  (action filled later)
    float* Xopi; int ld_opi;    // [obs | pi action]         (action filled later)
    float* Xo; int ld_o;        // [obs] alone (zero padded: operand of the actor's obs_net weight gradient)
    float* next_goal; int ld_ng;    // goal[ep, step] if use_goal else next_obs
    float* bin; int ld_bin;     // backward_input[perm]
    float* pgoal; int ld_pg;    // nullable (SFAgent's z-mix, sf.py:726-727): next_goal[pperm]
    const int32_t* pperm;
    const int32_t* future_idx;  // nullable: hindsight replay off
    float* fgoal; int ld_fg;    // (goal if use_goal else observation)[ep, future_idx - 1]
    float* disc;
    int B, o, a, g, use_goal; float gamma;
    int aoff;                   // column of the action inside Xoa (o, or o + z for the [obs|z|action] panels of preprocess == 0)
    float* act_idx;             // DiscreteFBAgent: action[ep, step] is ONE stored float holding the action index -> act_idx[i];
                                // no action columns in the panels (nullptr otherwise)
};
hipError_t launch_gather(const GatherArgs& ga, hipStream_t s);
// z[i] = mix ? sqrt(d) normalize(sqrt(d) normalize(ymix[i])) : sqrt(d) normalize(gauss[i]); scattered into the concat
// panels; advances the RNG counter when st != nullptr
// hindsight rows (future_uniform[i] < future_ratio) take sqrt(d) normalize(yfut[i]) instead (fb_ddpg.py:487-491)
struct ZPanels { float* p[3]; int ld[3]; };
hipError_t launch_mix_z(const float* gauss, int ldg, const float* ymix, int ldy, const float* mix_uniform, float mix_ratio,
                        float* z, int ldz, float* Xoz, int ld_oz, float* Xnoz, int ld_noz, int o, int B, int d,
                        StepState* st, const float* yfut, const float* future_uniform, float future_ratio,
                        const float* z_uniform /* nullable: norm_z */, int mix_projections /* 2, or 1 with rand_weight */,
                        ZPanels extra /* further panels that carry z at column o (entries may be null) */, hipStream_t s);
// rand_weight rows (fb_ddpg.py:477-480): W[i, :] = u[i] * raw[i, :] / max(|raw[i, :]|_2, 1e-12), in place.  generate != 0
// first draws raw[i, j] and u[i] ~ U(0,1) (Philox); else they are the injected values already sitting in W / u.
hipError_t launch_rand_weight(float* W, float* u, int B, int generate, uint64_t seed, uint32_t rank, const StepState* st,
                              hipStream_t s);
hipError_t launch_concat2(float* dst, int ld, const float* A, int lda, int na, const float* B, int ldb, int nb,
                          int rows, hipStream_t s);
hipError_t pairwise_prepare(int B, int d);     // one-time kernel attribute setup (outside graph capture)

inline int pad4(int x) { return (x + 3) & ~3; }

}  // namespace fbhip
