/*
 * fbhip.h -- C ABI of libfbhip.so: the MI355X (gfx950) implementation of the FB-DDPG update hot path
 * of facebookresearch/controllable_agent.
 *
 * The reference has no FFI (it is 100 % Python on top of torch ATen ops); what this library replaces is
 * the *sequence of ATen ops* issued by
 *     url_benchmark/agent/fb_ddpg.py:427-520   FBDDPGAgent.update            -> fbhip_update
 *     url_benchmark/agent/fb_ddpg.py:291-387   FBDDPGAgent.update_fb         -> (inside fbhip_update, phase FB)
 *     url_benchmark/agent/fb_ddpg.py:389-421   FBDDPGAgent.update_actor      -> (inside fbhip_update, phase ACTOR)
 *     url_benchmark/in_memory_replay_buffer.py:139-190  ReplayBuffer.sample  -> fbhip_replay_bind + sampler stage
 *     url_benchmark/agent/fb_modules.py:81-230 Actor/ForwardMap/BackwardMap  -> fbhip_actor_forward / fbhip_backward_map / ...
 *     url_benchmark/utils.py:66-69             soft_update_params            -> fused into fbhip_adam_ema
 *     torch.optim.Adam (fb_ddpg.py:146-151)                                  -> fbhip_adam_ema
 * The Python host side (controllable_agent_amd/) binds these with ctypes and mirrors the reference's Agent /
 * ReplayBuffer plugin surface; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (from torch ``tensor.data_ptr()``) unless named ``host_*``;
 *   - ``stream`` is a ``hipStream_t`` passed as ``void*`` (``torch.cuda.current_stream().cuda_stream``);
 *   - the caller (torch) OWNS all memory; the library never allocates or frees device memory for data,
 *     never synchronises the device behind the caller's back (except the explicitly blocking
 *     ``fbhip_read_metrics``), and launches everything asynchronously on the caller's stream;
 *   - every function returns 0 on success or a negative FBHIP_E_* code; the message is available from
 *     ``fbhip_last_error``; nothing is ever thrown across the ABI;
 *   - all matrices are float32 row-major with an explicit leading dimension (in floats).
 */
#ifndef FBHIP_H
#define FBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBHIP_ABI_VERSION 19

enum {
    FBHIP_OK = 0,
    FBHIP_E_INVALID = -1,     /* bad argument / unsupported shape */
    FBHIP_E_HIP = -2,         /* a HIP runtime call failed        */
    FBHIP_E_STATE = -3,       /* buffers / replay not bound       */
    FBHIP_E_NODEVICE = -4     /* no gfx950 device visible         */
};

/* nets (flat-buffer segments); fb_ddpg.py:122-141 */
enum { FBHIP_NET_FORWARD = 0, FBHIP_NET_BACKWARD = 1, FBHIP_NET_ACTOR = 2 };

/* phases of one update(); a mask selects which are enqueued (multi-GPU inserts all-reduces between them) */
enum {
    FBHIP_PHASE_SAMPLE = 1,      /* replay gather + z sampling + z mixing           (fb_ddpg.py:433-491) */
    FBHIP_PHASE_FB_FWD_ONLINE = 2,   /* online F(obs, z, action), the online / target B(next_goal) passes (fb_ddpg.py:312, 318-319) and
                                      * what forward_target computes without next_action (its obs_z trunk, the action-free part of
                                      * its obs_action trunk's first layer): need this step's batch and z but NOT the previous actor
                                      * step.  Runs before FB_FWD_TARGET of the same update (same call or an earlier one) */
    FBHIP_PHASE_FB_FWD_TARGET = 128, /* actor(next_obs) -> next_action -> forward_target (fb_ddpg.py:303-311) */
    FBHIP_PHASE_FB_FWD = 130,    /* = FB_FWD_ONLINE | FB_FWD_TARGET: everything up to the six embeddings F1 F2 B tF1 tF2 tB */
    FBHIP_PHASE_FB_BWD_A = 64,   /* pairwise loss (on the rows bound by fbhip_bind_global_batch when a global batch is bound) + the
                                  * first two backward rounds: they complete the gradients of the ForwardMap heads' hidden layers
                                  * (fbhip_fb_early_grad_range) */
    FBHIP_PHASE_FB_BWD_B = 256,  /* the rest of fb_loss.backward() (fb_ddpg.py:383) */
    FBHIP_PHASE_FB_BWD = 320,    /* = FB_BWD_A | FB_BWD_B */
    FBHIP_PHASE_FB_GRAD = 450,   /* = FB_FWD | FB_BWD */
    FBHIP_PHASE_FB_STEP = 4,     /* fb_opt.step() + both soft_update_params         (fb_ddpg.py:384,500-503) */
    FBHIP_PHASE_ACTOR_GRAD = 8,  /* Q through the UPDATED forward_net, actor backward  (fb_ddpg.py:398-410) */
    FBHIP_PHASE_ACTOR_STEP = 16, /* actor_opt.step()                                (fb_ddpg.py:411) */
    /* the actor's own forward pass of update_actor (fb_ddpg.py:395-397): it reads only the actor weights and (obs, z), so it
     * runs wherever it is cheapest in the call that carries this bit: with FB_FWD_TARGET (layer by layer in the same launches
     * as the target chain's actor(next_obs)), else with FB_BWD (sharing the FB backward's launches; pass the bit with both
     * halves then), else with ACTOR_GRAD.  An ACTOR_GRAD call WITHOUT this bit uses the pass of an earlier call on the same
     * batch.  Pass it to ONE place per step. */
    FBHIP_PHASE_ACTOR_FWD = 32,
    FBHIP_PHASE_ALL = 511
};

/* Both parameter structs start with their own size: the caller sets ``struct_size = sizeof(fbhip_dims)`` (resp.
 * ``sizeof(fbhip_hparams)``) as compiled against ITS copy of this header; every entry point that takes the struct
 * compares it with the library's own sizeof and returns FBHIP_E_INVALID on a mismatch, so a binding written against an
 * older header (a field short) is refused instead of being read past its end. */
typedef struct fbhip_dims {
    uint32_t struct_size;          /* = sizeof(fbhip_dims)                               */
    int32_t batch;                 /* B   cfg.batch_size                                 */
    int32_t obs_dim;               /* o                                                 */
    int32_t action_dim;            /* a                                                 */
    int32_t goal_dim;              /* g   == obs_dim when goal_space is None            */
    int32_t z_dim;                 /* d                                                 */
    int32_t hidden_dim;            /* H   (1024)                                        */
    int32_t feature_dim;           /* Fd  (512)                                         */
    int32_t backward_hidden_dim;   /* Hb  (526)                                         */
    int32_t use_goal;              /* goal_space is not None: B-net input = goal        */
    int32_t add_trunk;             /* cfg.add_trunk (default 0): Linear(2Fd, H) + ReLU "trunk" between the two preprocess nets and
                                    * the F1/F2 heads / the policy head (fb_modules.py:93-98, 168-173); their first layer is
                                    * then [H, H] instead of [H, 2Fd] */
    int32_t preprocess;            /* cfg.preprocess (default 1).  0: ForwardMap / Actor are ONE trunk mlp(in, H, "ntanh", H, "irelu",
                                    * H, "irelu") on cat([obs, z, action]) / cat([obs, z]) instead of the two preprocess nets
                                    * (fb_modules.py:99-103, 174-178); add_trunk is then ignored */
    int32_t norm_z;                /* cfg.norm_z (default 1).  0: BackwardMap output unprojected (fb_modules.py:228-229),
                                    * z = sqrt(d) U g/|g| (fb_ddpg.py:229-231), no re-projection of mixed rows (:483) */
    int32_t boltzmann;             /* cfg.boltzmann (default 0).  1: the actor is DiagGaussianActor (fb_modules.py:129-151; net
                                    * layout policy.{0,1,3,5}, head 2a wide) with a SquashedNormal policy (utils.py:188-232):
                                    * next_action = dist.sample(), update_actor uses dist.rsample() and
                                    * actor_loss = (temp * log_prob - Q).mean() (fb_ddpg.py:304-306, 391-393, 406);
                                    * stddev / stddev_clip are ignored; temp and log_std_bounds: fbhip_set_policy_squash */
    int32_t discrete;              /* 0: FBDDPGAgent.  1: the sibling DiscreteFBAgent (url_benchmark/agent/discrete_fb.py:103-468):
                                    * action_dim is A, the NUMBER of actions; the replay's ``action`` storage holds ONE float per
                                    * transition, the action index; ForwardMap = one trunk on cat([obs, z]) (preprocess must be 0,
                                    * discrete_fb.py:74-83) whose heads F{1,2}.2 emit z_dim * A values, element (k, a) at k * A + a
                                    * (:100); there is NO actor (its layout is empty, the four actor pointers of fbhip_bind_buffers
                                    * are ignored, the ACTOR_* phases are no-ops).  Target embedding = the greedy action's column
                                    * of the target ForwardMap on next_obs, or with ``boltzmann`` the softmax(next_Q / temp) mix
                                    * (temp: fbhip_set_policy_squash; :289-303); online embedding = the column of the stored action
                                    * (:309-311); q_loss uses that next_Q (:329).  Greedy actions: fbhip_discrete_act */
    int32_t sf;                    /* 0: FBDDPGAgent.  1..12: the sibling SFAgent (url_benchmark/agent/sf.py:383-768) with
                                    * feature_learner "icm" (1, sf.py:194-213) / "lap" (2, :100-116) / "random" (3, :84-92: feature_net
                                    * frozen, no feature loss) / "autoencoder" (4, :249-262) / "transition" (5, :215-227) / "svd_p"
                                    * (6, :337-362: a second LayerNorm mlp ``mu_net.{0,1,3,5}`` on cat[goal, action] behind feature_net)
                                    * / "latent" (7, :230-246: ``forward_dynamic_net`` z_dim + action_dim -> z_dim; its ``target_feature_net``
                                    * is the feature block of the TARGET buffer: own initial weights, moved at 0.01 towards the pre-step
                                    * feature_net by the optimiser pass) / "svd_sr" (8, :264-299) and "svd_srv2" (9, :303-335):
                                    * ``mu_net.{0,1,3,5}`` on the goal alone; ``target_feature_net`` and ``target_mu_net`` = the TARGET
                                    * buffer's copies of both blocks, moved like latent's / "contrastive" (10, :118-143): ``mu_net`` =
                                    * feature_net's modules (projection included) on the hindsight goal; hparams.future must be < 1 /
                                    * "contrastivev2" (11, :159-186): mu_net on the goal, feature_net also on the hindsight goal /
                                    * "identity" (12, :94-98): phi(goal) = goal, z_dim == goal_dim, the feature block unused.  NET_FORWARD is
                                    * ``successor_net`` (the same ForwardMap; its target = successor_target_net), NET_BACKWARD is
                                    * ``feature_learner``: ``feature_net.{0,1,3,5}`` (the BackwardMap architecture, projection
                                    * included) followed by the learner's head mlp(in, Hb, relu, Hb, relu, out) at Sequential indices
                                    * {0,2,4}: icm ``inverse_dynamic_net`` (2 z_dim -> action_dim, tanh), autoencoder ``decoder``
                                    * (z_dim -> goal_dim), transition ``forward_dynamic_net`` (z_dim + action_dim -> goal_dim); both optimisers of the reference map onto the two lr groups of the FB
                                    * flat buffer (sf_opt: lr; phi_opt: lr_coef * lr).  The critic loss is the TD regression of
                                    * sf.py:594-626 (hparams.q_loss: scalar Q regression (the reference default) or feature space);
                                    * z is sample_z; with hparams.mix_ratio > 0 the rows drawn by the mix uniform take
                                    * sqrt(d) normalize(phi(next_goal[perm]) @ inverse(phi^T phi / batch)) (sf.py:725-739; batch >= z_dim);
                                    * needs norm_z = 1, boltzmann = 0, discrete = 0.  The actor phase is FBDDPGAgent's. */
    int32_t backward_identity;     /* cfg.debug (fb_ddpg.py:128-130; default 0).  1: backward_net and backward_target_net are IdentityMap
                                    * (fb_modules.py:202-208): B(goal) = goal, unprojected whatever norm_z says, nothing to train;
                                    * needs z_dim == goal_dim and sf = 0 (DiscreteFBAgent has the same switch, discrete_fb.py:134-136).  The
                                    * NET_BACKWARD block keeps its place in the flat buffers (unused, its gradients stay zero). */
} fbhip_dims;

typedef struct fbhip_hparams {     /* FBDDPGAgentConfig fields, fb_ddpg.py:47-82 */
    uint32_t struct_size;          /* = sizeof(fbhip_hparams) */
    float lr;                      /* 1e-4   */
    float lr_coef;                 /* 1      */
    float fb_target_tau;           /* 0.01   */
    float stddev;                  /* utils.schedule(stddev_schedule, step), utils.py:235-255 */
    float stddev_clip;             /* 0.3    */
    float ortho_coef;              /* 1.0    */
    float mix_ratio;               /* 0.5    */
    float q_loss_coef;             /* 0.01   */
    float discount;                /* ReplayBuffer._discount (in_memory_replay_buffer.py:171) */
    float grad_scale;              /* 1/world_size when gradients were sum-all-reduced, else 1 */
    int32_t q_loss;                /* 0/1    */
    int32_t want_metrics;          /* compute the full metric set of fb_ddpg.py:356-377 */
    float future_ratio;            /* 0: off.  > 0: hindsight replay z[u < future_ratio] = B(future_goal) (fb_ddpg.py:487-491) */
    float future;                  /* ReplayBuffer._future (< 1 when future_ratio > 0): future_idx = clip(step_idx +
                                    * Geometric(1 - future), 0, episode_len) (in_memory_replay_buffer.py:157-161) */
    int32_t rand_weight;           /* cfg.rand_weight: mixed rows = (u * normalize(rand[B])) @ B(backward_input[perm]) (:475-482) */
} fbhip_hparams;

/* Injected random draws (parity mode).  NULL struct => draw on device with Philox4x32-10 keyed by
 * (seed, rank, on-device step counter).  Order = the order the reference draws them. */
typedef struct fbhip_inject {
    const int32_t* ep_idx;         /* [B]   in_memory_replay_buffer.py:147-151 */
    const int32_t* step_idx;       /* [B]   in_memory_replay_buffer.py:155 (1-based)  */
    const float* z_gauss;          /* [B,d] contiguous; fb_ddpg.py:225 */
    const int32_t* perm;           /* [B]   fb_ddpg.py:467 */
    const float* mix_uniform;      /* [B]   fb_ddpg.py:471 */
    const float* eps_next;         /* [B,a] contiguous; utils.py:178 via fb_ddpg.py:310 */
    const float* eps_actor;        /* [B,a] contiguous; utils.py:178 via fb_ddpg.py:397 */
    const int32_t* future_idx;     /* [B]   in_memory_replay_buffer.py:159-160 (1-based, clipped); used if future_ratio > 0 */
    const float* future_uniform;   /* [B]   fb_ddpg.py:490; used if future_ratio > 0 */
    const float* z_uniform;        /* [B,d] contiguous; torch.rand of fb_ddpg.py:230; used if norm_z == 0 */
    const float* rand_weight;      /* [B,B] contiguous RAW uniforms, row i = batch row i (fb_ddpg.py:477); used if rand_weight */
    const float* rand_weight_u;    /* [B]   fb_ddpg.py:479; used if rand_weight */
} fbhip_inject;

/* one named tensor inside a flat parameter buffer (names = the reference's state_dict keys) */
typedef struct fbhip_tensor_desc {
    char name[48];
    int64_t offset;                /* floats from the start of the net's segment */
    int32_t rows, cols, ld;        /* vectors: rows = 1 */
} fbhip_tensor_desc;

#define FBHIP_NUM_METRICS 32
/* indices into the device metrics array (fb_ddpg.py:356-377, 413-418) */
enum {
    FBHIP_M_TARGET_M = 0, FBHIP_M_M1, FBHIP_M_F1, FBHIP_M_B, FBHIP_M_B_NORM, FBHIP_M_Z_NORM,
    FBHIP_M_FB_LOSS, FBHIP_M_FB_DIAG, FBHIP_M_FB_OFFDIAG, FBHIP_M_Q_LOSS, FBHIP_M_ORTH_LOSS,
    FBHIP_M_ORTH_LOSS_DIAG, FBHIP_M_ORTH_LOSS_OFFDIAG, FBHIP_M_ORTH_LINF, FBHIP_M_ORTH_L2,
    FBHIP_M_ACTOR_LOSS, FBHIP_M_Q, FBHIP_M_ACTOR_LOGPROB,
    FBHIP_M_Q1_SUCCESS,            /* (Q1 > Q2).mean() of update_actor, reported when cfg.additional_metric (fb_ddpg.py:403-404, 417) */
    /* dims.sf (SFAgent, sf.py:627-636): sf_loss, target_F, phi, phi_norm, phi_loss; F1 and z_norm use the slots above */
    FBHIP_M_SF_LOSS, FBHIP_M_SF_TARGET_F, FBHIP_M_SF_PHI, FBHIP_M_SF_PHI_NORM, FBHIP_M_PHI_LOSS,
    FBHIP_M_COUNT
};

typedef struct fbhip_ctx fbhip_ctx;

/* ---- library / layout --------------------------------------------------------------------------- */
int fbhip_abi_version(void);
/* 1 when this process may replay graphs with parallel branches (the pipelined n-step graph of fbhip_update_many),
 * 0 when the library builds its graphs single-queue instead; *why (nullable) receives a static one-line reason.  ROCm 7.0's
 * hipGraphLaunch can walk off an exec's parallel-stream list; the library launches branched graphs from a high-priority stream,
 * which is verified to avoid it on HIP 7.0 and unnecessary on HIP >= 7.2 -- on any other runtime version, or without a
 * high-priority stream class, branched graphs are refused (same kernels and results, 3-6 % slower) rather than run unguarded.
 * FBHIP_BRANCHED_GRAPHS=1 forces them, =0 refuses them (read at every call). */
int fbhip_branched_graphs(const char** why);
const char* fbhip_last_error(const fbhip_ctx* ctx);            /* ctx may be NULL: last global error   */
int fbhip_device_ok(void);                                     /* 0 iff a gfx950 device is current     */

int64_t fbhip_net_numel(const fbhip_dims* dims, int net);      /* padded floats of one net's segment   */
int64_t fbhip_net_param_count(const fbhip_dims* dims, int net);/* logical parameter count (reference)  */
int fbhip_layout_count(const fbhip_dims* dims, int net);
int fbhip_layout_entry(const fbhip_dims* dims, int net, int idx, fbhip_tensor_desc* out);
size_t fbhip_workspace_bytes(const fbhip_dims* dims);

/* ---- context ------------------------------------------------------------------------------------- */
int fbhip_create(const fbhip_dims* dims, fbhip_ctx** out);
int fbhip_destroy(fbhip_ctx* ctx);       /* waits for the device first (work of this context may still be in flight) */
/* FB flat buffers hold forward_net ++ backward_net (fb_opt's two param groups, fb_ddpg.py:149-151);
 * fb_targets holds forward_target_net ++ backward_target_net in the same layout.
 * ZERO-INITIALISE before binding: (1) every flat buffer -- the physical layout pads matrices (fbhip_tensor_desc.ld >
 * cols, extra zero rows of backward_net) and the kernels rely on pad weights / gradients / Adam moments being and
 * staying zero; write parameters through the (offset, rows, cols, ld) views only; (2) the workspace -- it holds the
 * zero pad columns of the input panels, the device step counters (Adam t, RNG counters: fbhip_set_step_counts) and
 * the metrics.  The library never allocates, frees or clears these buffers itself. */
int fbhip_bind_buffers(fbhip_ctx* ctx,
                       float* fb_params, float* fb_grads, float* fb_adam_m, float* fb_adam_v, float* fb_targets,
                       float* actor_params, float* actor_grads, float* actor_adam_m, float* actor_adam_v,
                       void* workspace, size_t workspace_bytes);
/* Device-resident replay storage, episode-major float32[n_episodes, t1, dim] exactly like
 * ReplayBuffer._storage (in_memory_replay_buffer.py:119-125).  goal may be NULL when !use_goal.
 * episode_len int32[n_episodes]; cum_len int64[n_episodes+1] (exclusive prefix sum; used when lengths vary). */
int fbhip_replay_bind(fbhip_ctx* ctx, const float* observation, const float* action, const float* discount,
                      const float* goal, const int32_t* episode_len, const int64_t* cum_len,
                      int32_t n_episodes, int32_t t1, int32_t fixed_length);
int fbhip_set_seed(fbhip_ctx* ctx, uint64_t seed, uint32_t rank);
/* boltzmann contexts only: cfg.temp and cfg.log_std_bounds (fb_ddpg.py:70-71; defaults 1, (-5, 2) are in effect until
 * this is called).  Drops every captured graph (the values are baked into the launches). */
int fbhip_set_policy_squash(fbhip_ctx* ctx, float temp, float log_std_min, float log_std_max);
int fbhip_set_step_counts(fbhip_ctx* ctx, int32_t fb_steps, int32_t actor_steps, void* stream); /* Adam t */
int fbhip_get_step_counts(fbhip_ctx* ctx, int32_t* host_fb_steps, int32_t* host_actor_steps, void* stream);
/* The device-side Philox counters: number of update() calls drawn so far (sampler / z / noise streams) and number of act()
 * calls of the batch-1 fast path.  A checkpoint that restores them (with the same seed) continues the random streams where
 * they stopped instead of replaying the batches and exploration noise from the start of training.  Blocking, like
 * fbhip_get/set_step_counts. */
int fbhip_get_rng_counts(fbhip_ctx* ctx, uint32_t* host_update_count, uint32_t* host_act_count, void* stream);
int fbhip_set_rng_counts(fbhip_ctx* ctx, uint32_t update_count, uint32_t act_count, void* stream);

/* ---- the hot path --------------------------------------------------------------------------------- */
/* Enqueue the selected phases of one FBDDPGAgent.update() (fb_ddpg.py:427-520).  use_graph != 0 replays a
 * cached hipGraph of the launch sequence (captured on first use; re-captured when hparams change). */
int fbhip_update(fbhip_ctx* ctx, const fbhip_hparams* hp, const fbhip_inject* inject,
                 int32_t phase_mask, int32_t use_graph, void* stream);
/* n_steps consecutive complete updates (all phases, device-drawn batches) as ONE hipGraph launch: what the offline loop
 * (train_offline.py:101-134) does between two log lines.  The same kernels on the same operands as n_steps fbhip_update
 * calls -- the Adam / RNG counters advance on the device -- minus n_steps - 1 graph-launch gaps, with consecutive steps
 * pipelined: step t+1's SAMPLE | FB_FWD_ONLINE is captured as a second branch beside step t's actor phase, on the other
 * workspace set (so step t+1's batch is drawn before step t's actor step has finished -- from the same replay contents).
 * Results equal n_steps single updates bit for bit while no GEMM's K-slicing changes (small dims), to fp32 summation order
 * otherwise; FBHIP_UPDATE_PIPELINE=0 in the environment disables the pipelining.  Single-rank only (there is no place for
 * the gradient all-reduce inside the graph); ``hp`` is constant over the n_steps (1 <= n_steps <= 64).
 * After the call fbhip_workspace_view refers to set 0, which holds the last or the second-to-last step's intermediates. */
int fbhip_update_many(fbhip_ctx* ctx, const fbhip_hparams* hp, int32_t n_steps, void* stream);
/* How many update graphs (fbhip_update with use_graph, fbhip_update_many*, their data-parallel forms) this context has CAPTURED
 * and instantiated since fbhip_create -- a cache hit replays and does not count.  A host that launches a queue of k update()
 * calls as a fixed menu of n_steps (FBHipAgent.flush: 32 / 16 / 8 / 4 / 2 / 1; train_offline.py:116-119 is the caller) checks with
 * this that no capture happens inside its steady-state loop.  -1 for a NULL context. */
int64_t fbhip_graph_captures(const fbhip_ctx* ctx);
/* The same n-step pipelined graph with every random draw supplied: ``injects`` is an array of n_steps structs, one per step
 * (all fields a single update needs, like fbhip_update's parity mode).  For parity runs of the bench configuration -- the
 * reference's recorded draws through the pipelined multi-step graph; the graph is cached on (hp, n_steps, injects[0]), so a
 * caller that refills the same per-step device buffers replays it. */
int fbhip_update_many_injected(fbhip_ctx* ctx, const fbhip_hparams* hp, int32_t n_steps, const fbhip_inject* injects,
                               void* stream);
/* ---- data parallel on ONE node without host-issued collectives (SURVEY section 8e fallback; RCCL through torch.distributed
 * stays the default transport).  The host maps every peer's gradient buckets and barrier-flag arrays into this process (hipIpc,
 * e.g. torch.multiprocessing.reductions.reduce_tensor handles exchanged through the process group) and binds the pointers:
 *   fb_grad_ptrs[q] / actor_grad_ptrs[q]: rank q's FB / actor gradient bucket (index ``rank`` = this rank's own buffers, as bound
 *   by fbhip_bind_buffers; actor pointers ignored for dims.discrete); flag_ptrs[q]: rank q's int32[3 * 8] flag array;
 *   local_state: this rank's int32[4] (epochs + status).  Flags and state must be ZERO-initialised device memory that lives as
 * long as the context uses them.  world <= 8; world == 1 unbinds.
 * fbhip_peer_allreduce enqueues a sum-all-reduce of one bucket (which: 0 FB, 1 actor) as three kernels -- reduce-scatter,
 * all-gather, release, each behind a flag barrier across the ranks -- on ``stream``; capturable; deterministic; every rank must
 * enqueue the same sequence.  fbhip_update_many_dp is fbhip_update_many for a bound rank: n_steps complete data-parallel updates
 * (mode A: per-rank loss blocks, summed gradients, hp->grad_scale = 1 / world) with both all-reduces INSIDE the one graph.
 * Where branched graphs are usable (fbhip_branched_graphs) the steps are pipelined like fbhip_update_many's: step t+1's
 * SAMPLE | FB_FWD_ONLINE on a second capture branch beside step t's actor gradient pass, its ACTOR all-reduce and its actor step
 * (both collectives stay on the main branch, in program order: one communicator, one queue).  FBHIP_DP_PIPELINE=0, or a runtime
 * that refuses branched graphs, gives the SINGLE-QUEUE chain -- one stream, every step's phases and collectives in program order
 * (round 3's branched form replayed 2.3x slower or not depending on what else lived in the process: it was launched from a
 * normal-priority stream; branched graphs now go out from a high-priority one).  fbhip_update_many_dp_prepare captures and instantiates the same graph WITHOUT launching it: a host
 * whose ranks must agree that every one of them could build its graph (a capture of the collectives that fails on one rank would
 * leave the others waiting inside theirs) calls it first, exchanges the return codes, and only then launches.  fbhip_dp_status blocks and
 * returns the status word (0 ok, 1 = a peer did not arrive within the spin limit: results of that step are garbage, nothing hangs). */
/* The library's own RCCL transport for the same two buckets (csrc/rccl.hip): fbhip_update_many_dp then enqueues ncclAllReduce on
 * the update's stream INSIDE its capture -- one graph per rank per n_steps updates, collectives included, no process-group object
 * (and no c10d watchdog thread) on the hot path.  fbhip_rccl_load dlopens librccl (pass the path of the copy that matches the
 * process's HIP runtime -- the one torch bundles; NULL tries the loader's default); rank 0 draws fbhip_rccl_unique_id (128 bytes)
 * and the host hands the same bytes to every rank's fbhip_rccl_init (collective: every rank calls it; it also runs both buckets'
 * all-reduces once eagerly, on the zeroed gradient buffers, so that connection set-up happens outside any capture).  Takes
 * precedence over bound peers.  Works for every agent kind (the buckets are the flat gradient buffers).
 * fbhip_rccl_init(ctx, NULL, 0, rank, stream) RELEASES the context's communicator (local, not collective): rank = 0 by
 * ncclCommDestroy (orderly), rank != 0 by ncclCommAbort -- what a rank whose own init succeeded does when the host's agreement
 * says another rank's failed after the rendezvous (its collectives may be outstanding with no partner; Destroy would wait). */
int fbhip_rccl_load(const char* library_path);
int fbhip_rccl_version(void);                      /* ncclGetVersion of the loaded library, 0 if none */
int fbhip_rccl_unique_id(void* out_128_bytes);
int fbhip_rccl_init(fbhip_ctx* ctx, const void* unique_id_128_bytes, int32_t world, int32_t rank, void* stream);
int fbhip_dp_bind_peers(fbhip_ctx* ctx, int32_t world, int32_t rank, float* const* fb_grad_ptrs, float* const* actor_grad_ptrs,
                        int32_t* const* flag_ptrs, int32_t* local_state);
int fbhip_peer_allreduce(fbhip_ctx* ctx, int32_t which, void* stream);
int fbhip_update_many_dp(fbhip_ctx* ctx, const fbhip_hparams* hp, int32_t n_steps, void* stream);
int fbhip_update_many_dp_prepare(fbhip_ctx* ctx, const fbhip_hparams* hp, int32_t n_steps, void* stream);
int fbhip_dp_status(fbhip_ctx* ctx, int32_t* host_status, void* stream);
/* For hosts whose caller sits on the LEGACY default stream (torch's default stream, handle 0) while the entry points above ran on
 * `stream`: orders every later legacy-stream command after what has been enqueued on `stream` so far WITHOUT enqueuing anything on
 * the legacy stream (a wait is put on a per-device blocking helper stream; the runtime's legacy-stream rule does the rest when
 * the caller next uses the legacy stream).  A wait enqueued on the legacy stream itself stays pending while the n-step graph
 * runs and was measured to slow that graph down 1.5x (DESIGN.md section 6 "The legacy default stream").  No reference
 * counterpart (torch code is stream-ordered implicitly).  stream == NULL: no-op. */
int fbhip_order_legacy_stream_after(fbhip_ctx* ctx, void* stream);
/* The other direction: orders what is enqueued on `stream` from now on after everything the caller has enqueued on the legacy
 * default stream so far, again without a command on the legacy stream (an event recorded on a blocking helper stream, which the
 * runtime places behind the legacy stream's earlier work). */
int fbhip_order_stream_after_legacy(fbhip_ctx* ctx, void* stream);
/* The workspace holds two complete per-step sets (fbhip_update_many alternates them).  A host that pipelines steps itself
 * (data parallel: the next step's SAMPLE | FB_FWD_ONLINE under this step's actor all-reduce) selects the set the following
 * fbhip_update calls work on; phases of one step must all run on the same set.  Default 0. */
int fbhip_select_workspace_set(fbhip_ctx* ctx, int32_t which);
/* [offset, offset + count) floats of the FB gradient buffer (forward_net ++ backward_net) that are FINAL after
 * FBHIP_PHASE_FB_BWD_A: the gradients of the ForwardMap heads' hidden layers F1.0 F2.0 (weights and biases, contiguous). */
int fbhip_fb_early_grad_range(const fbhip_dims* dims, int64_t* offset, int64_t* count);
/* ---- global-batch data parallel (SURVEY section 8e "mode B"): the FB / orthonormality losses couple every row of the batch
 * with every other row (fb_ddpg.py:313-326, 344-346), so the exact loss of a batch spread over several devices needs one
 * exchange: after FB_FWD every rank exports its six [batch, pad4(z_dim)] embedding panels + discounts
 * (fbhip_embeddings_floats floats, order F1 F2 B tF1 tF2 tB discount), the host all-gathers them into
 * panels [6][global_rows][pad4(z_dim)] and discount [global_rows], binds them, and FB_BWD then evaluates rows
 * [row_offset, row_offset + batch) of the global_rows x global_rows loss: dF1 dF2 dB of the rank's own rows are complete
 * (no reduce-scatter), the normalisers are the global ones, so parameter gradients are SUMMED over ranks (grad_scale 1).
 * Metrics of that phase are the rank's share (pairwise terms, q_loss) -- sum them over ranks.  global_rows = 0 unbinds.  batch
 * and row_offset must be multiples of 32.  q_loss uses the covariance of the gathered B rows. */
size_t fbhip_embeddings_floats(const fbhip_dims* dims);
int fbhip_export_embeddings(fbhip_ctx* ctx, float* out, void* stream);
int fbhip_bind_global_batch(fbhip_ctx* ctx, const float* panels, const float* discount, int32_t global_rows,
                            int32_t row_offset);
/* The metrics of the LAST update that was enqueued with hp->want_metrics, as soon as they are final -- which is before that
 * update has finished: the step itself writes them into pinned host memory right after the actor loss (the last metric of
 * fb_ddpg.py:356-377, 413-418; for an agent without an actor: after the FB metrics) and then a sequence number; this call spins
 * on the number (no copy command, no stream synchronise) and returns while the update's remaining kernels (the actor's
 * backward pass and optimiser step) still run -- the caller's next update is enqueued behind them on the same stream.  Replaces
 * copy + synchronise for a caller that reads the metric dict of every update (README.md:50: use_tb=1 use_hiplog=1).  Falls back
 * to fbhip_read_metrics on the update's stream if the number does not arrive within 5 s or pinned memory was refused. */
int fbhip_wait_metrics(fbhip_ctx* ctx, float* host_out /* FBHIP_NUM_METRICS floats */);
/* Blocking: copies the FBHIP_NUM_METRICS device floats to host_out after the stream drains. */
int fbhip_read_metrics(fbhip_ctx* ctx, float* host_out, void* stream);
/* Named views into the workspace for tests / host code ("z", "F1", "dF1", "obs", ...). */
int fbhip_workspace_view(fbhip_ctx* ctx, const char* name, float** ptr, int32_t* rows, int32_t* cols, int32_t* ld);

/* ---- inference entry points (fb_ddpg.py:177-289) --------------------------------------------------- */
/* mu = Actor(obs, z).mean for ``rows`` rows (rows <= batch); obs [rows,o] ld=ld_obs, z [rows,d] ld=ld_z,
 * action_out [rows,a] ld=ld_out.  If noise != NULL: action = TruncatedNormal(mu, stddev).sample() with
 * noise [rows,a] contiguous (clip < 0 => no clip; utils.py:176-185). */
int fbhip_actor_forward(fbhip_ctx* ctx, const float* obs, int32_t ld_obs, const float* z, int32_t ld_z,
                        int32_t rows, const float* noise, float stddev, float clip,
                        float* action_out, int32_t ld_out, void* stream);
/* B(goal): which = 0 online backward_net, 1 backward_target_net (fb_modules.py:223-230). out [rows,d]. */
int fbhip_backward_map(fbhip_ctx* ctx, int32_t which, const float* goal, int32_t ld_goal, int32_t rows,
                       float* out, int32_t ld_out, void* stream);
/* F1,F2 = ForwardMap(obs, z, action): which = 0 online, 1 target (fb_modules.py:186-199). */
int fbhip_forward_map(fbhip_ctx* ctx, int32_t which, const float* obs, int32_t ld_obs, const float* z, int32_t ld_z,
                      const float* action, int32_t ld_act, int32_t rows,
                      float* f1_out, float* f2_out, int32_t ld_out, void* stream);
/* DiscreteFBAgent (dims.discrete): Q_i[a] = F_i(obs, z)[:, a] . z, action = argmax_a min(Q_1, Q_2)  -- ``act`` without
 * exploration (discrete_fb.py:263-268) -- for ``rows`` device rows; which = 0 online forward_net, 1 forward_target_net.
 * Optional outputs (NULL to skip): action_out int32 [rows]; next_q_out [rows] and f1_out / f2_out [rows, d] ld_out = the value
 * and the embeddings update_fb's target side selects (greedy column, or the softmax mix with ``boltzmann``; :289-303). */
int fbhip_discrete_act(fbhip_ctx* ctx, int32_t which, const float* obs, int32_t ld_obs, const float* z, int32_t ld_z,
                       int32_t rows, int32_t* action_out, float* next_q_out, float* f1_out, float* f2_out, int32_t ld_out,
                       void* stream);
/* The same arg-max for ONE observation on the batch-1 fast path (like fbhip_act: host arrays in, host result out, one hipGraph
 * launch holding H2D + a GEMV chain + the selection kernel + D2H); blocks until the action index is in *host_action_out. */
int fbhip_discrete_act_host(fbhip_ctx* ctx, const float* host_obs, const float* host_z, int32_t* host_action_out, void* stream);

/* ---- batch-1 fast path: what the online loop calls on every environment step (pretrain.py:628-632, 651-652) ----
 * HOST pointers in, HOST result out; BLOCKING (the caller needs the action to step the environment).  Each call is one
 * hipGraph launch on ``stream`` (H2D of the staged inputs through pinned memory, 4 kernel launches, D2H of the result)
 * followed by a stream synchronise.
 *
 * fbhip_act: FBDDPGAgent.act (fb_ddpg.py:258-281) for one observation: mu = Actor(obs, z).mean (fb_modules.py:107-121);
 * eval_mode != 0 -> action = mu, else action = TruncatedNormal(mu, stddev).sample(clip=None) (utils.py:176-185) with
 * eps = host_noise[a] when given, otherwise drawn on the device (Philox keyed by fbhip_set_seed and an internal counter).
 * (The reference's ``step < num_expl_steps`` uniform override and ``additional_metric`` stay on the caller's side.) */
int fbhip_act(fbhip_ctx* ctx, const float* host_obs, const float* host_z, const float* host_noise, float stddev,
              int32_t eval_mode, float* host_action_out, void* stream);
/* fbhip_z_correl: FBDDPGAgent.compute_z_correl (fb_ddpg.py:283-289): <normalize(B(goal), p=1), normalize(z, p=1)> with the
 * online backward_net (the reference's ``F.normalize(z, 1)`` passes 1 as p: L1 normalisation, kept for parity). */
int fbhip_z_correl(fbhip_ctx* ctx, const float* host_goal, const float* host_z, float* host_out, void* stream);

/* ---- individually testable kernels ----------------------------------------------------------------- */
/* C[M,N] = epi( sum_k A(m,k) * B(n,k) ).  a_kcontig: A(m,k) = A[m*lda+k] else A[k*lda+m]; same for B.
 * epi: 0 none | 1 +bias[n] | 2 relu(+bias[n]) | 3 acc*(aux>0) | 4 acc*(1-aux^2).
 * colsum (nullable, [M]) receives sum_k A(m,k) (the bias gradient of a weight-gradient GEMM). */
int fbhip_gemm(const float* A, int32_t lda, int32_t a_kcontig, const float* B, int32_t ldb, int32_t b_kcontig,
               float* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
               const float* bias, const float* aux, int32_t ldaux, int32_t epi, float* colsum, void* stream);
/* Same GEMM with an explicit tile configuration (0: 2x2x1, 1: 2x1x2, 2: 1x2x2, 3: 1x1x4, 4: 4x1x1 waves along
 * M x N x K); no epilogue.  For tests and kernel benchmarking. */
int fbhip_gemm_cfg(const float* A, int32_t lda, int32_t a_kcontig, const float* B, int32_t ldb, int32_t b_kcontig,
                   float* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t cfg, void* stream);
/* The last layer of an embedding head in ONE launch (csrc/fused.hip): c = x[rows, K] . w[N, K]^T + bias with N <= 64, and, when
 * out2 is given, out2 = scale * c / max(|c|_row, 1e-12), norms[row] = |c|_row  (fb_modules.py:78, :221, :229).  K % 4 == 0, rows
 * 16-byte aligned, ldc / ldo >= pad4(N) (pad columns are written as 0), bias readable to pad4(N); FBHIP_E_INVALID otherwise (the
 * update then runs the layer through fbhip_gemm's kernel).  replicas >= 1: the same problem that many times in one grouped launch. */
int fbhip_head(const float* x, int32_t ldx, const float* w, int32_t ldw, const float* bias, float* c, int32_t ldc, float* out2,
               int32_t ldo, float* norms, float scale, int32_t rows, int32_t N, int32_t K, int32_t replicas, void* stream);
/* y = tanh(LayerNorm(x; gamma, beta, eps=1e-5)); stats[rows,2] = (mean, rstd)  (fb_modules.py:49-50) */
int fbhip_ln_tanh_fwd(const float* x, int32_t ldx, const float* gamma, const float* beta, float* y, int32_t ldy,
                      float* stats, int32_t rows, int32_t n, void* stream);
/* dx (may alias dy); dgamma/dbeta nullable; partials scratch >= ceil(rows/8)*2*n floats when they are not. */
int fbhip_ln_tanh_bwd(const float* dy, int32_t lddy, const float* y, int32_t ldy, const float* x, int32_t ldx,
                      const float* stats, const float* gamma, float* dx, int32_t lddx,
                      float* dgamma, float* dbeta, float* partials, int32_t rows, int32_t n, void* stream);
/* out = sqrt(d) * y / max(||y||, 1e-12); norms[rows] nullable (F.normalize, fb_modules.py:229) */
int fbhip_l2norm_fwd(const float* y, int32_t ldy, float* out, int32_t ldo, float* norms,
                     int32_t rows, int32_t d, void* stream);
/* dy = (sqrt(d)/||y||) (dB - yhat (yhat . dB)), yhat = y/||y||  (autograd of the line above) */
int fbhip_l2norm_bwd(const float* dB, int32_t lddb, const float* y, int32_t ldy, const float* norms, float* dy,
                     int32_t lddy, int32_t rows, int32_t d, void* stream);
/* Actor loss (fb_ddpg.py:399-406): Q = min(F1.z, F2.z) row dots, loss = -mean Q, dF_i = -z/B on the arg-min
 * (1/2 each on exact ties); writes ACTOR_LOSS, Q, ACTOR_LOGPROB, Q1_SUCCESS into metrics.  scratch: >= 3*ceil(rows/4) floats. */
int fbhip_actor_loss(const float* F1, const float* F2, int32_t ldf, const float* z, int32_t ldz, const float* mu,
                     int32_t ldmu, const float* action, int32_t lda, float stddev, float* dF1, float* dF2, float* metrics,
                     float* scratch, int32_t rows, int32_t d, int32_t a, void* stream);
/* The two a-wide seams of the actor as stand-alone launches (tests; the update calls the same launchers): (1) premu = P W4^T + b4,
 * mu = tanh(premu), action = TruncatedNormal sample (utils.py:176-185; noise NULL: action = mu) and -- base != NULL -- the first layer
 * of the trunk that consumes the action: t1 = tanh(LayerNorm(base + W1[:, action columns] action)), with stats also the full
 * pre-activation over base and (mean, rstd) (fb_modules.py:112-126, :190); (2) d action = LayerNormTanhBackward(dt1) W1[:, action
 * columns], d premu = d action (1 - mu^2), d p = (d premu W4) relu'(P) (the reverse of the same lines).  premu / mu / action / d premu
 * share one leading dimension; [rows, H] operands share ldt.  Row kernels or 16-row MFMA tiles by FBHIP_HEAD_TILES / rows. */
int fbhip_policy_head(const float* P, int32_t ldp, const float* W4, int32_t ldw4, const float* b4, const float* noise, float stddev,
                      float clip, float* premu, float* mu, float* action, int32_t ld_out, const float* base, int32_t ldb,
                      const float* W1a, int32_t ldw1, const float* gamma, const float* beta, float* t1, int32_t ldt1, float* stats,
                      int32_t rows, int32_t H, int32_t a, void* stream);
int fbhip_actor_head_bwd(const float* dt1, int32_t ldt, const float* lnY, const float* lnX, const float* lnStats, const float* lnGamma,
                         const float* W1a, int32_t ldw1, const float* mu, int32_t ldmu, const float* W4, int32_t ldw4, const float* P,
                         float* dpremu, int32_t ldd, float* dp, int32_t rows, int32_t H, int32_t a, void* stream);
/* Pairwise FB + orthonormality loss and its gradients (fb_ddpg.py:313-348).
 * All inputs [B,d] with leading dim ld; discount [B].  Outputs dF1,dF2,dB [B,d] (ld), scalars -> metrics
 * (device float[FBHIP_NUM_METRICS]; writes TARGET_M, M1, FB_LOSS, FB_DIAG, FB_OFFDIAG, ORTH_*).
 * scratch: >= fbhip_pairwise_scratch_floats(B, d) floats. */
size_t fbhip_pairwise_scratch_floats(int32_t B, int32_t d);
int fbhip_pairwise_fb(const float* F1, const float* F2, const float* Bm, const float* tF1, const float* tF2,
                      const float* tB, const float* discount, int32_t B, int32_t d, int32_t ld, float ortho_coef,
                      float* dF1, float* dF2, float* dB, float* metrics, float* scratch, void* stream);
/* Rows [row_offset, row_offset + rows) of the same loss on the B-row panels (the global-batch data-parallel schedule): dF1 dF2
 * dB are [rows, d] (ld) and complete for those rows; metrics receive this block's SHARE of the scalars (B-row normalisers),
 * so the shares of all blocks add up to the values of fbhip_pairwise_fb.  row_offset and rows: multiples of 32 unless rows == B.
 * scratch: >= fbhip_pairwise_scratch_floats(rows, d) floats. */
int fbhip_pairwise_fb_block(const float* F1, const float* F2, const float* Bm, const float* tF1, const float* tF2,
                            const float* tB, const float* discount, int32_t B, int32_t d, int32_t ld, float ortho_coef,
                            int32_t row_offset, int32_t rows, float* dF1, float* dF2, float* dB, float* metrics,
                            float* scratch, void* stream);
/* Fused Adam (+ optional target EMA) over a flat segment: torch.optim.Adam defaults (betas .9/.999, eps 1e-8),
 * t = 1-based step count; target nullable (utils.py:66-69 fused when given). */
int fbhip_adam_ema(float* params, const float* grads, float* m, float* v, float* target, int64_t numel,
                   float lr, int32_t t, float grad_scale, float tau, void* stream);
/* out[d,d] = inverse(scale * A[d,d]), 1 <= d <= 128: Gauss-Jordan with partial pivoting in fp64 on one workgroup -- torch.inverse of
 * the q_loss covariance (fb_ddpg.py:334-335) and the pinv of SFAgent's z-mix covariance (sf.py:731-732, full rank). */
int fbhip_inverse(const float* A, int32_t lda, int32_t d, float scale, float* out, int32_t ldo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FBHIP_H */
