"""Two REAL ranks of the HIP data-parallel path on one GPU (both on cuda:0, backend gloo -- RCCL refuses two ranks on
one device, gloo all-reduces CUDA tensors fine): every rank runs FBHipAgent.update on its own replay shard through
controllable_agent_amd.distributed.dp_update (3 hipGraphs + 2 gradient all-reduces per step).  Must equal ONE process fed
both micro-batches with gradient averaging (the oracle, as in tests/test_distributed_cpu.py) and leave the replicas
bit-identical.  This is the rehearsal of the 8-GPU RCCL run the driver does at round end."""
import os

import numpy as np
import pytest
import torch

from oracle import fb_oracle as fo
from tests import helpers as H
from tests import test_distributed_cpu as T

pytestmark = pytest.mark.gpu


def _worker(rank, port, out_q):
    import torch.distributed as dist
    from controllable_agent_amd.replay import DeviceReplayBuffer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=T.WORLD)
    cfg, nets, storage, lengths = T._setup()
    agent = H.make_hip_agent(cfg, nets)
    assert agent._world() == T.WORLD and agent._rank() == rank
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda").shard(rank, T.WORLD)
    for step in range(T.STEPS):
        rng = np.random.default_rng(1000 * step + rank)                 # == tests/test_distributed_cpu.py::_shard_batch
        d = fo.make_draws(rng, cfg, len(rb), rb._episodes_length)
        agent.update_injected(rb, step, H.draws_dict(d), use_graph=True)
    torch.cuda.synchronize()
    out_q.put((rank, H.get_agent_state(agent), agent.step_counts()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_gradient_averaging():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(T.WORLD)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(T.WORLD)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    results = {r: st for r, st, _ in got}
    assert all(cnt == (T.STEPS, T.STEPS) for _, _, cnt in got)
    # single process (oracle): both micro-batches, averaged gradients -- same construction as the CPU test
    torch.set_num_threads(1)
    cfg, nets, storage, lengths = T._setup()
    ref = fo.OracleAgent(cfg, nets)
    twins = [fo.OracleAgent(cfg, nets) for _ in range(T.WORLD)]
    for step in range(T.STEPS):
        gFs, gBs, gAs = [], [], []
        for r, tw in enumerate(twins):
            for n in H.NETS5:
                for k, v in getattr(ref, n).items():
                    getattr(tw, n)[k].copy_(v)
            tw.dp_begin(*T._shard_batch(cfg, storage, lengths, r, step))
            gF, gB = tw.dp_fb_grads()
            gFs.append(gF), gBs.append(gB)
        avg = lambda gs: {k: sum(g[k] for g in gs) / T.WORLD for k in gs[0]}
        ref.dp_fb_step(avg(gFs), avg(gBs))
        for tw in twins:
            for n in ("forward_net", "backward_net"):
                for k, v in getattr(ref, n).items():
                    getattr(tw, n)[k].copy_(v)
            gAs.append(tw.dp_actor_grads())
        ref.dp_actor_step(avg(gAs))
    want = ref.state_tensors()
    for k in results[0]:                                                  # replicas stay bit-identical
        np.testing.assert_array_equal(results[0][k], results[1][k], err_msg=k)
    for k, v in want.items():
        if k.startswith("adam_"):
            assert H.rel_err(results[0][k], v) < 2e-4, k
        else:
            np.testing.assert_allclose(results[0][k], v, rtol=0, atol=3e-6, err_msg=k)


# ------------------------------------------------------------------------------------------ mode B: exact global batch
CFG_B = dict(T.CFG, batch_size=32)


def _setup_b(q_loss=False):
    cfg = fo.OracleConfig(**CFG_B, q_loss=q_loss)
    rng = np.random.default_rng(11)
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    storage, lengths = fo.synthetic_storage(rng, T.N_EPS, T.T, cfg.obs_dim, cfg.action_dim)
    return cfg, nets, storage, lengths


def _worker_global(rank, port, out_q, q_loss):
    import torch.distributed as dist
    from controllable_agent_amd.agent import FBHipAgent
    from controllable_agent_amd.replay import DeviceReplayBuffer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=T.WORLD)
    cfg, nets, storage, lengths = _setup_b(q_loss)
    agent = FBHipAgent(**H.agent_kwargs(cfg, dp_global_batch=True))
    agent.load_nets({n: dict(p) for n, p in nets.items()})
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda").shard(rank, T.WORLD)
    metrics = []
    for step in range(T.STEPS):
        d = fo.make_draws(np.random.default_rng(1000 * step + rank), cfg, len(rb), rb._episodes_length)
        metrics.append(agent.update_injected(rb, step, H.draws_dict(d), use_graph=True))
    torch.cuda.synchronize()
    out_q.put((rank, H.get_agent_state(agent), metrics))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("q_loss", [False, True])
def test_two_ranks_global_batch_equal_one_device_on_the_concatenated_batch(q_loss):
    """dp_global_batch=True (SURVEY 8e mode B): two ranks x 32 rows == ONE oracle update on the 64-row batch made of both
    ranks' rows (block-diagonal permutation for the z-mix), parameters, Adam state and metrics; replicas bit-identical.
    With q_loss the covariance / inverse come from the gathered B rows."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=_worker_global, args=(r, port, q, q_loss)) for r in range(T.WORLD)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(T.WORLD)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    results = {r: st for r, st, _ in got}
    metrics = {r: m for r, _, m in got}
    torch.set_num_threads(1)
    cfg, nets, storage, lengths = _setup_b(q_loss)
    ref = fo.OracleAgent(cfg, nets)
    from controllable_agent_amd.replay import DeviceReplayBuffer
    for step in range(T.STEPS):
        batches, draws = [], []
        for r in range(T.WORLD):
            rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cpu").shard(r, T.WORLD)
            sh = {k: v.numpy() for k, v in rb._storage.items()}
            d = fo.make_draws(np.random.default_rng(1000 * step + r), cfg, len(rb), rb._episodes_length)
            batches.append(fo.gather_batch(sh, d.ep_idx, d.step_idx, cfg.discount))
            draws.append(d)
        B = cfg.batch_size
        cat = lambda f: np.concatenate([getattr(d, f) for d in draws])
        both = fo.Draws(ep_idx=cat("ep_idx"), step_idx=cat("step_idx"), z_gauss=cat("z_gauss"),
                        perm=np.concatenate([d.perm + r * B for r, d in enumerate(draws)]), mix_uniform=cat("mix_uniform"),
                        eps_next=cat("eps_next"), eps_actor=cat("eps_actor"))
        batch = {k: np.concatenate([b[k] for b in batches]) for k in batches[0] if batches[0][k] is not None}
        m = ref.update(batch, both)
        for k in ("fb_loss", "fb_offdiag", "fb_diag", "orth_loss", "orth_loss_offdiag", "orth_linf", "orth_l2", "M1", "target_M",
                  "F1", "B", "B_norm", "z_norm", "actor_loss", "q") + (("q_loss",) if q_loss else ()):
            for r in range(T.WORLD):
                assert metrics[r][step][k] == pytest.approx(m[k], rel=5e-5, abs=2e-6), (step, r, k)
    want = ref.state_tensors()
    for k in results[0]:
        np.testing.assert_array_equal(results[0][k], results[1][k], err_msg=k)
    for k, v in want.items():
        if k.startswith("adam_"):
            assert H.rel_err(results[0][k], v) < 2e-4, k
        else:
            np.testing.assert_allclose(results[0][k], v, rtol=0, atol=3e-6, err_msg=k)


def test_global_batch_schedule_on_one_rank_is_the_plain_update():
    """world 1: the mode-B schedule (export -> bind -> block pairwise over all rows) lands bit-exactly on the default path."""
    cfg, nets, storage, lengths = T._setup()
    from controllable_agent_amd.agent import FBHipAgent
    from controllable_agent_amd.replay import DeviceReplayBuffer
    states = []
    for flag in (False, True):
        agent = FBHipAgent(**H.agent_kwargs(cfg, dp_global_batch=flag))
        agent.load_nets({n: dict(p) for n, p in nets.items()})
        rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda")
        ms = []
        for step in range(3):
            d = fo.make_draws(np.random.default_rng(77 + step), cfg, len(rb), rb._episodes_length)
            ms.append(agent.update_injected(rb, step, H.draws_dict(d), use_graph=True))
        states.append((H.get_agent_state(agent), ms))
    for k in states[0][0]:
        np.testing.assert_array_equal(states[0][0][k], states[1][0][k], err_msg=k)
    for a, b in zip(states[0][1], states[1][1]):
        for k in a:
            assert a[k] == pytest.approx(b[k], rel=1e-6, abs=1e-7), k


def test_global_batch_updates_are_never_queued():
    """Mode B runs an embedding exchange between the phases of EVERY update: such agents launch each metrics-off update() call at once
    (agent.py "deferred batching" queues only what fbhip_update_many can run), world 1 included."""
    cfg, nets, storage, lengths = T._setup()
    from controllable_agent_amd.agent import FBHipAgent
    from controllable_agent_amd.replay import DeviceReplayBuffer
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda")
    for flag, queued in ((True, 0), (False, 2)):              # (mode A: the first call of a run goes out at once, the rest queue)
        agent = FBHipAgent(**H.agent_kwargs(cfg, metrics=False, dp_global_batch=flag))
        agent.load_nets({n: dict(p) for n, p in nets.items()})
        for step in range(3):
            assert agent.update(rb, step) == {}
        p = agent.__dict__.get("_pending")
        assert (0 if p is None else p[3]) == queued
        assert agent.step_counts() == (3, 3)


# ------------------------------------------------------------------- pipelined data-parallel steps (dp_update_many)
def _worker_many(rank, port, out_q):
    import torch.distributed as dist
    from controllable_agent_amd.agent import FBHipAgent
    from controllable_agent_amd.replay import DeviceReplayBuffer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=T.WORLD)
    cfg, nets, storage, lengths = T._setup()
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda").shard(rank, T.WORLD)
    states = []
    for many in (False, True):
        torch.manual_seed(5)                                   # the agent's Philox key
        agent = FBHipAgent(**H.agent_kwargs(cfg))
        agent.load_nets({n: dict(p) for n, p in nets.items()})
        if many:
            m = agent.update_many(rb, 0, 5)
        else:
            for step in range(5):
                m = agent.update(rb, step)
        torch.cuda.synchronize()
        states.append((H.get_agent_state(agent), m, agent.step_counts()))
    out_q.put((rank, states))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_dp_steps_equal_single_dp_updates():
    """world 2: update_many (next step's sampling + online forward launched under the actor all-reduce, twin workspace
    sets) is bit-identical to the same number of plain data-parallel updates, on every rank, and replicas stay identical."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=_worker_many, args=(r, port, q)) for r in range(T.WORLD)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(T.WORLD))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(T.WORLD):
        (single, m1, c1), (many, m2, c2) = got[r]
        assert c1 == c2 == (5, 5)
        for k in single:
            np.testing.assert_array_equal(single[k], many[k], err_msg=f"rank {r} {k}")
        for k in m1:
            assert m1[k] == m2[k], (r, k)
    for k in got[0][1][0]:
        np.testing.assert_array_equal(got[0][1][0][k], got[1][1][0][k], err_msg=k)


# ------------------------------------------------------------------------------------------ replica consistency
def _worker_replicas(rank, port, out_q):
    """ranks construct under DIFFERENT seeds (the common seed + rank convention): the first data-parallel update must refuse;
    after sync_from_rank0 the same update runs and the replicas end bit-identical"""
    import torch.distributed as dist
    from controllable_agent_amd.agent import FBHipAgent
    from controllable_agent_amd.replay import DeviceReplayBuffer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=T.WORLD)
    cfg, nets, storage, lengths = T._setup()
    torch.manual_seed(100 + rank)
    agent = FBHipAgent(**H.agent_kwargs(cfg))                              # reference-style init: differs per rank
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda").shard(rank, T.WORLD)
    refused = False
    try:
        agent.update(rb, 0)
    except RuntimeError as e:
        refused = "replicas differ" in str(e)
    before = H.get_agent_state(agent)
    agent.sync_from_rank0()
    after = H.get_agent_state(agent)
    agent.update(rb, 0)
    agent.update_many(rb, 1, 3)
    torch.cuda.synchronize()
    out_q.put((rank, refused, before["actor/policy.0.weight"], after["actor/policy.0.weight"], H.get_agent_state(agent),
               agent.step_counts()))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_built_under_different_seeds_are_refused_then_synced():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=_worker_replicas, args=(r, port, q)) for r in range(T.WORLD)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(T.WORLD)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(g[1] for g in got), "the first data-parallel update must refuse replicas that differ"
    assert not np.array_equal(got[0][2], got[1][2])                      # they really started different
    np.testing.assert_array_equal(got[0][3], got[1][3])                  # rank 0's weights everywhere after the sync
    np.testing.assert_array_equal(got[0][3], got[0][2])
    for k in got[0][4]:                                                   # ... and they stay identical through training
        np.testing.assert_array_equal(got[0][4][k], got[1][4][k], err_msg=k)
    assert got[0][5] == got[1][5] == (4, 4)


# ------------------------------------------------------------------------------------------ configs[3] rehearsal at walker dims
def test_bench_two_rank_rehearsal_at_walker_dims_keeps_replicas_identical():
    """configs[3]'s code path at ITS per-GPU size (walker dims, batch 1024 per rank) with two ranks: ``bench.py --gpus 2
    --rehearse-on-one-gpu`` under torch.distributed.run -- the pipelined data-parallel schedule (dp_update_many: early FB
    bucket, next head on a side stream under the actor all-reduce), 64 timed steps + warm-up.  Both ranks must finish and
    report bit-identical parameter / target / Adam checksums.  (All ranks share this box's one GPU over gloo: it is a
    correctness rehearsal of the RCCL run, not a scaling measurement.)"""
    import json, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    port = T._free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "64",
           "--warmup", "8", "--repeats", "1", "--episodes", "400", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root),
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 2048
    assert res["replicas"]["identical"] is True and res["replicas"]["ranks"] == 2
    assert res["replicas"]["adam_steps"][0] >= 64 + 8 and res["value"] > 0 and "REHEARSAL" in res["data"]


# ------------------------------------------------------------------------------------------ peer-access all-reduce (in-graph)
def test_bench_launches_its_own_ranks_when_not_under_torchrun():
    """``python bench.py --gpus 2 --rehearse-on-one-gpu`` -- the plain form the driver uses for N = 1 -- with WORLD_SIZE unset: the
    script re-executes itself under torch.distributed.run (it used to die on an assertion), both ranks finish
    with identical replicas, rank 0 prints the one JSON line and it names the transport that carried the gradients (two ranks on
    one device: the library's RCCL transport refuses BEFORE any communicator call and the torch.distributed schedule takes over)."""
    import json, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "32", "--warmup", "8",
           "--repeats", "1", "--episodes", "400", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(root), env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["replicas"]["identical"] is True and res["replicas"]["ranks"] == 2
    dp = res["data_parallel"]
    assert dp["library_rccl_refused"] is True and "one device per rank" in dp["transport"], dp
    assert len(dp["per_rank_steps_per_s_median_repeat"]) == 2


@pytest.mark.parametrize("how", ["crash", "hang"])
def test_bench_supervisor_moves_to_the_next_transport_when_a_rank_is_lost(how):
    """``bench.py --gpus N`` supervises its ranks (bench.py::supervise_ranks): every launched worker runs the real rank in a child
    process, one attempt per gradient transport.  Here rank 1 of the FIRST attempt dies (exit code 7) or stops making progress
    (sleeps; --stall-timeout 30) right after building its agent: the attempt must be given up on BOTH ranks -- rank 0's child is
    then blocked in its first collective -- and the second transport must deliver the one JSON line, with the history in it."""
    import json, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", FBHIP_BENCH_FAIL_TRANSPORT=f"rccl:{how}")
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "32", "--warmup", "8",
           "--repeats", "1", "--episodes", "400", "--no-cpu-baseline", "--stall-timeout", "30"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=str(root), env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                    # ONE JSON line, from the attempt that finished
    res = json.loads(lines[0])
    att = res["data_parallel"]["attempts"]
    assert [a["transport"] for a in att] == ["rccl", "rccl", "c10d"], att           # (pipelined graph, single-queue chain, next transport)
    assert att[1].get("env") == {"FBHIP_DP_PIPELINE": "0"}
    first = [r["outcome"] for r in att[0]["ranks"]]
    # (rank 0's child either is killed while blocked in its first collective or notices the closed connection by itself)
    assert first[1].startswith("failed (" + ("exit code 7" if how == "crash" else "no progress")) and first[0].startswith("failed"), att
    assert [r["outcome"] for r in att[2]["ranks"]] == ["ok", "ok"]
    assert res["n_gpus"] == 2 and res["replicas"]["identical"] is True and res["value"] > 0


def _worker_peer(rank, port, out_q, mode, world=T.WORLD):
    """``mode`` "peer": FBHIP_DP_ALLREDUCE=peer -- the ranks map each other's gradient buckets (hipIpc) and every data-parallel
    step is ONE graph launch per rank with the all-reduce kernels inside (csrc/peer.hip);  "host": the default schedule with
    torch.distributed (gloo here) between the phase graphs.  Same seeds, same shards, device-drawn batches."""
    import torch.distributed as dist
    from controllable_agent_amd import peer
    from controllable_agent_amd.replay import DeviceReplayBuffer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FBHIP_DP_ALLREDUCE="peer" if mode == "peer" else "c10d",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, nets, storage, lengths = T._setup()
    torch.manual_seed(4321)                   # the agent's device RNG key is torch.initial_seed(): same batches in both modes
    agent = H.make_hip_agent(cfg, nets, metrics=False)
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda").shard(rank, world)
    agent.update(rb, 0)                       # one single step (n = 1 graph) ...
    agent.update_many(rb, 1, 5)               # ... then five pipelined ones in one launch, twice (graph replay)
    agent.update_many(rb, 6, 5)
    torch.cuda.synchronize()
    st = peer.status(agent) if mode == "peer" else 0
    # the raw primitive on a known pattern: bucket value = rank + 1 everywhere -> sum = W (W + 1) / 2 on every rank
    summed = None
    if mode == "peer":
        agent._fb_grads.fill_(float(rank + 1))
        torch.cuda.synchronize()
        dist.barrier()
        from controllable_agent_amd import _lib
        _lib.check(_lib.load().fbhip_peer_allreduce(agent._ctx, 0, _lib.stream_ptr()), agent._ctx)
        torch.cuda.synchronize()
        summed = (float(agent._fb_grads.min()), float(agent._fb_grads.max()), peer.status(agent))
    out_q.put((rank, H.get_agent_state(agent), agent.step_counts(), st, summed))
    dist.barrier()
    dist.destroy_process_group()


def _run_peer(mode, world=T.WORLD):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=_worker_peer, args=(r, port, q, mode, world)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_peer_allreduce_with_four_ranks():
    """the same kernels with world 4 (four processes sharing the GPU): chunking of the bucket over four owners, four flag slots
    per barrier -- replicas bit-identical, no barrier timeout, a known pattern summed exactly"""
    res = _run_peer("peer", world=4)
    assert all(r[2] == (11, 11) for r in res)
    for r in res[1:]:
        for k in res[0][1]:
            np.testing.assert_array_equal(res[0][1][k], r[1][k], err_msg=k)
    assert all(r[3] == 0 for r in res), "a peer barrier timed out"
    assert all(r[4] == (10.0, 10.0, 0) for r in res), [r[4] for r in res]


def test_peer_allreduce_inside_the_graph_equals_the_host_schedule():
    """two REAL ranks on one GPU: the in-graph peer all-reduce (one graph launch per rank per call) leaves the replicas
    bit-identical to each other, never times out, sums a known pattern exactly, and lands where the host-issued schedule lands
    (the two schedules group a few launches differently: fp32 summation order at most)"""
    peer_res, host_res = _run_peer("peer"), _run_peer("host")
    for res in (peer_res, host_res):
        assert res[0][2] == res[1][2] == (11, 11)
        for k in res[0][1]:
            np.testing.assert_array_equal(res[0][1][k], res[1][1][k], err_msg=k)                 # replicas identical
    assert all(r[3] == 0 for r in peer_res), "a peer barrier timed out"
    want = float(sum(range(1, T.WORLD + 1)))
    assert all(r[4] == (want, want, 0) for r in peer_res), [r[4] for r in peer_res]
    for k, v in host_res[0][1].items():
        if k.startswith("adam_"):
            assert H.rel_err(peer_res[0][1][k], v) < 2e-3, k
        else:
            np.testing.assert_allclose(peer_res[0][1][k], v, rtol=0, atol=2e-5, err_msg=k)


# ------------------------------------------------------------------------------ the schedule captured into one graph (round 2)
def _run_c10d_schedule(monkeypatch=None):
    from controllable_agent_amd.replay import DeviceReplayBuffer
    cfg, nets, storage, lengths = T._setup()
    agent = H.make_hip_agent(cfg, nets)
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda")
    for call in range(3):
        agent.update_many(rb, 6 * call, 6)
    agent.update(rb, 18)
    # another buffer object (other device pointers): whatever was captured holds the old ones and must not be replayed
    rb2 = DeviceReplayBuffer.from_arrays({k: v[::-1].copy() for k, v in storage.items()}, lengths[::-1].copy(), cfg.discount, device="cuda")
    agent.update_many(rb2, 19, 5)
    torch.cuda.synchronize()
    return H.get_agent_state(agent), agent.step_counts()


def _worker_c10d_nccl_world1(port, out_q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FBHIP_FORCE_PHASE_SPLIT="1", FBHIP_DP_ALLREDUCE="c10d")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    torch.manual_seed(4321)            # the agent's device RNG key comes from torch's seed, which differs between fresh processes
    state, counts = _run_c10d_schedule()
    out_q.put((state, counts))
    dist.barrier()
    dist.destroy_process_group()


def test_c10d_schedule_beside_a_live_nccl_group_equals_the_schedule_without_one(monkeypatch):
    """The torch.distributed schedule (FBHIP_DP_ALLREDUCE=c10d: the fallback transport) with a live RCCL process group (backend
    nccl, world 1 -- all this box can hold): its phases are issued as eager launches there (no stream capture beside c10d's
    watchdog thread, FBHipAgent._c10d_collectives_in_flight) with the real all-reduce calls in between; without a process
    group the same schedule replays one library graph per phase.  Same kernels, operands and order: bit-identical state, and the
    process with the group exits cleanly."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_c10d_nccl_world1, args=(T._free_port(), q))
    p.start()
    state_nccl, counts_nccl = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    monkeypatch.setenv("FBHIP_FORCE_PHASE_SPLIT", "1")
    monkeypatch.setenv("FBHIP_DP_ALLREDUCE", "c10d")
    torch.manual_seed(4321)
    state, counts = _run_c10d_schedule()
    assert counts == counts_nccl == (24, 24)
    for k in state:
        np.testing.assert_array_equal(state[k], state_nccl[k], err_msg=k)


def _dot_out_degrees(path):
    import collections, re as _re
    deg = collections.Counter()
    for m in _re.finditer(r'"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', open(path).read()):
        deg[m.group(1)] += 1
    return deg


def test_the_data_parallel_graph_is_pipelined_and_keeps_a_single_queue_fallback(monkeypatch, tmp_path):
    """fbhip_update_many_dp's n-step graph forks like the single-rank one (round 6): step t+1's head on a second branch beside step
    t's actor gradient pass, actor all-reduce and actor step -- both collectives stay on the main branch, in program order.  With
    FBHIP_DP_PIPELINE=0 (or on a runtime that refuses branched graphs) it is a CHAIN: no node has two successors -- the fallback
    every rank can agree on (round 3's branched form, launched from a normal-priority stream, replayed 2.3x slower or not depending
    on what else lived in the process).  Same kernels and operands either way: at these dims the same bits."""
    from controllable_agent_amd.replay import DeviceReplayBuffer
    cfg, nets, storage, lengths = T._setup()
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda")
    monkeypatch.setenv("FBHIP_GRAPH_DOT", str(tmp_path / "single.dot"))
    a1 = H.make_hip_agent(cfg, nets)
    a1.update_many(rb, 0, 4)
    torch.cuda.synchronize()
    assert max(_dot_out_degrees(tmp_path / "single.dot").values()) >= 2          # (sanity of the reader: this one forks)
    monkeypatch.setenv("FBHIP_FORCE_PHASE_SPLIT", "1")                            # the library RCCL transport at world 1
    states = {}
    for form, env in (("pipelined", None), ("chain", "0")):
        monkeypatch.setenv("FBHIP_GRAPH_DOT", str(tmp_path / f"dp_{form}.dot"))
        if env is not None:
            monkeypatch.setenv("FBHIP_DP_PIPELINE", env)
        torch.manual_seed(11)
        a2 = H.make_hip_agent(cfg, nets)
        a2.update_many(rb, 0, 4)
        torch.cuda.synchronize()
        assert "rccl-library" in a2._dp_transport
        assert a2.step_counts() == (4, 4)
        states[form] = H.get_agent_state(a2)
        deg = _dot_out_degrees(tmp_path / f"dp_{form}.dot")
        if form == "chain":
            assert deg and max(deg.values()) == 1, deg.most_common(3)
        else:
            from controllable_agent_amd import _lib
            if _lib.load().fbhip_branched_graphs(None):
                assert deg and max(deg.values()) >= 2, deg.most_common(3)
    for k in states["chain"]:
        np.testing.assert_array_equal(states["chain"][k], states["pipelined"][k], err_msg=k)


# ------------------------------------------------------------------------------------------ the SF sibling, data parallel
def _sf_inputs(name="tiny_sf_svdp_trace"):
    """(agent, replay, three steps of recorded draws, state reader) of a reference trace: an SFAgent one, or (tiny_debug_*) an
    FBDDPGAgent one with the IdentityMap backward nets"""
    from tests.test_oracle_golden import sf_trace_inputs
    from tests.test_sf_agent_gpu import make_sf_agent, _buffer, get_sf_state
    if name.startswith("tiny_sf_"):
        meta, z, cfg, nets, storage, lengths = sf_trace_inputs(name)
        agent = make_sf_agent(cfg, nets, meta["feature_learner"], meta["sf_q_loss"], meta["goal_space"])
        reader = get_sf_state
    else:
        meta = H.load_meta(name)
        cfg = H.cfg_from_meta(meta)
        z = np.load(H.GOLDEN / f"{name}.npz")
        storage, lengths = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith("storage/")}, z["lengths"]
        nets = {n: {k.split("/", 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"init/{n}/")}
                for n in ("actor", "forward_net", "backward_net")}
        agent = H.make_hip_agent(cfg, nets, meta["goal_space"])
        reader = H.get_agent_state
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)              # NOT sharded: every rank sees the same batches
    draws = [H.draws_dict(fo.Draws(**{f: z[f"draws/{s}/{f}"] for f in fo.Draws.__dataclass_fields__ if f"draws/{s}/{f}" in z.files}))
             for s in range(3)]
    return agent, rb, draws, reader


def _worker_sf(rank, port, out_q, name):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=T.WORLD)
    torch.manual_seed(4321)
    agent, rb, draws, get_sf_state = _sf_inputs(name)
    for s, d in enumerate(draws):
        agent.update_injected(rb, s, d)
    torch.cuda.synchronize()
    after3 = get_sf_state(agent)
    agent.update_many(rb, 3, 2)                                           # (falls back to data-parallel single updates; device-drawn)
    torch.cuda.synchronize()
    out_q.put((rank, after3, get_sf_state(agent), agent.step_counts()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tiny_sf_svdp_trace", "tiny_sf_mix_lap_trace", "tiny_debug_trace", "tiny_debug_future_randw_trace"])
def test_sf_agent_two_ranks_average_gradients(name):
    """(also: SFAgent's z-mix, and FBDDPGAgent with the IdentityMap backward nets of cfg.debug, whose block of the bucket stays zero)
    SFHipAgent under the data-parallel schedule (gradients | all-reduce | sf_opt + phi_opt step, actor gradient | all-reduce |
    actor step): two ranks fed IDENTICAL batches average two equal gradients -- (g + g) / 2 is exact in fp32 -- so both must land
    on the state of ONE process running the same three updates (to the fp32 summation order of the regrouped launches), and the
    replicas stay bit-identical, also through the device-drawn update_many that follows (same seed, unsharded buffer)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=_worker_sf, args=(r, port, q, name)) for r in range(T.WORLD)]
    for p in procs:
        p.start()
    got = {r: (a3, fin, cnt) for r, a3, fin, cnt in (q.get(timeout=300) for _ in range(T.WORLD))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][2] == got[1][2] == (5, 5)
    for which in (0, 1):
        for k in got[0][which]:
            np.testing.assert_array_equal(got[0][which][k], got[1][which][k], err_msg=k)
    agent, rb, draws, get_sf_state = _sf_inputs(name)
    for s, d in enumerate(draws):
        agent.update_injected(rb, s, d)
    ref = get_sf_state(agent)
    for k, v in ref.items():
        np.testing.assert_allclose(got[0][0][k], v, rtol=0, atol=3e-6, err_msg=k)


def _worker_sf_transport(rank, port, out_q, name, mode):
    """device-drawn SF updates on two ranks, gradients carried by the peer-access kernels inside the update graph (``mode`` "peer")
    or by torch.distributed between the phase launches ("host")"""
    import torch.distributed as dist
    from controllable_agent_amd import peer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FBHIP_DP_ALLREDUCE="peer" if mode == "peer" else "c10d",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=T.WORLD)
    torch.manual_seed(4321)
    agent, rb, _, get_sf_state = _sf_inputs(name)
    agent.update(rb, 0)                        # a single step (n = 1 graph) ...
    agent.update_many(rb, 1, 4)                # ... four pipelined ones in one launch, twice (graph replay)
    agent.update_many(rb, 5, 4)
    torch.cuda.synchronize()
    st = peer.status(agent) if mode == "peer" else 0
    out_q.put((rank, get_sf_state(agent), agent.step_counts(), st, getattr(agent, "_peer_bound", False)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tiny_sf_svdp_trace", "tiny_sf_icm_trace"])
def test_sf_agent_peer_allreduce_inside_the_graph_equals_the_host_schedule(name):
    """the SF sibling on the in-graph transports.  Two REAL ranks on one GPU (RCCL refuses two ranks per device,
    so the peer kernels stand in for the library's communicator: same entry point, fbhip_update_many_dp, same places in the graph):
    replicas bit-identical, no barrier timeout, and the state lands where the host-issued schedule lands (fp32 summation order of
    the regrouped launches at most)."""
    import torch.multiprocessing as mp
    res = {}
    for mode in ("peer", "host"):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = T._free_port()
        procs = [ctx.Process(target=_worker_sf_transport, args=(r, port, q, name, mode)) for r in range(T.WORLD)]
        for p in procs:
            p.start()
        res[mode] = sorted((q.get(timeout=300) for _ in range(T.WORLD)), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    for mode, got in res.items():
        assert got[0][2] == got[1][2], (mode, got[0][2], got[1][2])
        for k in got[0][1]:
            np.testing.assert_array_equal(got[0][1][k], got[1][1][k], err_msg=f"{mode} {k}")         # replicas identical
    assert res["peer"][0][2] == res["host"][0][2]
    assert all(r[3] == 0 for r in res["peer"]), "a peer barrier timed out"
    assert all(r[4] for r in res["peer"]) and not any(r[4] for r in res["host"])
    for k, v in res["host"][0][1].items():
        if "adam_" in k or k.startswith(("m/", "v/")):
            assert H.rel_err(res["peer"][0][1][k], v) < 2e-3, k
        else:
            np.testing.assert_allclose(res["peer"][0][1][k], v, rtol=0, atol=2e-5, err_msg=k)


def _run_library_rccl(monkeypatch, transport, sf=False):
    from controllable_agent_amd.replay import DeviceReplayBuffer
    if transport is None:
        monkeypatch.delenv("FBHIP_FORCE_PHASE_SPLIT", raising=False)
    else:
        monkeypatch.setenv("FBHIP_FORCE_PHASE_SPLIT", "1")
        monkeypatch.setenv("FBHIP_DP_ALLREDUCE", transport)
    torch.manual_seed(77)              # (the agent's device RNG key comes from torch's seed: the three runs must draw the same batches)
    if sf:
        agent, rb, _, get_state = _sf_inputs("tiny_sf_icm_trace")
    else:
        cfg, nets, storage, lengths = T._setup()
        agent = H.make_hip_agent(cfg, nets)
        rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda")
        get_state = H.get_agent_state
    agent.update(rb, 0)
    for call in range(3):
        agent.update_many(rb, 1 + 6 * call, 6)
    torch.cuda.synchronize()
    return get_state(agent), agent.step_counts(), getattr(agent, "_dp_transport", "")


@pytest.mark.parametrize("sf", [False, True])
def test_library_rccl_transport_on_one_rank_lands_where_the_other_schedules_land(monkeypatch, sf):
    """``fbhip_update_many_dp`` with the library's OWN communicator (dlopen'ed librccl, ncclCommInitRank from a unique id, ncclAllReduce
    of both buckets captured inside the n-step graph -- world 1 is all this box can hold): 1 + 3 x 6 updates end on the state of the
    c10d schedule and of the plain single-GPU path (same kernels and order inside each step: bit-identical at these dims), for the FB
    agent and for an SFAgent (whose update_many now pipelines under data parallelism too)."""
    s_rccl, c_rccl, t_rccl = _run_library_rccl(monkeypatch, "rccl", sf)
    assert t_rccl.startswith("rccl-library"), t_rccl
    s_c10d, c_c10d, _ = _run_library_rccl(monkeypatch, "c10d", sf)
    s_one, c_one, _ = _run_library_rccl(monkeypatch, None, sf)
    assert c_rccl == c_c10d == c_one == (19, 19)
    for k in s_one:
        np.testing.assert_array_equal(s_rccl[k], s_c10d[k], err_msg=k)
        np.testing.assert_array_equal(s_rccl[k], s_one[k], err_msg=k)


def _worker_library_rccl_beside_c10d(port, out_q):
    import torch.distributed as dist
    from controllable_agent_amd.replay import DeviceReplayBuffer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FBHIP_FORCE_PHASE_SPLIT="1", FBHIP_DP_ALLREDUCE="rccl")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    cfg, nets, storage, lengths = T._setup()
    agents = []
    for i in range(6):                                        # several agents, each with captures of its own, under a LIVE c10d group:
        torch.manual_seed(100 + i)                            # the round-2 abort scenario (watchdog poll during a capture) without any sleep
        a = H.make_hip_agent(cfg, nets)
        rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda")
        dist.all_reduce(torch.ones(8, device="cuda"))         # keeps the watchdog busy polling end events
        a.update_many(rb, 0, 6)
        a.update(rb, 6)
        a.act(np.zeros(cfg.obs_dim, np.float32), a.init_meta(), 0, eval_mode=True)
        agents.append(a)
    torch.cuda.synchronize()
    out_q.put(([a.step_counts() for a in agents], [getattr(a, "_dp_transport", "") for a in agents]))
    dist.barrier()
    dist.destroy_process_group()


def test_library_rccl_transport_beside_a_live_c10d_group_needs_no_quiescing():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_library_rccl_beside_c10d, args=(T._free_port(), q))
    p.start()
    counts, transports = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert counts == [(7, 7)] * 6 and all(t.startswith("rccl-library") for t in transports), (counts, transports)
