"""world_size-2 gloo test (CPU) of the data-parallel schedule controllable_agent_amd.distributed.dp_update:
two ranks, each with its own replay shard / micro-batch, must end on exactly the state of ONE process that is fed
both micro-batches and averages their gradients (mode A).  The compute engine behind the
schedule is the oracle (the HIP engine needs a GPU); the schedule, bucket plumbing and sharding are the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from controllable_agent_amd import distributed as D
from controllable_agent_amd.replay import DeviceReplayBuffer
from oracle import fb_oracle as fo

CFG = dict(obs_dim=5, action_dim=3, goal_dim=5, z_dim=8, hidden_dim=32, feature_dim=16, backward_hidden_dim=18,
           batch_size=16, lr=1e-3)
N_EPS, T, WORLD, STEPS = 8, 12, 2, 3


def _setup(seed=7):
    cfg = fo.OracleConfig(**CFG)
    rng = np.random.default_rng(seed)
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    storage, lengths = fo.synthetic_storage(rng, N_EPS, T, cfg.obs_dim, cfg.action_dim)
    return cfg, nets, storage, lengths


def _flat(grads):
    keys = sorted(grads)
    return keys, torch.cat([grads[k].reshape(-1) for k in keys])


def _unflat(keys, like, flat):
    out, o = {}, 0
    for k in keys:
        n = like[k].numel()
        out[k] = flat[o:o + n].reshape(like[k].shape)
        o += n
    return out


def _shard_batch(cfg, storage, lengths, rank, step):
    """rank's micro-batch of step ``step``: drawn from ITS shard (episodes ep % world == rank)"""
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cpu").shard(rank, WORLD)
    sh = {k: v.numpy() for k, v in rb._storage.items()}
    rng = np.random.default_rng(1000 * step + rank)
    d = fo.make_draws(rng, cfg, len(rb), rb._episodes_length)
    return fo.gather_batch(sh, d.ep_idx, d.step_idx, cfg.discount), d


def _worker(rank, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(1)
    cfg, nets, storage, lengths = _setup()
    agent = fo.OracleAgent(cfg, nets)
    assert list(D.shard_episodes(N_EPS, rank, WORLD)) == list(range(rank, N_EPS, WORLD))
    for step in range(STEPS):
        batch, draws = _shard_batch(cfg, storage, lengths, rank, step)
        agent.dp_begin(batch, draws)
        st = {}
        # flat gradient buckets, like FBHipAgent._fb_grads / _actor_grads
        fb_bucket = torch.zeros(sum(v.numel() for n in ("forward_net", "backward_net") for v in getattr(agent, n).values()))
        ac_bucket = torch.zeros(sum(v.numel() for v in agent.actor.values()))

        def run_phases(mask):
            if mask & D.PHASE_FB_GRAD:
                gF, gB = agent.dp_fb_grads()
                st["kF"], fF = _flat(gF)
                st["kB"], fB = _flat(gB)
                st["nF"] = fF.numel()
                fb_bucket.copy_(torch.cat([fF, fB]))
            if mask & D.PHASE_FB_STEP:
                scale = 1.0 / D.world_size()                       # grad_scale folded into the optimiser pass
                agent.dp_fb_step(_unflat(st["kF"], agent.forward_net, fb_bucket[:st["nF"]] * scale),
                                 _unflat(st["kB"], agent.backward_net, fb_bucket[st["nF"]:] * scale))
            if mask & D.PHASE_ACTOR_GRAD:
                st["kA"], fA = _flat(agent.dp_actor_grads())
                ac_bucket.copy_(fA)
            if mask & D.PHASE_ACTOR_STEP:
                agent.dp_actor_step(_unflat(st["kA"], agent.actor, ac_bucket / D.world_size()))

        D.dp_update(run_phases, fb_bucket, ac_bucket)
    out_q.put((rank, {k: v.copy() for k, v in agent.state_tensors().items()}))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_dp_schedule_equals_gradient_averaging_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: both micro-batches, averaged gradients
    torch.set_num_threads(1)
    cfg, nets, storage, lengths = _setup()
    ref = fo.OracleAgent(cfg, nets)
    twins = [fo.OracleAgent(cfg, nets) for _ in range(WORLD)]             # per-micro-batch gradient evaluators
    for step in range(STEPS):
        gFs, gBs, gAs = [], [], []
        for r, tw in enumerate(twins):
            for n in ("actor", "forward_net", "backward_net", "forward_target_net", "backward_target_net"):
                for k, v in getattr(ref, n).items():
                    getattr(tw, n)[k].copy_(v)
            tw.dp_begin(*_shard_batch(cfg, storage, lengths, r, step))
            gF, gB = tw.dp_fb_grads()
            gFs.append(gF), gBs.append(gB)
        avg = lambda gs: {k: sum(g[k] for g in gs) / WORLD for k in gs[0]}
        ref.dp_fb_step(avg(gFs), avg(gBs))
        for tw in twins:
            for n in ("forward_net", "backward_net"):
                for k, v in getattr(ref, n).items():
                    getattr(tw, n)[k].copy_(v)
            gAs.append(tw.dp_actor_grads())
        ref.dp_actor_step(avg(gAs))
    want = ref.state_tensors()
    for r in range(WORLD):
        for k, v in want.items():
            np.testing.assert_allclose(results[r][k], v, rtol=1e-5, atol=1e-7, err_msg=f"rank {r} {k}")
    for k in want:                                                        # replicas stay bit-identical
        np.testing.assert_array_equal(results[0][k], results[1][k], err_msg=k)


def test_phase_split_oracle_equals_monolithic_update():
    """the oracle's phase-split statement == its monolithic update() (which is pinned to the reference)"""
    cfg, nets, storage, lengths = _setup()
    a, b = fo.OracleAgent(cfg, nets), fo.OracleAgent(cfg, nets)
    rng = np.random.default_rng(3)
    for _ in range(3):
        d = fo.make_draws(rng, cfg, N_EPS, lengths)
        batch = fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount)
        a.update(batch, d)
        b.dp_begin(batch, d)
        b.dp_fb_step(*b.dp_fb_grads())
        b.dp_actor_step(b.dp_actor_grads())
    sa, sb = a.state_tensors(), b.state_tensors()
    for k in sa:
        np.testing.assert_allclose(sb[k], sa[k], rtol=1e-6, atol=1e-8, err_msg=k)


def test_dp_update_many_call_order_and_workspace_sets():
    """Host logic of the pipelined schedule (no process group: the all-reduces are skipped): every step's phases run on ONE
    workspace set, consecutive steps alternate sets, the next head is issued between ACTOR_GRAD and ACTOR_STEP, and set 0 is
    restored at the end."""
    calls, cur = [], [0]
    D.dp_update_many(lambda mask: calls.append((cur[0], mask)), lambda w: cur.__setitem__(0, w), torch.zeros(1), torch.zeros(1), 3)
    head = D.PHASE_SAMPLE | D.PHASE_FB_FWD_ONLINE
    mid = D.PHASE_FB_FWD_TARGET | D.PHASE_FB_BWD | D.PHASE_ACTOR_FWD
    grad = D.PHASE_FB_STEP | D.PHASE_ACTOR_GRAD
    assert calls == [(0, head),
                     (0, mid), (0, grad), (1, head), (0, D.PHASE_ACTOR_STEP),
                     (1, mid), (1, grad), (0, head), (1, D.PHASE_ACTOR_STEP),
                     (0, mid), (0, grad), (0, D.PHASE_ACTOR_STEP)]
    assert cur[0] == 0
    # every phase bit of an update is issued exactly once per step
    per_step = [0, 0, 0]
    step_of = [0, 0, 0, 1, 0, 1, 1, 2, 1, 2, 2, 2]
    for (_, m), st in zip(calls, step_of):
        assert per_step[st] & m == 0
        per_step[st] |= m
    assert per_step == [D.PHASE_ALL] * 3


@pytest.mark.parametrize("plan,expect", [
    ("", [("rccl", "ok")]),
    ("rccl:*:1:crash", [("rccl", "failed"), ("rccl", "failed"), ("c10d", "ok")]),
    ("rccl:*:1:hang,c10d:*:0:crash", [("rccl", "failed"), ("rccl", "failed"), ("c10d", "failed"), ("peer", "ok")]),
])
def test_bench_supervisor_walks_its_plan_of_transports(plan, expect, tmp_path):
    """bench.py::supervise_ranks without a GPU: two supervisors under torch.distributed.run (gloo), the real ranks replaced by
    tests/fake_bench_child.py.  A child that exits non-zero or stops writing its heartbeat fails the attempt on BOTH ranks; the
    plan goes library RCCL (pipelined graph, then FBHIP_DP_PIPELINE=0: the single-queue chain) -> torch.distributed schedule -> peer
    kernels; rank 0
    prints exactly one JSON line, the finishing child's, with the attempt history added."""
    import json, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "ROC_CPU_WAIT_FOR_SIGNAL",
                                                            "FBHIP_BENCH_CHILD")}
    env.update(FBHIP_BENCH_CHILD_CMD=json.dumps([sys.executable, str(root / "tests" / "fake_bench_child.py")]), FAKE_CHILD_PLAN=plan)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "4", "--stall-timeout", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(root), env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    att = res["data_parallel"]["attempts"]
    got = [(a["transport"], "ok" if all(r["outcome"] == "ok" for r in a["ranks"]) else "failed") for a in att]
    assert got == expect, att
    assert [a.get("env") for a in att if a["transport"] == "rccl"][1:] == [{"FBHIP_DP_PIPELINE": "0"}] * (len([a for a in att if a["transport"] == "rccl"]) - 1)
    assert res["data_parallel"]["transport"] == expect[-1][0] and "some library banner" not in out.stdout


def _worker_agreement(rank, port, out_q):
    """two ranks, the library transport refused on rank 1 only (bind raises there) / the graph build failing on rank 0 only"""
    import types
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FBHIP_DP_ALLREDUCE="rccl")
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from controllable_agent_amd import agent as A, rccl

    def stub():
        me = types.SimpleNamespace(_device="cpu", _ctx=None, _replay_token=("rb", 0))
        me._world = lambda: WORLD
        me._all_ranks_ok = lambda ok: A.FBHipAgent._all_ranks_ok(me, ok)
        me._on_update_stream = lambda fn: fn()
        return me

    # (1) bind: a local refusal becomes everybody's decision
    def bind(agent):
        if rank == 1:
            raise RuntimeError("no librccl on this rank")
        agent._rccl_bound = True
    rccl.bind = bind
    a = stub()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ready = A.FBHipAgent._rccl_ready(a)
    # (2) graph preparation: rank 0's capture fails, rank 1's succeeds -- nobody launches
    launched = []

    class Lib:
        @staticmethod
        def fbhip_update_many_dp_prepare(ctx, hp, n, s):
            return 3 if rank == 0 else 0

        @staticmethod
        def fbhip_update_many_dp(ctx, hp, n, s):
            launched.append(n)
            return 0
    A._lib.load = lambda: Lib
    A._lib.last_error = lambda ctx=None: "capture refused"
    A.stream_ptr = lambda: 0
    b = stub()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ran = A.FBHipAgent._rccl_run(b, A.HParams(), 70)
    out_q.put((rank, ready, bool(getattr(a, "_rccl_failed", False)), a._dp_transport, ran, bool(getattr(b, "_rccl_failed", False)), launched))
    dist.barrier()
    dist.destroy_process_group()


def test_transport_fallback_is_agreed_by_all_ranks():
    """the demotion from the library's RCCL transport to the torch.distributed schedule must be collective -- a rank
    that falls back alone leaves the others inside a captured ncclAllReduce it never joins.  FBHipAgent._rccl_ready and _rccl_run
    MAX-reduce their local failure flag over the default process group: with the bind refused on rank 1 only, and with the graph
    build failing on rank 0 only, BOTH ranks end on the fallback and nothing was launched."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_agreement, args=(r, _free_port_shared(), q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ready, failed, transport, ran, failed2, launched in got:
        assert ready is False and failed is True and transport.startswith("c10d (library RCCL transport refused"), got
        assert ran is False and failed2 is True and launched == [], got
    assert "no librccl on this rank" in got[1][3] and "another rank" in got[0][3]


_SHARED_PORT = []


def _free_port_shared():
    if not _SHARED_PORT:
        _SHARED_PORT.append(_free_port())
    return _SHARED_PORT[0]
