"""bench.py -- FB update-steps/sec of the HIP path on MI355X (BASELINE.json metric, config[1]: walker offline
replay, batch=1024, z_dim=50), with the roofline of the step and the CPU baseline timed beside it.

    python bench.py [--gpus N --steps K --warmup W]                 # N=1: plain process
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                      # N>1: one rank per GPU over RCCL

One "step" = one FBDDPGAgent.update(): on-device replay sample + FB step + actor step + target EMA, replayed
as one hipGraph; inputs (the 5000-episode synthetic replay buffer) are resident in HBM before the timed region.
N>1 is data parallel (SURVEY.md section 8e mode A): every rank owns a replay shard and a 1024-transition
minibatch, gradients are all-reduced (RCCL) twice per step -> weak scaling; value = N * K / max-over-ranks time.
For N>1 every launched worker supervises a child process that is the real rank (supervise_ranks): a crash or a stall moves all
ranks on to the next gradient transport; the line then carries data_parallel.attempts.

Environment read by this script (diagnostics, none needed for the line): ROC_CPU_WAIT_FOR_SIGNAL (default set here to 1 for N = 1,
see below), FBHIP_BENCH_LEGACY_STREAM=1 (enqueue from torch's legacy default stream), FBHIP_BENCH_WORLD1_BACKEND /
FBHIP_BENCH_EXTRA_STREAMS (one-rank probes), FBHIP_BENCH_FAIL_TRANSPORT (tests of the supervisor).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

# Runtime configuration, set before the HIP runtime is loaded (import torch).  ROC_CPU_WAIT_FOR_SIGNAL=1: ROCclr resolves a
# dependency on another hardware queue's signal by waiting for it on the host instead of parking a barrier packet on it.  The
# SINGLE-GPU n-step update graph has two to three branches, i.e. cross-queue dependencies at every fork and join: measured +3.2 %
# on the bench line (1116.5 -> 1151.9 update-steps/s, same box).  Runs with --gpus N > 1 keep the runtime's default: the setting has
# never been exercised beside RCCL's own signal waits on more than one device (the data-parallel graph is pipelined too since round
# 6 -- on ONE device with the real RCCL calls the setting is fine, 1205 update-steps/s -- and its single-queue fallback form has no
# cross-queue dependency at all).  An explicit setting in the environment wins.  Reported in config.runtime_env.
def _multi_gpu_invocation():
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return True
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            return sys.argv[i + 1].isdigit() and int(sys.argv[i + 1]) > 1
        if a.startswith("--gpus="):
            return a[7:].isdigit() and int(a[7:]) > 1
    return False


if "ROC_CPU_WAIT_FOR_SIGNAL" not in os.environ and not _multi_gpu_invocation():
    os.environ["ROC_CPU_WAIT_FOR_SIGNAL"] = "1"

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WALKER = dict(obs_dim=24, action_dim=6, goal_dim=24, z_dim=50, hidden_dim=1024, feature_dim=512,
              backward_hidden_dim=526, batch_size=1024)
# configs[2] (not the bench line: `--workload quadruped` for the record in DESIGN.md): quadruped_walk, goal space
# simplified_quadruped (g = 2), batch 2048, z_dim 100
QUADRUPED = dict(obs_dim=78, action_dim=12, goal_dim=2, z_dim=100, hidden_dim=1024, feature_dim=512,
                 backward_hidden_dim=526, batch_size=2048)
PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md chip table


def algorithmic_gflop_per_update(o, a, g, d, H, Fd, Hb, B):
    """minimal algorithm (reference math minus the weight gradients its actor step computes
    and discards); FLOP = 2 MAC, forward 1x, trained passes 3x."""
    Ff = (o + a) * H + H * Fd + (o + d) * H + H * Fd + 2 * (2 * Fd * H + H * d)
    Fa = o * H + H * Fd + (o + d) * H + H * Fd + 2 * Fd * H + H * a
    Fb = g * Hb + Hb * Hb + Hb * d
    Fp = 11 * B * d
    Fdg = 2 * (H * d + 2 * Fd * H) + H * Fd + a * H
    mac_row = 0.5 * Fb + Fa + Ff + Fb + 3 * Ff + 3 * Fb + 3 * Fa + Ff + Fdg + Fp
    return 2 * mac_row * B / 1e9


def make_replay(n_episodes, T, o, a, device, seed, goal_dim=None):
    """Synthetic buffer per BASELINE.md section 4: obs ~ N(0,1), action ~ U(-1,1), stored discount 1."""
    from controllable_agent_amd.replay import DeviceReplayBuffer
    g = torch.Generator(device=device).manual_seed(seed)
    rb = DeviceReplayBuffer(n_episodes, discount=0.99, future=0.99, device=device)
    rb._storage = {
        "observation": torch.randn((n_episodes, T + 1, o), device=device, generator=g),
        "action": torch.rand((n_episodes, T + 1, a), device=device, generator=g) * 2 - 1,
        "reward": torch.rand((n_episodes, T + 1, 1), device=device, generator=g),
        "discount": torch.ones((n_episodes, T + 1, 1), device=device),
    }
    if goal_dim is not None:
        rb._storage["goal"] = torch.randn((n_episodes, T + 1, goal_dim), device=device, generator=g)
    rb._episodes_length = np.full(n_episodes, T, np.int32)
    rb._idx, rb._full = 0, True
    rb._touch()
    return rb


def _cpu_port_rate(batch_size, threads, budget_s, max_steps, seed=1):
    """updates/s of the oracle (CPU restatement of fb_ddpg.py:427-520) at walker dims with ``threads`` torch threads"""
    from oracle import fb_oracle as fo
    cfg = fo.OracleConfig(**dict(WALKER, batch_size=batch_size))
    rng = np.random.default_rng(seed)
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    n_eps, T = 50, 1000
    storage, lengths = fo.synthetic_storage(rng, n_eps, T, cfg.obs_dim, cfg.action_dim)
    agent = fo.OracleAgent(cfg, nets)

    def one():
        d = fo.make_draws(rng, cfg, n_eps, lengths)
        agent.update(fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount), d)
    torch.set_num_threads(threads)
    one()                                              # warm-up (allocator, thread pool)
    t0, n = time.time(), 0
    while n < max_steps and (n == 0 or time.time() - t0 < budget_s):
        one()
        n += 1
    dt = time.time() - t0
    return n / dt, n, dt


def cpu_baseline(seed=1, budget_s=12.0, max_steps=120):
    """The oracle (our CPU restatement of the reference's update, pinned to it by tests/golden) timed on this box's host
    cores.  Against the imported reference on the same 8 cores of the build container the port needs 1.11x (batch 1024) to
    1.18x (batch 256) the time per update (tools/port_vs_reference_timing.py -> profiles/r02_port_vs_reference_timing.json;
    BASELINE.md section 4 hoped for +-10 %): read ``value`` as a LOWER bound of the reference's CPU rate, ~10-18 % low -- the reference's Python cannot travel to the GPU box.  Bounded sample of the same
    workload.  Headline: batch 1024 at the fastest of a few thread counts; ``also``: BASELINE.md section 4's other figures
    (configs[0] = batch 256, and the 1-thread rates), each on a 2-4 s sample."""
    # torch-CPU oversubscribes badly when given every hardware thread of a big host (75 s/update at 256 threads):
    # probe a few thread counts on the cores this process may actually use and keep the fastest
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best = None
    for nt in sorted({min(avail, n) for n in (8, 16, 32, 64)}):
        rate, _, _ = _cpu_port_rate(1024, nt, 0.0, 1, seed)
        if best is None or rate > best[1]:
            best = (nt, rate)
        if rate < 1.0 / 3.0:
            break
    rate, n, dt = _cpu_port_rate(1024, best[0], budget_s, max_steps, seed)
    also = []
    for bs, nt, bud in ((256, best[0], 4.0), (1024, 1, 3.0), (256, 1, 3.0)):
        r, k, d = _cpu_port_rate(bs, nt, bud, 60, seed)
        also.append({"batch": bs, "threads": nt, "value": r, "unit": "update-steps/s", "sample": f"{k} updates, {d:.1f} s"})
    return {"value": rate, "unit": "update-steps/s", "cores": best[0], "kind": "port",
            "sample": f"{n} updates of the same workload (walker dims, batch 1024, z_dim 50, metrics on) with "
                      f"oracle/fb_oracle.py (torch-CPU fp32 restatement of fb_ddpg.py:427-520, autograd + "
                      f"boolean-mask loss like the reference), {dt:.1f} s; host has {avail} usable hardware threads",
            "port_vs_reference": "the port takes 1.11x (batch 1024) / 1.18x (batch 256) the reference's time per update on the "
                                 "same 8 cores (profiles/r02_port_vs_reference_timing.json, build container)",
            "also": also}


def kernel_sources_sha16():
    """digest of controllable_agent_amd/csrc/*.{hip,h,inc}: which build a recorded profile belongs to"""
    import hashlib
    h = hashlib.sha256()
    src = Path(__file__).resolve().parent / "controllable_agent_amd" / "csrc"
    for f in sorted(list(src.glob("*.hip")) + list(src.glob("*.h")) + list(src.glob("*.inc"))):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def measured_traffic(workload="walker"):
    """(bytes per update, where the figure comes from) of the last COMMITTED PMC passes (profiles/traffic*.json, made by
    tools/profile_round.sh + tools/pmc_summary.py); (None, None) when no pass is on file.  ``--pmc`` measures it live instead."""
    f = Path(__file__).resolve().parent / "profiles" / ("traffic.json" if workload == "walker" else f"traffic_{workload}.json")
    try:
        d = json.loads(f.read_text())
        same = d.get("kernel_sources_sha16") == kernel_sources_sha16()
        return float(d["hbm_bytes_per_update"]), (f"profiles/{f.name} (round {d.get('round', '?')}: PMC passes of "
                                                  + ("THIS build -- the same kernel sources, csrc digest " + str(d.get("kernel_sources_sha16")) if same else
                                                     "an earlier build, not this run") + "; --pmc measures live)")
    except (OSError, KeyError, ValueError):
        return None, None


def live_traffic(workload):
    """--pmc: HBM-side bytes per update of THIS build, collected now: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE -- the TCC block
    cannot hold both, and counters are collected with --kernel-trace only, MI355X_MICROARCH.md "rocprofv3 PMC slots") of a short
    run of this script, summed over every kernel between two sampler launches (tools/pmc_summary.py's step definition), FETCH_SIZE
    doubled per the guide's gfx950 note (wide coalesced reads are tallied at half).  Returns (bytes, description) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "--pmc: rocprofv3 not on PATH"
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fbhip_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               str(Path(__file__).resolve()), "--steps", "64", "--warmup", "32", "--repeats", "1", "--no-cpu-baseline",
               "--no-single-update-probe", "--workload", workload]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=900)
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                rows += [x for x in csv.DictReader(fh) if x["Counter_Name"] == counter]
        shutil.rmtree(d, ignore_errors=True)
        if r.returncode != 0 or not rows:
            return None, f"--pmc: the {counter} pass failed (exit code {r.returncode}, {len(rows)} rows)"
        rows.sort(key=lambda x: int(x["Dispatch_Id"]))
        draws = [int(x["Dispatch_Id"]) for x in rows if "draw_kernel" in x["Kernel_Name"]]
        skip = 12 if len(draws) > 14 else 0
        lo, hi, n = draws[skip], draws[-1], len(draws) - skip - 1
        out[counter] = sum(float(x["Counter_Value"]) for x in rows if lo <= int(x["Dispatch_Id"]) < hi) * 1024.0 / max(1, n)
    total = 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]
    return total, (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build ({out['FETCH_SIZE'] / 1e6:.1f} MB raw x 2 + "
                   f"{out['WRITE_SIZE'] / 1e6:.1f} MB per update)")


def dominant_kernel_probe(stream_iters=50):
    """HIP-event timing (on the launch stream) of the step's dominant kernel shape: the stacked F1|F2 hidden layer
    GEMM  p[1024,2048] = h[1024,1024] . W3s^T  through the same fbhip gemm_kernel the update launches."""
    from controllable_agent_amd import kernels as K
    M, N, Kd = 1024, 2048, 1024
    A, B = torch.randn(M, Kd, device="cuda"), torch.randn(N, Kd, device="cuda")
    C = torch.empty(M, N, device="cuda")
    for _ in range(5):
        K.gemm(A, B, out=C)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(stream_iters):
        K.gemm(A, B, out=C)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / stream_iters
    return {"kernel": "fbhip::gemm_kernel<2,2,1,32>", "shape": [M, N, Kd], "us": us,
            "tflops": 2 * M * N * Kd / us / 1e6, "frac_of_peak": 2 * M * N * Kd / us / 1e6 / PEAK_FP32_MFMA_TFLOPS,
            "note": "the kernel's most expensive single-problem shape in the step, launched alone and timed with HIP events on its "
                    "launch stream; the same launches inside the step: profiles/*_step_timeline.txt (42-45 us).  The rocprofv3 "
                    "--stats average of this kernel NAME (profiles/*_kernel_stats.txt, ~37 us) is over all 14 grouped launches "
                    "per update, whose shapes differ (4.3-8.7 GFLOP, 7-95 us)"}


def dominant_in_step():
    """What the dominant kernel gets INSIDE the benchmarked step (its launched-alone figure above is its best single shape): sum of
    GFLOP / sum of kernel time over all its launches of an update, from the kernel trace of the same bench command
    (tools/profile_round.sh -> tools/dominant_in_step.py -> profiles/dominant_in_step.json; not measured by this run)."""
    f = Path(__file__).resolve().parent / "profiles" / "dominant_in_step.json"
    try:
        d = json.loads(f.read_text())
        k = d["kernels"][0]
        return {"kernel": k["kernel"], "launches_per_update": k["launches_per_update"], "gflop_per_update": k["gflop_per_update"],
                "us_per_update": k["us_per_update"], "tflops": k["tflops_in_step"], "frac_of_peak": k["frac_of_peak_in_step"],
                "source": f"profiles/dominant_in_step.json ({d.get('tag', '?')}, "
                          + ("the same kernel sources as this run" if d.get("kernel_sources_sha16") == kernel_sources_sha16() else "an earlier build")
                          + ": rocprofv3 kernel trace of this bench command + the library's launch log; kernel times of the graph's two "
                          "branches overlap)"}
    except (OSError, KeyError, IndexError, ValueError):
        return None


def _beat(what):
    """progress mark for the supervising parent (supervise_ranks): the child is alive and got this far"""
    f = os.environ.get("FBHIP_BENCH_HEARTBEAT")
    if f:
        try:
            Path(f).write_text(f"{time.time():.3f} {what}\n")
        except OSError:
            pass


def supervise_ranks(args, rank, world):
    """N > 1 under torch.distributed.run: every launched worker becomes a SUPERVISOR that holds no GPU context and runs the real
    bench rank in a child process, one attempt per gradient transport -- the library's RCCL communicator first (its pipelined graph,
    then its single-queue chain), then the torch.distributed schedule, then the peer-access kernels.  A multi-GPU node is available to this script once per round and
    none of the three transports has ever run on more than one physical device: an attempt that crashes on any rank, or whose
    slowest rank stops making progress (heartbeat file, --stall-timeout), is killed on ALL ranks (process groups this supervisor
    started, by pid) and the next transport gets a fresh rendezvous on another port.  The supervisors agree once a second through
    a gloo group on the launcher's rendezvous.  Rank 0 prints the successful child's JSON line with the attempt history added."""
    import subprocess
    import tempfile
    import signal
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=1800))
    first = "peer" if args.peer_allreduce else (args.transport or "rccl")
    transports = [first] if (args.global_batch or args.no_fallback_transports) else [first] + [t for t in ("rccl", "c10d", "peer") if t != first]
    # (transport, extra environment): the library's graph is PIPELINED by default since round 6 (the next step's head on a second branch);
    # that form has run on one device only, so its single-queue chain (FBHIP_DP_PIPELINE=0, the form of rounds 4-5) is the next attempt
    # before the transport itself is given up
    plan = []
    for t in transports:
        plan.append((t, {}))
        if t == "rccl" and not args.no_fallback_transports and os.environ.get("FBHIP_DP_PIPELINE") != "0":
            plan.append((t, {"FBHIP_DP_PIPELINE": "0"}))
    argv = [a for a in sys.argv[1:] if a != "--peer-allreduce"]
    while "--transport" in argv:
        i = argv.index("--transport")
        del argv[i:i + 2]
    argv = [a for a in argv if not a.startswith("--transport=")]
    history, line = [], None
    tmp = Path(tempfile.mkdtemp(prefix=f"fbhip_bench_r{rank}_"))
    for attempt, (tr, extra_env) in enumerate(plan):
        port = [0]
        if rank == 0:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port[0] = sk.getsockname()[1]
        dist.broadcast_object_list(port, src=0)
        hb, so = tmp / f"beat{attempt}", tmp / f"out{attempt}"
        env = dict(os.environ, MASTER_PORT=str(port[0]), FBHIP_BENCH_CHILD="1", FBHIP_BENCH_HEARTBEAT=str(hb),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **extra_env)
        for k in ("TORCHELASTIC_USE_AGENT_STORE",):          # the child ranks rendezvous among themselves: rank 0 hosts the store
            env.pop(k, None)
        t0 = time.time()
        with open(so, "wb") as fo:
            cmd = json.loads(os.environ["FBHIP_BENCH_CHILD_CMD"]) if os.environ.get("FBHIP_BENCH_CHILD_CMD") else \
                [sys.executable, str(Path(__file__).resolve())]                    # (the override: CPU tests of this function)
            child = subprocess.Popen(cmd + argv + ["--transport", tr], env=env, stdout=fo, stderr=None, start_new_session=True)
        outcome = None
        while True:
            time.sleep(1.0)
            rc = child.poll()
            last = hb.stat().st_mtime if hb.exists() else t0
            allow = args.stall_timeout if hb.exists() else max(args.stall_timeout, 420.0)      # (a fresh box pages torch in for minutes)
            mine_failed = (rc is not None and rc != 0) or (rc is None and time.time() - last > allow)
            st = torch.tensor([1.0 if mine_failed else 0.0, 1.0 if rc == 0 else 0.0])
            failed, done = st[:1].clone(), st[1:].clone()
            dist.all_reduce(failed, op=dist.ReduceOp.MAX)
            dist.all_reduce(done, op=dist.ReduceOp.MIN)
            if float(failed.item()) > 0:
                why = f"exit code {rc}" if (rc is not None and rc != 0) else ("no progress" if rc is None and mine_failed else "another rank failed")
                outcome = f"failed ({why} on rank {rank})" if why != "another rank failed" else "failed (another rank)"
                if child.poll() is None:
                    try:
                        os.killpg(child.pid, signal.SIGKILL)          # the session this supervisor started, nothing else
                    except ProcessLookupError:
                        pass
                child.wait()
                break
            if float(done.item()) > 0:
                outcome = "ok"
                break
        beat = hb.read_text().strip() if hb.exists() else "never started"
        every = [None] * world
        dist.all_gather_object(every, {"outcome": outcome, "last_progress": beat.split(" ", 1)[-1]})
        history.append({"transport": tr, **({"env": extra_env} if extra_env else {}), "seconds": round(time.time() - t0, 1), "ranks": every})
        if outcome == "ok":
            if rank == 0:
                text = so.read_text(errors="replace")
                for ln in text.splitlines():
                    if ln.startswith("{"):
                        line = ln
                    else:
                        print(ln, file=sys.stderr)
            break
        if rank == 0:
            print(f"bench.py: transport {tr} did not finish ({[e['outcome'] for e in every]}); "
                  + ("trying the next one" if attempt + 1 < len(plan) else "no transport left"), file=sys.stderr, flush=True)
    ok = [line is not None]
    dist.broadcast_object_list(ok, src=0)
    if rank == 0 and line is not None:
        out = json.loads(line)
        out.setdefault("data_parallel", {})["attempts"] = history
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    if not ok[0]:
        raise SystemExit("bench.py: no gradient transport completed: " + json.dumps(history))


def _branched_ok() -> bool:
    from controllable_agent_amd import _lib
    return bool(_lib.load().fbhip_branched_graphs(None))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--episodes", type=int, default=5000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps updates between barrier + synchronize pairs) is run this many times; "
                         "value / ms_per_step are the MEDIAN repeat, every repeat is listed under 'repeats'")
    ap.add_argument("--no-dominant-probe", action="store_true",
                    help="skip the launched-alone timing of the dominant GEMM shape (kernel traces of the step without its 55 extra launches)")
    ap.add_argument("--no-single-update-probe", action="store_true",
                    help="skip the extra measurement of plain agent.update() calls (config.single_update_steps_per_s)")
    ap.add_argument("--steps-per-launch", type=int, default=32,
                    help="consecutive updates handed to one hipGraph launch (FBHipAgent.update_many, as run_offline does "
                         "between two log lines); 1 = one launch per update.  With N > 1 the same call pipelines the steps "
                         "around the gradient all-reduces (the next step's sampling + online forward under the actor all-reduce)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on cuda:0 with the gloo backend (RCCL refuses two ranks per device): exercises the "
                         "multi-rank code path of this script on a 1-GPU box; the number it prints is NOT a scaling result")
    ap.add_argument("--nccl-world1", action="store_true",
                    help="with ONE rank under torch.distributed.run: still create the nccl (RCCL) process group, so that together "
                         "with FBHIP_FORCE_PHASE_SPLIT=1 the data-parallel schedule issues its real RCCL all-reduces (on one "
                         "rank): a kept execution of the RCCL code path on a 1-GPU box")
    ap.add_argument("--peer-allreduce", action="store_true",
                    help="N > 1: the gradient all-reduces as kernels INSIDE each rank's update graph (peers' buckets mapped with hipIpc, "
                         "csrc/peer.hip; FBHIP_DP_ALLREDUCE=peer) instead of RCCL calls between three phase graphs: one graph launch "
                         "per rank per --steps-per-launch updates.  Single node only")
    ap.add_argument("--transport", choices=("rccl", "c10d", "peer"), default=None,
                    help="N > 1: how the two gradient buckets are all-reduced.  rccl (default): the library's own RCCL communicator, "
                         "ncclAllReduce captured inside each rank's n-step update graph (csrc/rccl.hip); c10d: torch.distributed "
                         "collectives between / inside phase graphs (round 2's path); peer: hand-written peer-access kernels "
                         "(= --peer-allreduce).  Falls back to c10d, and says so in the JSON line, if the library transport cannot be set up")
    ap.add_argument("--stall-timeout", type=float, default=240.0,
                    help="N > 1: seconds without progress (initialisation, warm-up, each timed repeat) after which the supervising "
                         "parent gives an attempt up on every rank and tries the next transport")
    ap.add_argument("--no-fallback-transports", action="store_true", help="N > 1: one attempt, with --transport only")
    ap.add_argument("--no-supervisor", action="store_true",
                    help="N > 1: run the rank in the launched process itself (no child process, no second attempt)")
    ap.add_argument("--global-batch", action="store_true",
                    help="data-parallel mode B (FBHipAgent(dp_global_batch=True)): the exact loss of the concatenated "
                         "world x batch rows (one embedding all-gather per step) instead of per-rank blocks with gradient "
                         "averaging.  NOT the default bench line")
    ap.add_argument("--pretend-world", type=int, default=0,
                    help="with --global-batch on ONE rank: replicate the rank's embeddings N times in the exchange step, so the "
                         "pairwise kernel runs its share of an N x batch global loss (cost rehearsal of mode B at world N; the "
                         "loss itself is not meaningful)")
    ap.add_argument("--pmc", action="store_true",
                    help="collect roofline.traffic live: two extra rocprofv3 --pmc passes of a short run of this script (N = 1)")
    ap.add_argument("--workload", choices=("walker", "quadruped"), default="walker",
                    help="walker = configs[1], THE bench line; quadruped = configs[2] (no cpu_baseline / kernel probe)")
    args = ap.parse_args()
    W = WALKER if args.workload == "walker" else QUADRUPED
    goal_space = None if args.workload == "walker" else "simplified_quadruped"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain ``python bench.py --gpus N``: become the launcher -- one rank per GPU under torch.distributed.run on this node,
        # rendezvous on 127.0.0.1 and a free port; rank 0 of the children prints the one JSON line (the torchrun form documented
        # at the top keeps working: it arrives here with WORLD_SIZE set)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        sys.stdout.flush()
        os.execve(sys.executable, cmd, env)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launched under torch.distributed.run with another --nproc-per-node?)")
    if world > 1 and os.environ.get("FBHIP_BENCH_CHILD") != "1" and not args.no_supervisor:
        return supervise_ranks(args, rank, world)
    _beat("started")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane: with the library-owned RCCL transport (the default) or the peer kernels the process group only carries the
        # 128-byte unique id / the hipIpc handles, the replica checksums, the ranks' agreement on the transport and this script's
        # barriers: a gloo group.  Only the torch.distributed schedule (--transport c10d) needs an nccl group, for its all-reduces.
        # (Round 3 kept an nccl group here because the BRANCHED data-parallel graph happened to replay 2.3x faster beside one; that
        # graph form is gone -- the data-parallel graph is single-queue by construction -- and with it the reason.)
        transport = "peer" if args.peer_allreduce else (args.transport or "rccl")
        if args.rehearse_on_one_gpu or transport in ("rccl", "peer"):
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    elif args.nccl_world1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29535")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(os.environ.get("FBHIP_BENCH_WORLD1_BACKEND", "nccl"), **({"device_id": torch.device(dev)} if os.environ.get("FBHIP_BENCH_WORLD1_BACKEND", "nccl") == "nccl" else {}))

    if os.environ.get("FBHIP_BENCH_EXTRA_STREAMS"):          # diagnostic: shift the runtime's stream -> hardware-queue assignment
        _extra_streams = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ["FBHIP_BENCH_EXTRA_STREAMS"]))]
    _beat("process group up")
    if args.peer_allreduce or args.transport == "peer":
        args.peer_allreduce = True
        os.environ["FBHIP_DP_ALLREDUCE"] = "peer"
    elif args.transport is not None:
        os.environ["FBHIP_DP_ALLREDUCE"] = args.transport
    if args.pretend_world > 1:
        os.environ["FBHIP_PRETEND_WORLD"] = str(args.pretend_world)
    from controllable_agent_amd.agent import FBHipAgent
    torch.manual_seed(1)                       # identical initial weights on every rank
    agent = FBHipAgent(obs_type="states", obs_shape=(W["obs_dim"],), action_shape=(W["action_dim"],),
                       device=dev, num_expl_steps=0, update_every_steps=1, batch_size=W["batch_size"],
                       z_dim=W["z_dim"], hidden_dim=W["hidden_dim"], feature_dim=W["feature_dim"],
                       backward_hidden_dim=W["backward_hidden_dim"], goal_space=goal_space,
                       use_tb=False, use_wandb=False, use_hiplog=False, dp_global_batch=args.global_batch)
    # each rank's shard of the 5000-episode buffer (episodes ep % world == rank  <=>  an independent 5000/world-episode draw)
    n_eps = max(args.episodes // world, 8) if args.workload == "walker" else max(min(args.episodes, 1000) // world, 8)
    rb = make_replay(n_eps, 1000, W["obs_dim"], W["action_dim"], dev, seed=100 + rank,
                     goal_dim=W["goal_dim"] if goal_space else None)

    _beat("agent and replay shard built")
    inject = os.environ.get("FBHIP_BENCH_FAIL_TRANSPORT", "")          # tests only: "<transport>:crash" | "<transport>:hang" on rank 1
    if world > 1 and rank == 1 and inject.split(":")[0] == (args.transport or "rccl"):
        if inject.endswith(":hang"):
            time.sleep(3600)
        os._exit(7)

    def barrier():
        agent.flush()                             # update() calls the agent still holds back (deferred batching) go out first
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    spl = max(1, args.steps_per_launch) if not args.global_batch else 1

    def run(first_step, n_steps):                 # exactly n_steps updates, spl per graph launch
        done = 0
        while done < n_steps:
            k = min(spl, n_steps - done)
            if k == 1:
                agent.update(rb, first_step + done)
            else:
                agent.update_many(rb, first_step + done, k)
            done += k

    # everything is enqueued on ONE explicit stream so that the HIP events below bracket the same launches the wall clock does
    # (torch.cuda.Event only sees the stream it is recorded on; on the legacy default stream the agent would hop to its own)
    bench_stream = torch.cuda.default_stream(dev) if os.environ.get("FBHIP_BENCH_LEGACY_STREAM") == "1" else torch.cuda.Stream(device=dev, priority=-1)
    with torch.cuda.stream(bench_stream):
        run(0, args.warmup)
        torch.cuda.synchronize()
        _beat("warm-up done")
        if (world > 1 and not args.rehearse_on_one_gpu and os.environ.get("FBHIP_DP_ALLREDUCE", "rccl") == "rccl" and
                getattr(agent, "_rccl_failed", False) and dist.get_backend() != "nccl"):
            # the library transport was refused (the agent's ranks agreed on it, FBHipAgent._rccl_ready / _rccl_run; the JSON line
            # says why): every rank is on the torch.distributed schedule now, whose all-reduces want an nccl group -- host-side
            # gloo collectives are not what this line times
            torch.cuda.synchronize()
            dist.destroy_process_group()
            dist.init_process_group("nccl", device_id=torch.device(dev))
            run(0, args.warmup)
        # N > 1, gradients reduced inside the library's graph: that graph has two forms since round 6 -- pipelined (the next step's
        # head on a second branch; the default) and the single-queue chain (FBHIP_DP_PIPELINE=0; rounds 4-5).  The pipelined form has
        # only ever run on ONE device (a branched data-parallel graph replayed at half speed in round 3, for a reason that was a
        # launch-queue matter then), and a slow replay is not something the supervisor can see.  So, unless the environment pins the
        # form: a short trial of both on every rank (8 launches each, outside the timed region), the slower rank's time decides, all ranks take the same form; the line
        # reports both rates (data_parallel.graph_form_trial).
        form_trial = None
        in_graph = world > 1 and spl > 1 and not args.global_batch and not getattr(agent, "_rccl_failed", False) and \
            os.environ.get("FBHIP_DP_ALLREDUCE", "rccl") in ("rccl", "peer") and (args.peer_allreduce or getattr(agent, "_rccl_bound", False))
        if in_graph and "FBHIP_DP_PIPELINE" not in os.environ and os.environ.get("FBHIP_UPDATE_PIPELINE") != "0" and _branched_ok():
            rates = {}
            for form, env in (("chain", "0"), ("pipelined", "1")):
                os.environ["FBHIP_DP_PIPELINE"] = env
                agent._rccl_prepared = []                 # (the other form's graphs are PREPARED under the ranks' agreement, like the first)
                run(0, 2 * spl)
                barrier()
                t0 = time.perf_counter()
                run(0, 8 * spl)
                barrier()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                rates[form] = 8 * spl / float(t.item())
            chosen = "pipelined" if rates["pipelined"] >= rates["chain"] else "chain"
            os.environ["FBHIP_DP_PIPELINE"] = "1" if chosen == "pipelined" else "0"
            agent._rccl_prepared = []
            form_trial = {"per_rank_steps_per_s": rates, "chosen": chosen, "trial": f"8 x {spl} steps per form after 2 x {spl} untimed ones, slowest rank"}
            _beat(f"graph form trial done: {chosen}")
        # every graph size the timed region will launch must already be captured (a capture costs milliseconds): one extra
        # untimed launch of each size (these are additional warm-up steps)
        sizes = ({spl} if args.steps >= spl else set()) | ({args.steps % spl} if args.steps % spl else set())
        for sz in sorted(sizes):
            run(args.warmup, sz)
        walls, events, host_enqueue, per_rank_walls = [], [], [], []
        for rep in range(max(1, args.repeats)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            t0 = time.perf_counter()
            e0.record(bench_stream)
            run(args.warmup + rep * args.steps, args.steps)
            host_enqueue.append(time.perf_counter() - t0)         # the host's share: every launch of the region is enqueued by now
            e1.record(bench_stream)
            barrier()
            wall = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([wall], device=dev, dtype=torch.float64)
                mine = [torch.zeros_like(t) for _ in range(world)]
                dist.all_gather(mine, t)
                per_rank_walls.append([float(x.item()) for x in mine])
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                wall = float(t.item())
            walls.append(wall)
            events.append(e0.elapsed_time(e1) * 1e-3)
            _beat(f"timed repeat {rep + 1} of {max(1, args.repeats)} done")
        # the host's own cost of a data-parallel step: one update_many call into an EMPTY queue returns as soon as its launches
        # (and, on the RCCL / gloo path, its collectives) are issued
        host_idle = None
        if world > 1 and spl > 1:
            barrier()
            th = time.perf_counter()
            run(0, spl)
            host_idle = (time.perf_counter() - th) / spl
            barrier()
        order = sorted(range(len(walls)), key=lambda i: walls[i])
        mid = order[len(order) // 2]
        dt = walls[mid]                                   # the median repeat IS one contiguous region of exactly --steps updates
        single = None
        if world == 1 and args.workload == "walker" and not args.no_single_update_probe and not args.global_batch:
            # what a reference workspace literally does: agent.update(replay_loader, step) once per loop iteration
            # (train_offline.py:118).  With metrics off the agent queues such calls and launches them as n-step graphs
            # (FBHipAgent "deferred batching"); barrier() flushes the rest of the queue before it synchronises
            n1 = max(300, min(args.steps, 1000))
            # a queue goes out as graphs of 32 / 16 / 8 / 4 / 2 / 1 steps (FBHipAgent.DEFER_MENU): warm every size first -- one
            # eager call, a full queue, and 31 = 16 + 8 + 4 + 2 + 1 -- so that the timed loop below captures nothing
            for i in range(1 + 32 + 31):
                agent.update(rb, i)
            barrier()
            caps0 = agent.graph_captures()
            t1 = time.perf_counter()
            for i in range(n1):
                agent.update(rb, i)
            barrier()
            single = n1 / (time.perf_counter() - t1)
            single_captures = agent.graph_captures() - caps0

    replicas = None
    if world > 1:
        # data-parallel replicas must END identical (the schedule never broadcasts parameters): every rank's float64
        # checksums of its parameter / target buffers, gathered and compared on rank 0
        mine = torch.stack([t.double().sum() for t in agent._replica_buffers()] +
                           [t.double().pow(2).sum() for t in agent._replica_buffers()]).cpu()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        if dist.get_backend() == "nccl":
            g_dev = [torch.zeros_like(mine, device=dev) for _ in range(world)]
            dist.all_gather(g_dev, mine.to(dev))
            gathered = [g.cpu() for g in g_dev]
        else:
            dist.all_gather(gathered, mine)
        replicas = {"identical": all(torch.equal(gathered[0], g) for g in gathered[1:]),
                    "checksum_rank0": [float(x) for x in gathered[0][:3]], "ranks": world, "adam_steps": list(agent.step_counts())}

    if rank == 0:
        steps_per_s = args.steps / dt                     # per-rank update rate (== global step rate)
        value = world * steps_per_s                       # update-steps/s summed over ranks (weak scaling)
        gflop = algorithmic_gflop_per_update(W["obs_dim"], W["action_dim"], W["goal_dim"], W["z_dim"], W["hidden_dim"],
                                             W["feature_dim"], W["backward_hidden_dim"], W["batch_size"])
        achieved = gflop * steps_per_s / 1e3              # TFLOP/s per GPU
        traffic, traffic_src = measured_traffic(args.workload)
        if args.pmc and world == 1:
            live, src = live_traffic(args.workload)
            traffic, traffic_src = (live, src) if live is not None else (traffic, f"{src}; fell back to {traffic_src}")
        out = {
            "metric": f"FB update-steps/sec (batch={W['batch_size']}, z_dim={W['z_dim']})", "value": value, "unit": "update-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not args.rehearse_on_one_gpu else "synthetic; REHEARSAL: all ranks share cuda:0 over gloo",
            "config": {"workload": ("fb_ddpg offline on walker_walk replay (configs[1]): obs 24, action 6, z_dim 50, "
                                    "hidden 1024, feature 512, backward hidden 526; batch 1024 per GPU; "
                                    f"{args.episodes}-episode x 1000-step synthetic RND-style replay resident in HBM; "
                                    "metrics off in the timed loop (reference default)") if args.workload == "walker" else
                                   ("fb_ddpg offline on quadruped_walk replay (configs[2]): obs 78, action 12, goal space "
                                    f"simplified_quadruped (g=2), z_dim 100, batch 2048 per GPU; {n_eps}-episode x 1000-step "
                                    "synthetic replay resident in HBM; metrics off"),
                       "steps_per_graph_launch": spl, "runtime_env": {"ROC_CPU_WAIT_FOR_SIGNAL": os.environ.get("ROC_CPU_WAIT_FOR_SIGNAL")}, **({"dp_loss": "global batch (mode B)" + (f", pretend world {args.pretend_world}" if args.pretend_world > 1 else "")} if args.global_batch else {}),
                       "global_batch": W["batch_size"] * world, "parallelism": f"dp{world}",
                       **({"allreduce": getattr(agent, "_dp_transport", None) or
                                        ("peer-access kernels inside the update graph (csrc/peer.hip)" if args.peer_allreduce else
                                         ("gloo (rehearsal)" if args.rehearse_on_one_gpu else "RCCL via torch.distributed between phase graphs"))} if world > 1 else {}),
                       "host_enqueue_ms_per_step": 1e3 * host_enqueue[mid] / args.steps,
                       **({"host_issue_ms_per_step_idle_queue": 1e3 * host_idle} if host_idle is not None else {}),
                       "updates_per_hour": 3600 * value, "steps_per_s_per_gpu": steps_per_s,
                       # data parallel = gradient averaging: ONE optimiser step per global step, on a batch of world x B samples
                       "global_steps_per_s": steps_per_s, "samples_per_s": world * W["batch_size"] * steps_per_s,
                       **({"value_is": f"{world} ranks x {steps_per_s:.1f} per-rank update-steps/s of batch {W['batch_size']} each (weak scaling: "
                                       f"batch-{W['batch_size']}-equivalents per second); the model itself makes global_steps_per_s "
                                       "optimiser steps per second on the global batch"} if world > 1 else {}),
                       **({"single_update_steps_per_s": single, "single_update_probe_graph_captures": single_captures} if single is not None else {})},
            "repeats": {"n": len(walls), "steps_each": args.steps, "reported": "median by wall time",
                        "wall_s": walls, "hip_event_s": events,
                        "steps_per_s": {"median": world * args.steps / dt, "min": world * args.steps / max(walls),
                                        "max": world * args.steps / min(walls)},
                        "hip_event_steps_per_s_median_repeat": world * args.steps / events[mid]},
            "timed_region_s": dt,
            **({"replicas": replicas} if replicas is not None else {}),
            **({"data_parallel": {
                # which path actually carried the gradients
                "transport": getattr(agent, "_dp_transport", None) or
                             ("peer" if args.peer_allreduce else ("gloo-host" if args.rehearse_on_one_gpu else "c10d-rccl-host")),
                "library_rccl_refused": bool(getattr(agent, "_rccl_failed", False)),
                "graph_form": ("single-queue (one stream: every step's phases and both all-reduces in program order)"
                               if os.environ.get("FBHIP_DP_PIPELINE") == "0" or os.environ.get("FBHIP_UPDATE_PIPELINE") == "0" or not _branched_ok() else
                               "pipelined (step t+1's head on a second graph branch beside step t's actor pass, actor all-reduce and actor "
                               "step; both all-reduces on the main branch in program order; FBHIP_DP_PIPELINE=0 = the single-queue chain)"),
                **({"graph_form_trial": form_trial} if form_trial is not None else {}),
                "control_plane": dist.get_backend(),
                "nccl_env": {k: os.environ.get(k, "default") for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                                     "HSA_ENABLE_IPC_MODE_LEGACY")},
                "per_rank_steps_per_s_median_repeat": [args.steps / w for w in per_rank_walls[mid]] if per_rank_walls else None,
                "scaling_efficiency_note": "efficiency = value / (N x the value of the same script at --gpus 1); the driver computes it from its own N = 1 run"}}
               if world > 1 else {}),
            **({"flags": [f"timed region of {dt * 1e3:.1f} ms < 0.5 s: --steps {args.steps} is too short for a stable rate "
                          f"(host launch jitter); the {len(walls)} repeats bound it, prefer --steps >= 1000"]} if dt < 0.5 else {}),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "what": f"whole update step: {gflop:.2f} algorithmic GFLOP/update (SURVEY.md section 8d) x "
                                 "measured updates/s, per GPU, vs the exact-fp32 MFMA peak"},
        }
        if args.workload == "walker" and world == 1:
            if not args.no_dominant_probe:
                out["roofline"]["dominant_kernel"] = dominant_kernel_probe()
                out["roofline"]["dominant_kernel"]["in_step"] = dominant_in_step()
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1 or args.nccl_world1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
