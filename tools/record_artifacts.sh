#!/bin/bash
# Run ON THE GPU BOX (through gpurun): every number DESIGN.md quotes that is not part of tools/profile_round.sh, each into its
# own file under gpurun_out/<tag>_*  (copy into profiles/ afterwards).
#   tools/record_artifacts.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
# configs[2] (quadruped + goal space, batch 2048, z_dim 100): below, under tools/profile_cmd.sh
# configs[4] rehearsal: online loop at quadruped dims, 2000-episode ring, synthetic env (free, and 300 us of emulated physics)
python tools/online_bench.py --frames 12000 > $OUT/${TAG}_online_bench.txt 2>&1
python tools/online_bench.py --frames 12000 --env-us 300 >> $OUT/${TAG}_online_bench.txt 2>&1
# metrics dict ON, read back after every update (and the single-update-graph yardstick with metrics off; the host-side breakdown)
python tools/metrics_on_bench.py > $OUT/${TAG}_metrics_on_bench.txt 2>&1
python tools/metrics_on_bench.py --off >> $OUT/${TAG}_metrics_on_bench.txt 2>&1
python tools/metrics_turnaround.py 2>&1 | grep "metrics ON" > $OUT/${TAG}_metrics_turnaround.txt
# siblings
python tools/discrete_bench.py > $OUT/${TAG}_discrete_bench.json 2> $OUT/${TAG}_discrete_bench.err
python tools/sf_bench.py --learner icm > $OUT/${TAG}_sf_icm_bench.json 2> $OUT/${TAG}_sf_icm_bench.err
python tools/sf_bench.py --learner lap > $OUT/${TAG}_sf_lap_bench.json 2> $OUT/${TAG}_sf_lap_bench.err
# configs[2] under the profiler: kernel table, timeline, PMC passes
tools/profile_cmd.sh ${TAG}_quadruped "python bench.py --workload quadruped --steps 1000 --warmup 100 --repeats 3" "python bench.py --workload quadruped --steps 160 --warmup 32 --repeats 1 --no-cpu-baseline --no-single-update-probe" > /dev/null 2>&1
# sustained rate + finiteness
python tools/soak.py 100000 > $OUT/${TAG}_soak_100k.txt 2>&1
# batch-1 act / compute_z_correl latency
python tools/act_bench.py > $OUT/${TAG}_act_bench.txt 2>&1
# the RCCL code path with one rank: torchrun + backend nccl, NCCL_DEBUG=INFO kept
NCCL_DEBUG=INFO FBHIP_FORCE_PHASE_SPLIT=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port 29533 bench.py --gpus 1 --nccl-world1 --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline > $OUT/${TAG}_torchrun_world1_nccl.log 2>&1
# the same schedule without torchrun / without the process group (phase split forced, no collectives): what the all-reduce calls cost
FBHIP_FORCE_PHASE_SPLIT=1 python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline > $OUT/${TAG}_phase_split_world1.json 2>&1
# mode B (exact global batch) cost rehearsal
for N in 1 2 8; do
  python bench.py --global-batch --pretend-world $N --steps 400 --warmup 40 --repeats 3 --no-cpu-baseline > $OUT/${TAG}_mode_b_pretend_world$N.json 2>&1
done
# two-rank rehearsal of bench.py on this one GPU (gloo): replicas identical?
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 \
    --rehearse-on-one-gpu --steps 320 --warmup 32 --repeats 3 --episodes 1000 --no-cpu-baseline > $OUT/${TAG}_two_rank_rehearsal.log 2>&1
ls -la $OUT/${TAG}_*
# the same two ranks with the gradient all-reduces as peer-access kernels inside each rank's graph (csrc/peer.hip)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 \
    --rehearse-on-one-gpu --peer-allreduce --steps 320 --warmup 32 --repeats 3 --episodes 1000 --no-cpu-baseline > $OUT/${TAG}_two_rank_peer_allreduce.log 2>&1
