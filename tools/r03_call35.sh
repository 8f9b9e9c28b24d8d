#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python bench.py > $OUT/r03m_bench.json 2> $OUT/r03m_bench.err; tail -c 2000 $OUT/r03m_bench.json | cut -c1-2000; echo
timeout 200 python bench.py --workload quadruped --steps 640 --warmup 64 --repeats 3 > $OUT/r03m_quadruped_bench.json 2>/dev/null; cut -c1-160 $OUT/r03m_quadruped_bench.json
timeout 200 python tools/sf_bench.py --learner icm --steps 640 --warmup 64 --no-cpu-baseline > $OUT/r03m_sf_icm_bench.json 2>/dev/null; cut -c1-160 $OUT/r03m_sf_icm_bench.json
timeout 300 python bench.py --gpus 2 --rehearse-on-one-gpu --peer-allreduce --steps 320 --warmup 32 --repeats 3 --episodes 1000 --no-cpu-baseline --no-fallback-transports > $OUT/r03m_two_rank_peer.log 2>&1; grep "^{" $OUT/r03m_two_rank_peer.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['data_parallel']['graph_form'])"
ROC_CPU_WAIT_FOR_SIGNAL=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03m_suite_cpu_wait.log 2>&1; grep -E "passed|failed" $OUT/r03m_suite_cpu_wait.log | tail -3 | cut -c1-300
