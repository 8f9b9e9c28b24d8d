#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_update_parity_gpu.py -m gpu -q -p no:cacheprovider -k "legacy or update_many or pipelined" 2>&1 | tail -4 | cut -c1-300
{
echo "# legacy default stream vs explicit stream, and the two ways of ordering the caller's legacy-stream work (gate = blocking helper streams, event = events on the legacy stream)"
for mode in gate event; do
  echo "== sf icm, legacy stream, order=$mode"; FBHIP_LEGACY_STREAM_ORDER=$mode python tools/sf_bench.py --learner icm --steps 640 --warmup 64 --no-cpu-baseline 2>/dev/null | cut -c1-230
  echo "== fb bench, legacy stream, order=$mode"; FBHIP_LEGACY_STREAM_ORDER=$mode FBHIP_BENCH_LEGACY_STREAM=1 python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'single_update', d['config'].get('single_update_steps_per_s'))"
done
echo "== sf icm, explicit stream"; python tools/sf_bench.py --learner icm --steps 640 --warmup 64 --no-cpu-baseline --explicit-stream 2>/dev/null | cut -c1-230
echo "== fb bench, explicit stream"; python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'single_update', d['config'].get('single_update_steps_per_s'))"
} > $OUT/r03_legacy_stream.txt 2>&1
cat $OUT/r03_legacy_stream.txt | cut -c1-250
