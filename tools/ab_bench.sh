#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of two builds of libfbhip.so through bench.py (boxes of the pool differ by +-3 %).
#   tools/ab_bench.sh <baseline.so> [rounds] [extra bench.py flags]   -> one "A"/"B" line per run: update-steps/s
# The baseline library must be ABI-compatible with the Python side only as far as bench.py goes; it is swapped in place.
set -u
BASE=$1; ROUNDS=${2:-2}; shift; shift || true
LIB=controllable_agent_amd/libfbhip.so
cp $LIB /tmp/new.so
val() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(d['value'], d['config'].get('single_update_steps_per_s'))" $1; }
for r in $(seq 1 $ROUNDS); do
  cp $BASE $LIB; python bench.py --steps 960 --warmup 96 --repeats 3 --no-cpu-baseline "$@" > /tmp/a.json 2>/tmp/a.err || tail -3 /tmp/a.err
  echo "A(base) $(val /tmp/a.json)"
  cp /tmp/new.so $LIB; python bench.py --steps 960 --warmup 96 --repeats 3 --no-cpu-baseline "$@" > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  echo "B(new)  $(val /tmp/b.json)"
done
cp /tmp/new.so $LIB
