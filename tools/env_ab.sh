#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of an environment switch through bench.py (alternating pairs; boxes of the pool differ by +-3 %).
#   tools/env_ab.sh VAR A_VALUE B_VALUE [rounds] [extra bench.py flags]     e.g.  tools/env_ab.sh FBHIP_HEAD_TILES 0 1 3 --workload quadruped
set -u
VAR=$1; A=$2; B=$3; ROUNDS=${4:-2}; shift 4 || shift $#
val() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(d['value'], d['config'].get('single_update_steps_per_s'))" $1; }
for r in $(seq 1 $ROUNDS); do
  env $VAR=$A python bench.py --steps 960 --warmup 96 --repeats 3 --no-cpu-baseline --no-dominant-probe "$@" > /tmp/a.json 2>/tmp/a.err || tail -3 /tmp/a.err
  echo "$VAR=$A  $(val /tmp/a.json)"
  env $VAR=$B python bench.py --steps 960 --warmup 96 --repeats 3 --no-cpu-baseline --no-dominant-probe "$@" > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  echo "$VAR=$B  $(val /tmp/b.json)"
done
