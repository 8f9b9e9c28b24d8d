#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of environment settings through bench.py.
#   tools/env_ab.sh <rounds> "<ENV_A>" "<ENV_B>" ... [-- extra bench flags]      each ENV_x: space-separated VAR=value list ("-" = none)
set -u
ROUNDS=$1; shift
VARIANTS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARIANTS+=("$1"); shift; done
[ $# -gt 0 ] && shift
val() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(round(d['value'],1), round(d['config'].get('single_update_steps_per_s') or 0,1))" $1; }
for r in $(seq 1 $ROUNDS); do
  for v in "${VARIANTS[@]}"; do
    if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
    env $envs python bench.py --steps 960 --warmup 96 --repeats 3 --no-cpu-baseline "$@" > /tmp/v.json 2>/tmp/v.err || tail -3 /tmp/v.err
    echo "[$v] $(val /tmp/v.json)"
  done
done
