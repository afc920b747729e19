#!/bin/bash
# GPU box: A/B of environment knobs on any bench workload.  usage: bash tools/ab_env_wl.sh ROUNDS "bench args" "KNOB=v" ... ("X=1" = defaults)
R=$1; A=$2; shift; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py $A --no-cpu-baseline --no-also --no-build 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('env [$v]', round(d['value']), round(d['ms_per_step'],3))"
  done
done
