#!/bin/bash
# GPU box: A/B of environment knobs on the default bench.  usage: bash tools/ab_env.sh ROUNDS "KNOB=v KNOB2=w" "..." ("X=1" = defaults)
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-build 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels_us',{})
def avg(n):
    e=k.get(n); return round(e['total_us']/e['launches']) if e else None
print('env [$v]', round(d['value']), round(d['ms_per_step'],2), 'alu.frac', round(d['roofline']['alu']['frac'],3), 'open_serial us', avg('k_open_serial(witness+rng)'), 'acc us', avg('k_msm_acc'), 'fold us', avg('k_msm_fold'))"
  done
done
