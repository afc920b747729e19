# usage: bash tools/run_variants.sh  (GPU box) -- quick A/B of environment knobs on the default bench
for v in "BBP_SLICES=3" "BBP_SLICES=2" "GPU_MAX_HW_QUEUES=8 BBP_SLICES=4" "BBP_SLICES=3"; do
  echo "== $v"
  env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done
