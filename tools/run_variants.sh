# usage: bash tools/run_variants.sh  (GPU box) -- quick A/B of environment knobs on the default bench
for v in "BBP_STAGGER=0" "BBP_STAGGER=1" "BBP_STAGGER=3" "BBP_STAGGER=1 BBP_SLICES=2" "BBP_STAGGER=3 BBP_SLICES=2"; do
  echo "== $v"
  env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))"
done
