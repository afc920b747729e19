# usage: bash tools/run_variants.sh  (GPU box) -- quick A/B of environment knobs on the default bench
for v in "BBP_SERIAL_BLOCK=64" "BBP_SERIAL_BLOCK=128" "BBP_SERIAL_BLOCK=256" "BBP_SERIAL_BLOCK=64" "BBP_SERIAL_BLOCK=256"; do
  echo "== $v"
  env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v['total_us']/d['steps']/1000,1) for k,v in d['kernels_us'].items()})"
done
