# usage: bash tools/run_variants.sh  (GPU box) -- quick A/B of environment knobs on the default bench
for v in "BBP_SERIAL_LDS=163840" "BBP_SERIAL_LDS=163840 BBP_SLICES=1" "BBP_SERIAL_LDS=163840 BBP_SLICES=3" "BBP_SERIAL_LDS=159744"; do
  echo "== $v"
  env $v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>&1 | tail -1 | cut -c1-140
done
