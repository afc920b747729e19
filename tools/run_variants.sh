for v in w2 w3 w4; do
  echo "== $v"
  BBP_LIB_VARIANT=$v BBP_SLICES=1 BBP_TAIL_ROUND=12 timeout -k 10 200 python tools/prof_step.py 1024 8 prove 2>&1 | grep k_msm | tr '\n' ' ' | sed 's/k_msm *//g'
  echo
  BBP_LIB_VARIANT=$v timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also 2>&1 | tail -1 | cut -c1-140
done
