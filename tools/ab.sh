#!/bin/bash
# GPU box: A/B of experiment builds on the default bench.  usage: bash tools/ab.sh ROUNDS variant1 variant2 ...   ("-" = the product library)
# Variants are built beforehand with tools/build_variant.py (they travel with the snapshot).
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset BBP_LIB_VARIANT; else export BBP_LIB_VARIANT=$v; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-build 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant [$v]', round(d['value']), round(d['ms_per_step'],2), 'alu.frac', round(d['roofline']['alu']['frac'],3))"
  done
done
