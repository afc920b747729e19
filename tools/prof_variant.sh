#!/bin/bash
# GPU box: exclusive kernel statistics (single-slice prove pass) for experiment builds.  usage: bash tools/prof_variant.sh variant...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_var
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export BBP_SLICES=1 BBP_BENCH_NO_CHECK=1
for v in "$@"; do
  if [ "$v" = "-" ]; then unset BBP_LIB_VARIANT; n=product; else export BBP_LIB_VARIANT=$v; n=$v; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$n" -o x -- python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --steps 4 --warmup 2 > "$OUT/$n.json" 2> "$OUT/$n.log"
  python3 - "$OUT/$n/x_kernel_stats.csv" "$n" <<'PY'
import csv,sys
print("==", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ("k_msm","k_tail","k_encode","k_ipa_round","k_flatten","k_open")):
        print("  %-40s calls %4s avg %8.1f us total %8.2f ms" % (n[:40], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
