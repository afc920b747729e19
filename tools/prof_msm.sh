#!/bin/bash
# GPU box: exclusive kernel statistics of the MSM stage alone (configs[1] unit) and of a single-slice prove pass.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_msm
rm -rf "$OUT"; mkdir -p "$OUT"
python3 $REPO/__graft_entry__.py > "$OUT/build.log" 2>&1
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/msm" -o msm -- $B --workload msm --steps 5 --warmup 2 > "$OUT/msm.json" 2> "$OUT/msm.log"
export BBP_SLICES=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/excl" -o excl -- $B --steps 5 --warmup 2 > "$OUT/excl.json" 2> "$OUT/excl.log"
find "$OUT" -name "*kernel_stats.csv"
