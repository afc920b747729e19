#!/bin/bash
# GPU box: kernel stats of the SHIPPED prove schedule (whole calls in rotation once the caller's pipeline is deep) and of the exclusive
# pass, without bbp_reserve's dummy batches in the profile; then the un-profiled bench line.   bash tools/profile_shipped.sh [tag]
set -e
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof2
rm -rf "$OUT"; mkdir -p "$OUT"
python3 $REPO/__graft_entry__.py > "$OUT/build.log" 2>&1
cd /tmp && export TMPDIR=/tmp
export BBP_BENCH_NO_RESERVE=1
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --no-exclusive"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $B --steps 10 --warmup 6 > "$OUT/${TAG}_bench_line_under_rocprof.json" 2> "$OUT/stats.log"
echo "shipped stats pass done"
BBP_SLICES=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/excl" -o excl -- $B --steps 5 --warmup 2 > "$OUT/${TAG}_bench_line_exclusive_under_rocprof.json" 2> "$OUT/excl.log"
echo "exclusive stats pass done"
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_rocprofv3_kernel_stats.csv"
cp $(find "$OUT/excl" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_rocprofv3_kernel_stats_exclusive.csv"
unset BBP_BENCH_NO_RESERVE
cd $REPO && python3 bench.py > "$OUT/${TAG}_bench_line.json" 2> "$OUT/bench.err"
echo "bench line done"
grep -E "k_msm_acc|k_msm_fold|k_msm_sort|k_open" "$OUT/${TAG}_rocprofv3_kernel_stats.csv" | cut -c1-60,120-
