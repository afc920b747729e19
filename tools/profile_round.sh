#!/bin/bash
# GPU box: the rocprofv3 passes behind profiles/ (kernel stats, then FETCH_SIZE and WRITE_SIZE in their own runs).
#   bash tools/profile_round.sh [round-tag, default r02]
# Every profiled command is `python3 bench.py ... --no-build`: native artefacts are (re)built HERE, before any profiler starts,
# so the profiled process never spawns a compiler (its children would inherit the profiler's preloaded library, which has
# initialised the GPU -- the exec hop this pool forbids).
set -e
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
python3 $REPO/__graft_entry__.py > "$OUT/build.log" 2>&1
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build"
# 1. the shipped configuration (three heavy-stage slices: launches of different slices overlap, per-launch times are stretched)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $B --steps 5 --warmup 2 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
echo "stats pass done"
# 2. EXCLUSIVE pass: one slice, so no two MSM launches overlap and a launch covers the whole 1024-proof batch
export BBP_SLICES=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/excl" -o excl -- $B --steps 5 --warmup 2 > "$OUT/bench_exclusive_under_rocprof.json" 2> "$OUT/excl.log"
echo "exclusive stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $B --steps 1 --warmup 1 > /dev/null 2> "$OUT/fetch.log"
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $B --steps 1 --warmup 1 > /dev/null 2> "$OUT/write.log"
echo "write pass done"
unset BBP_SLICES
# 3. configs[3] shard: 8192 verifications per step
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/verify" -o verify -- $B --workload verify --batch 8192 --steps 3 --warmup 1 > "$OUT/bench_verify8192_under_rocprof.json" 2> "$OUT/verify.log"
echo "verify stats pass done"
python3 $REPO/tools/pmc_aggregate.py "$OUT/fetch" "$OUT/write" > "$OUT/${TAG}_rocprofv3_pmc_hbm_exclusive.csv"
find "$OUT" -name "*stats*.csv" -o -name "*kernel_stats*.csv" | head -20
