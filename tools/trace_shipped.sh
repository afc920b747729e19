#!/bin/bash
# GPU box: kernel trace of the shipped configuration (three slices), analysed by tools/timeline_stats.py and tools/timeline_gaps.py
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -o x -- python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --steps 6 --warmup 2 > "$OUT/bench.json" 2> "$OUT/log.txt"
T=$(find "$OUT" -name "x_kernel_trace.csv")
python3 $REPO/tools/timeline_stats.py $T
python3 $REPO/tools/timeline_gaps.py $T
