#!/bin/bash
# GPU box: the three rocprofv3 passes behind profiles/ (kernel stats, then FETCH_SIZE and WRITE_SIZE in their own runs)
set -e
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 /root/repo/bench.py --steps 5 --warmup 2 --no-also --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 /root/repo/bench.py --steps 1 --warmup 1 --no-also --no-cpu-baseline > /dev/null 2> "$OUT/fetch.log"
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 /root/repo/bench.py --steps 1 --warmup 1 --no-also --no-cpu-baseline > /dev/null 2> "$OUT/write.log"
echo "write pass done"
find "$OUT" -name "*.csv" | head -20
