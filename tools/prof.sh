#!/bin/bash
# GPU box: ONE kernel-level profiling entry (replaces prof_msm / prof_variant / prof_verify / verify_profile / single_profile /
# trace_shipped / midsize_trace of rounds 1-3).
#   bash tools/prof.sh stats  NAME "<program and args>"      rocprofv3 --kernel-trace --stats, per-kernel table of the bbp:: kernels
#   bash tools/prof.sh trace  NAME "<program and args>"      rocprofv3 --kernel-trace, then tools/timeline_stats.py + timeline_gaps.py
# <program and args> is what goes after `--` and must be a python3 program (no env / bash hop under the profiler), e.g.
#   bash tools/prof.sh stats verify1024 "python3 bench.py --workload verify --batch 1024 --steps 10 --warmup 2 --no-also --no-cpu-baseline --no-build --no-exclusive"
#   BBP_SLICES=1 BBP_LIB_VARIANT=coop bash tools/prof.sh stats excl_coop "python3 bench.py --steps 4 --warmup 2 --no-also --no-cpu-baseline --no-build --no-exclusive"
#   bash tools/prof.sh trace midsize256 "python3 tools/midsize.py 256"
# Environment knobs (BBP_*, BBP_LIB_VARIANT) are taken from the caller's environment: set them in front of `bash`.
set -e
MODE=$1; NAME=$2; CMD=$3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
python3 $REPO/__graft_entry__.py > "$OUT/build.log" 2>&1   # artefacts are built BEFORE the profiler starts
cd $REPO && export TMPDIR=/tmp
if [ "$MODE" = "stats" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p" -o x -- $CMD > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
  F=$(find "$OUT/p" -name "x_kernel_stats.csv" | head -1)
  cp "$F" "$OUT/${NAME}_kernel_stats.csv"
  python3 - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "bbp::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if float(r["TotalDurationNs"]) < 0.002 * tot:
        continue
    print("  %-46s calls %5s avg %9.1f us total %9.2f ms %5.1f %%" % (r["Name"].replace("void ", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                     float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
  tail -1 "$OUT/stdout.txt" | cut -c1-400
else
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/p" -o x -- $CMD > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
  T=$(find "$OUT/p" -name "x_kernel_trace.csv" | head -1)
  python3 $REPO/tools/timeline_stats.py "$T" | tee "$OUT/${NAME}_timeline_stats.txt"
  python3 $REPO/tools/timeline_gaps.py "$T" | tee "$OUT/${NAME}_timeline_gaps.txt"
fi
