#!/bin/bash
# GPU box: kernel statistics of the verification workload.  usage: bash tools/prof_verify.sh [batch=1024]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1024}
OUT=$REPO/gpurun_out/prof_verify
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/v" -o x -- python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --workload verify --batch $B --steps 10 --warmup 2 > "$OUT/v.json" 2> "$OUT/v.log"
python3 - "$OUT/v/x_kernel_stats.csv" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'bbp::' in n and float(r['TotalDurationNs'])>2e5:
        print("  %-44s calls %4s avg %8.1f us total %8.2f ms" % (n[:44].replace('void ',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
python3 -c "
import json; d=json.loads(open('$OUT/v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
