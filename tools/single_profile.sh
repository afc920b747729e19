set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sprof; rm -rf $OUT; mkdir -p $OUT
python3 $REPO/tools/latency.py > $OUT/latency.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/one -o one -- python3 $REPO/tools/single_trace.py 1 6 > $OUT/one.log 2>&1
