set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/vprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --no-exclusive --workload verify --batch 1024 --steps 40 --warmup 4"
BBP_BENCH_VERIFY_LANES=${1:-2} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l2 -o l2 -- $B > $OUT/l2.json 2> $OUT/l2.log
