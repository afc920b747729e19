# GPU box: bash tools/midsize_trace.sh B   -> kernel-trace analysis of back-to-back B-proof calls
REPO=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-256}; OUT=$REPO/gpurun_out/mtrace$B; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/tools/midsize.py $B > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
grep "B=" $OUT/run.log
python3 $REPO/tools/midsize_trace.py $OUT/t_kernel_trace.csv 9 | tee $REPO/gpurun_out/midsize_trace_$B.txt
