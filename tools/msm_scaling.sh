# GPU box: bash tools/msm_scaling.sh [VARIANT]  -> gpurun_out/msm_scaling[_VARIANT].txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; OUT=$REPO/gpurun_out/mscal$V; rm -rf $OUT; mkdir -p $OUT
[ -n "$V" ] && export BBP_LIB_VARIANT=$V
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/tools/msm_scaling.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 $REPO/tools/msm_scaling.py parse $OUT/t_kernel_trace.csv | tee $REPO/gpurun_out/msm_scaling${V:+_$V}.txt
