REPO=${GRAFT_REPO_ROOT:-/root/repo}
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --no-exclusive"
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],3))'
run() { "${@:2}" $B --workload verify --batch 1024 --steps 200 --warmup 10 | python3 -c "$P" "$1"; }
export BBP_VERIFY_SERIAL_ACC=0 BBP_VERIFY_OVERLAP=1
for Q in 16 32; do for L in 4 5 6; do
run "hwq=$Q vl6 lanes=$L" env GPU_MAX_HW_QUEUES=$Q BBP_LIB_VARIANT=vl6 BBP_BENCH_VERIFY_LANES=$L
done; done
run8() { "${@:2}" $B --workload $W --batch 8192 --steps 16 --warmup 4 | python3 -c "$P" "$1"; }
for W in verify verify_aggregated; do for L in 2 3 4; do
run8 "$W B=8192 hwq=16 lanes=$L" env GPU_MAX_HW_QUEUES=16 BBP_LIB_VARIANT=vl4 BBP_BENCH_VERIFY_LANES=$L
done; done
W=verify_aggregated; for L in 2 4; do GPU_MAX_HW_QUEUES=16 BBP_LIB_VARIANT=vl4 BBP_BENCH_VERIFY_LANES=$L $B --workload verify_aggregated --batch 1024 --steps 200 --warmup 10 | python3 -c "$P" "agg B=1024 lanes=$L"; done
for S in 3 4; do for Q in 8 16 32; do GPU_MAX_HW_QUEUES=$Q BBP_SLICES=$S $B --steps 24 --warmup 4 | python3 -c "$P" "prove slices=$S hwq=$Q"; done; done
