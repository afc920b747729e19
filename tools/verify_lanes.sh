REPO=${GRAFT_REPO_ROOT:-/root/repo}
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --no-exclusive"
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],3))'
run() { "${@:2}" $B --workload verify --batch 1024 --steps 200 --warmup 10 | python3 -c "$P" "$1"; }
for R in 1 2; do
run "round $R chain after acc, 2 lanes" env BBP_VERIFY_OVERLAP=0 BBP_BENCH_VERIFY_LANES=2
run "round $R chain after fold, 2 lanes" env BBP_VERIFY_CHAIN_AFTER_FOLD=1 BBP_VERIFY_OVERLAP=0 BBP_BENCH_VERIFY_LANES=2
run "round $R chain after acc, 3 lanes" env BBP_VERIFY_OVERLAP=0 BBP_BENCH_VERIFY_LANES=3
run "round $R chain after fold, 3 lanes" env BBP_VERIFY_CHAIN_AFTER_FOLD=1 BBP_VERIFY_OVERLAP=0 BBP_BENCH_VERIFY_LANES=3
done
