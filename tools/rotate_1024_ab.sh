# GPU box: the 1024-proof bench step on the sliced path (default) and on the small-batch path (unsliced chains in rotation, five buffers)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --no-exclusive --steps 24 --warmup 6"
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2))'
for R in 1 2; do
$B | python3 -c "$P" "round $R sliced(default)"
BBP_ROTATE_BELOW=1024 BBP_DUAL_OPEN_BELOW=1025 $B | python3 -c "$P" "round $R rotate<=1024"
BBP_ROTATE_BELOW=1024 BBP_DUAL_OPEN_BELOW=1025 BBP_RNG_COOP_BELOW=1100 $B | python3 -c "$P" "round $R rotate<=1024+coop"
done
BBP_ROTATE_BELOW=2048 BBP_DUAL_OPEN_BELOW=2049 $B --batch 2048 | python3 -c "$P" "B=2048 rotate"
$B --batch 2048 | python3 -c "$P" "B=2048 sliced"
