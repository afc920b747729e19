# GPU box: knobs of the rotating schedule at 1024 proofs per call, interleaved on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2))'
B="python3 $REPO/bench.py --no-cpu-baseline --no-also --no-exclusive --steps 24 --warmup 6"
for R in 1 2; do
$B 2>/dev/null | python3 -c "$P" "round $R default"
BBP_SERIAL_BLOCK=128 $B 2>/dev/null | python3 -c "$P" "round $R serial_block=128"
BBP_SERIAL_BLOCK=256 $B 2>/dev/null | python3 -c "$P" "round $R serial_block=256"
BBP_FOLD_HALF_FROM=4096 $B 2>/dev/null | python3 -c "$P" "round $R fold128"
done
