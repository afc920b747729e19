# GPU box: bench step and the stream workload with the deep-pipeline rotation at thresholds 3 (default) / 2 / off, interleaved on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), d.get("chunk_latency_ms",{}).get("p50",""))'
for R in 1 2 3; do
python3 $REPO/bench.py --no-cpu-baseline --no-also --no-exclusive 2>/dev/null | python3 -c "$P" "round $R prove deep_from=3"
BBP_ROTATE_DEEP_FROM=2 python3 $REPO/bench.py --no-cpu-baseline --no-also --no-exclusive 2>/dev/null | python3 -c "$P" "round $R prove deep_from=2"
BBP_ROTATE_DEEP_MAX=0 python3 $REPO/bench.py --no-cpu-baseline --no-also --no-exclusive 2>/dev/null | python3 -c "$P" "round $R prove off"
done
for R in 1 2; do
python3 $REPO/bench.py --workload stream --no-cpu-baseline --no-also 2>/dev/null | python3 -c "$P" "round $R stream deep_from=3"
BBP_ROTATE_DEEP_MAX=0 python3 $REPO/bench.py --workload stream --no-cpu-baseline --no-also 2>/dev/null | python3 -c "$P" "round $R stream off"
done
