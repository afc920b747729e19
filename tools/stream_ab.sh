REPO=${GRAFT_REPO_ROOT:-/root/repo}
B="python3 $REPO/bench.py --no-cpu-baseline --no-build --no-exclusive --workload stream"
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), d["chunk_latency_ms"]["p50"], d["chunk_latency_ms"]["p99"], d["failed_verifications"])'
for R in 1 2; do
BBP_BENCH_STREAM_VERIFY_INLINE=1 $B --steps 64 --warmup 6 | python3 -c "$P" "inline verify, chunks of 1024"
$B --steps 64 --warmup 6 | python3 -c "$P" "verify on lanes, chunks of 1024"
done
BBP_BENCH_STREAM_VERIFY_INLINE=1 $B --batch 2048 --steps 32 --warmup 4 | python3 -c "$P" "inline verify, chunks of 2048"
$B --batch 2048 --steps 32 --warmup 4 | python3 -c "$P" "verify on lanes, chunks of 2048"
