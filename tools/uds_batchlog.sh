# GPU box: closed loop 3072 connections prove-only with the combiner's batch log, 16 and 8 hardware queues alternating
mkdir -p gpurun_out/blog; rm -f gpurun_out/blog/*
for P in 1 2 3; do for Q in 16 8; do
BBP_BATCH_LOG=$PWD/gpurun_out/blog/q${Q}_$P.log python3 tools/uds_bench.py --connections 3072 --no-verify --ops 110592 --hwq $Q | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('hwq $Q pass $P:', round(d['proofs_per_s']), d['prove_latency_ms'])"
python3 tools/batch_log.py gpurun_out/blog/q${Q}_$P.log
done; done
