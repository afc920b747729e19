# GPU box: prove batches in flight (combiner leaders) 2 vs 3 at low offered loads, five prover buffers
O=${1:-gpurun_out/r3_leaders_low.jsonl}; : > $O
for L in 2 3; do
BBP_BATCH_PROVE_LEADERS=$L BBP_BATCH_STAGGER_SMALL_US=5000 python3 tools/uds_bench.py --connections 4096 --no-verify --sweep 250,1000,2000,4000,8000 --duration 6 | sed "s/^{/{\"prove_leaders\": $L, \"small_stagger_us\": 5000, /" >> $O
done
BBP_BATCH_PROVE_LEADERS=2 python3 tools/uds_bench.py --connections 4096 --no-verify --sweep 250,1000,2000,4000,8000 --duration 6 | sed "s/^{/{\"prove_leaders\": 2, \"small_stagger_us\": 15000, /" >> $O
