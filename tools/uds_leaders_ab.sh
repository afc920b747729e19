# GPU box: prove batches in flight per device (combiner leaders) 2 vs 3: low offered loads and saturation
O=${1:-gpurun_out/r3_leaders.jsonl}; : > $O
for L in 2 3; do
BBP_BATCH_PROVE_LEADERS=$L python3 tools/uds_bench.py --connections 4096 --no-verify --sweep 250,1000,4000,8000,12000 --duration 6 | sed "s/^{/{\"prove_leaders\": $L, /" >> $O
BBP_BATCH_PROVE_LEADERS=$L python3 tools/uds_bench.py --connections 3072 --no-verify --ops 110592 | sed "s/^{/{\"prove_leaders\": $L, /" >> $O
done
