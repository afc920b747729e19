# GPU box: combiner prove leaders 2 vs 3 at and near saturation, prove-only and prove+verify, two passes
O=${1:-gpurun_out/r3_leaders2.jsonl}; : > $O
for P in 1 2; do for L in 2 3; do
BBP_BATCH_PROVE_LEADERS=$L python3 tools/uds_bench.py --connections 3072 --no-verify --ops 110592 | sed "s/^{/{\"prove_leaders\": $L, \"what\": \"closed 3072 prove-only\", /" >> $O
BBP_BATCH_PROVE_LEADERS=$L python3 tools/uds_bench.py --connections 2048 --ops 98304 | sed "s/^{/{\"prove_leaders\": $L, \"what\": \"closed 2048 prove+verify\", /" >> $O
BBP_BATCH_PROVE_LEADERS=$L python3 tools/uds_bench.py --connections 8192 --sweep 12000,15000 --duration 6 | sed "s/^{/{\"prove_leaders\": $L, \"what\": \"open prove+verify\", /" >> $O
BBP_BATCH_PROVE_LEADERS=$L python3 tools/uds_bench.py --connections 8192 --no-verify --sweep 16000,20000 --duration 6 | sed "s/^{/{\"prove_leaders\": $L, \"what\": \"open prove-only\", /" >> $O
done; done
