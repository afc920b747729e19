# GPU box: server with 8 vs 16 hardware queues under the final combiner rules: prove-only and prove+verify, closed loop, alternating
O=${1:-gpurun_out/r3_hwq_ab2.jsonl}; : > $O
for P in 1 2 3; do for Q in 8 16; do
python3 tools/uds_bench.py --connections 3072 --no-verify --ops 110592 --hwq $Q | sed "s/^{/{\"hwq\": $Q, \"what\": \"closed 3072 prove-only\", /" >> $O
python3 tools/uds_bench.py --connections 2048 --ops 98304 --hwq $Q | sed "s/^{/{\"hwq\": $Q, \"what\": \"closed 2048 prove+verify\", /" >> $O
done; done
