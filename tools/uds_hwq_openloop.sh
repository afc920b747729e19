# GPU box: the server's latency against offered load with 8 and 16 hardware queues (open loop; five prover buffers)
O=${1:-gpurun_out/r3_hwq_openloop.jsonl}; : > $O
for Q in 8 16; do
python3 tools/uds_bench.py --connections 8192 --no-verify --sweep 1000,4000,8000,12000,16000 --duration 6 --hwq $Q | sed "s/^{/{\"hwq\": $Q, \"what\": \"open prove-only\", /" >> $O
python3 tools/uds_bench.py --connections 8192 --sweep 4000,8000,12000 --duration 6 --hwq $Q | sed "s/^{/{\"hwq\": $Q, \"what\": \"open prove+verify\", /" >> $O
done
