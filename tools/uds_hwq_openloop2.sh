# GPU box: the server at and near saturation, open loop, 8 vs 16 hardware queues, two passes
O=${1:-gpurun_out/r3_hwq_openloop2.jsonl}; : > $O
for P in 1 2; do for Q in 8 16; do
python3 tools/uds_bench.py --connections 12288 --no-verify --sweep 18000,20000,21000 --duration 6 --hwq $Q | sed "s/^{/{\"hwq\": $Q, \"what\": \"open prove-only\", /" >> $O
python3 tools/uds_bench.py --connections 12288 --sweep 14000,16000 --duration 6 --hwq $Q | sed "s/^{/{\"hwq\": $Q, \"what\": \"open prove+verify\", /" >> $O
done; done
