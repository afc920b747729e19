# GPU box: the server's prove-only rate against GPU_MAX_HW_QUEUES of the server process
O=${1:-gpurun_out/r3_hwq_sweep.jsonl}; : > $O
for Q in 8 10 12 16 8; do
python3 tools/uds_bench.py --connections 3072 --no-verify --ops 110592 --hwq $Q | sed "s/^{/{\"hwq\": $Q, /" >> $O
done
