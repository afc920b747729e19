# GPU box: the other server workloads with the final combiner rules (compare with profiles/r03_uds_prove_leaders_ab.jsonl, leaders = 2)
O=${1:-gpurun_out/r3_check_b.jsonl}; : > $O
for P in 1 2; do
python3 tools/uds_bench.py --connections 2048 --ops 98304 | sed "s/^{/{\"what\": \"closed 2048 prove+verify\", /" >> $O
python3 tools/uds_bench.py --connections 8192 --sweep 12000,15000 --duration 6 | sed "s/^{/{\"what\": \"open prove+verify\", /" >> $O
python3 tools/uds_bench.py --connections 8192 --no-verify --sweep 8000,16000,20000 --duration 6 | sed "s/^{/{\"what\": \"open prove-only\", /" >> $O
done
