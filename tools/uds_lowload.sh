# GPU box: low offered loads through the UDS server: stagger behind small batches 15 ms (default) vs 35 ms
O=${1:-gpurun_out/r3_lowload.jsonl}; : > $O
for S in ${STAGGERS:-15000 35000 8000}; do
BBP_BATCH_STAGGER_SMALL_US=$S python3 tools/uds_bench.py --connections 4096 --no-verify --sweep 250,500,1000,2000,4000,8000 --duration 6 | sed "s/^{/{\"small_stagger_us\": $S, /" >> $O
done
