# GPU box: closed loop, 3072 connections prove-only, six runs each with and without the lopsided-pair wait (each run falls into one of two states)
O=${1:-gpurun_out/r3_lopsided.jsonl}; : > $O
for P in 1 2 3 4 5 6; do for W in 1 0; do
BBP_BATCH_LOPSIDED_WAIT=$W python3 tools/uds_bench.py --connections 3072 --no-verify --ops 110592 | sed "s/^{/{\"lopsided_wait\": $W, /" >> $O
done; done
