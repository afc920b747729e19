# GPU box: the server (8 hardware queues) with the second opening stream created before the verifier lanes (own queue) vs lazily (shares lane 1's)
O=${1:-gpurun_out/r3_side2.jsonl}; : > $O
for V in "" side2 "" side2; do
BBP_LIB_VARIANT=$V python3 tools/uds_bench.py --connections 8192 --no-verify --sweep 4000,8000,12000,16000 --duration 5 | sed "s/^{/{\"variant\": \"$V\", \"what\": \"open prove-only\", /" >> $O
BBP_LIB_VARIANT=$V python3 tools/uds_bench.py --connections 8192 --sweep 8000,14000 --duration 5 | sed "s/^{/{\"variant\": \"$V\", \"what\": \"open prove+verify\", /" >> $O
BBP_LIB_VARIANT=$V python3 tools/uds_bench.py --connections 3072 --no-verify --ops 110592 | sed "s/^{/{\"variant\": \"$V\", \"what\": \"closed 3072 prove-only\", /" >> $O
done
