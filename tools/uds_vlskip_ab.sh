# GPU box: the server (8 hardware queues), prove + verify: verifier lanes 1-3 on the queues of the caller's / opening / slice-1 streams (default
# creation order) vs skipped past the first two (placeholder streams): lanes 1-3 then share the three slice streams' queues
O=${1:-gpurun_out/r3_vlskip.jsonl}; : > $O
for V in "" vlskip2 "" vlskip2; do
BBP_LIB_VARIANT=$V python3 tools/uds_bench.py --connections 8192 --sweep 4000,8000,12000,14000 --duration 5 | sed "s/^{/{\"variant\": \"$V\", \"what\": \"open prove+verify\", /" >> $O
BBP_LIB_VARIANT=$V python3 tools/uds_bench.py --connections 2048 --ops 98304 | sed "s/^{/{\"variant\": \"$V\", \"what\": \"closed 2048 prove+verify\", /" >> $O
done
