O=gpurun_out/r3_uds_ab.jsonl; : > $O
run() { "${@:2}" python3 tools/uds_bench.py --connections $C --ops $((C*32)) $X | sed "s/^{/{\"variant\": \"$1 C=$C $X\", /" >> $O; }
for R in 1 2; do
C=2048; X=""; run "hwq=8" env GPU_MAX_HW_QUEUES=8; run "hwq=16" env GPU_MAX_HW_QUEUES=16; run "hwq=12" env GPU_MAX_HW_QUEUES=12
C=4096; X=""; run "hwq=8" env GPU_MAX_HW_QUEUES=8; run "hwq=16" env GPU_MAX_HW_QUEUES=16
C=3072; X="--no-verify"; run "hwq=12" env GPU_MAX_HW_QUEUES=12; run "hwq=10" env GPU_MAX_HW_QUEUES=10
done
