# GPU box: host-pointer prove path vs hardware queue count (two host threads, B = 1536 and 1024)
for Q in 8 16 12 8 16; do for B in 1536; do
echo "hwq=$Q B=$B: $(GPU_MAX_HW_QUEUES=$Q python3 tools/host_pipeline.py --batch $B --threads 2 --iters 14 2>/dev/null | tail -1)"
done; done
echo "hwq=16 poll=50: $(GPU_MAX_HW_QUEUES=16 BBP_WAIT_POLL_US=50 python3 tools/host_pipeline.py --batch 1536 --threads 2 --iters 14 2>/dev/null | tail -1)"
echo "hwq=8 poll=50: $(GPU_MAX_HW_QUEUES=8 BBP_WAIT_POLL_US=50 python3 tools/host_pipeline.py --batch 1536 --threads 2 --iters 14 2>/dev/null | tail -1)"
