# GPU box: is the server's 16-queue slowdown host-side?  Fewer I/O and generator threads, closed loop 3072 connections prove-only
O=${1:-gpurun_out/r3_hwq_threads.jsonl}; : > $O
for P in 1 2; do for CFG in "16 2 2" "16 1 1" "8 1 1"; do set -- $CFG
python3 tools/uds_bench.py --connections 3072 --no-verify --ops 110592 --hwq $1 --io-threads $2 --gen-threads $3 | sed "s/^{/{\"hwq\": $1, \"io\": $2, \"gen\": $3, /" >> $O
done; done
