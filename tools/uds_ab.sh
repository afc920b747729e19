O=gpurun_out/r3_uds_ab.jsonl; : > $O
run() { python3 tools/uds_bench.py --connections $C --ops $((C*32)) $X | sed "s/^{/{\"variant\": \"$1 C=$C $X\", /" >> $O; }
for R in 1 2; do
C=2048; X=""; run "plain"; X="--verify-aggregate 32"; run "agg32"
C=4096; X=""; run "plain"; X="--verify-aggregate 32"; run "agg32"
done
python3 tools/uds_bench.py --connections 16384 --sweep 14000,16000,17000 --duration 8 --verify-aggregate 32 | sed "s/^{/{\"variant\": \"agg32 open\", /" >> $O
