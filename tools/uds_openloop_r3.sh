# GPU box: latency vs offered load through the UDS server (open loop, Poisson arrivals): bash tools/uds_openloop_r3.sh OUT.jsonl
O=${1:-gpurun_out/r3_uds_open.jsonl}; : > $O
E=${O%.jsonl}.err; : > $E
python tools/uds_bench.py --connections 16384 --no-verify --sweep 1000,2000,4000,8000,12000,16000,18000,20000 --duration 8 --max-batch 4096 >> $O 2>>$E
python tools/uds_bench.py --connections 16384 --sweep 2000,4000,8000,10000,12000,14000 --duration 8 --max-batch 4096 >> $O 2>>$E
