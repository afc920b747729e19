# GPU box: configs[4] through the UDS server, closed-loop points (bash tools/uds_sweep_r3.sh OUT.jsonl)
O=${1:-gpurun_out/r3_uds.jsonl}; : > $O
E=${O%.jsonl}.err; : > $E
python tools/uds_bench.py --connections 1 --ops 64 >> $O 2>>$E
python tools/uds_bench.py --connections 64 --ops 4096 >> $O 2>>$E
python tools/uds_bench.py --connections 256 --ops 8192 >> $O 2>>$E
python tools/uds_bench.py --connections 1024 --ops 32768 >> $O 2>>$E
python tools/uds_bench.py --connections 2048 --ops 65536 >> $O 2>>$E
python tools/uds_bench.py --connections 4096 --ops 98304 >> $O 2>>$E
python tools/uds_bench.py --connections 3072 --ops 98304 --no-verify >> $O 2>>$E
python tools/uds_bench.py --connections 4096 --ops 131072 --no-verify >> $O 2>>$E
python tools/uds_bench.py --connections 16384 --no-verify --sweep 21000,22000 --duration 8 >> $O 2>>$E
python tools/uds_bench.py --connections 16384 --sweep 15000,16000 --duration 8 >> $O 2>>$E
