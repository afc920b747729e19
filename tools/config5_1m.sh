# GPU box: BASELINE configs[4] at its stated volume on ONE GPU: a million synthetic bids streamed (a) through the device API
# (bench.py --workload stream: host ingest -> witness -> prove -> verify -> D2H, chunks of 1024), (b) through the UDS server under
# open-loop Poisson load (prove then verify per op).
O=${1:-gpurun_out/r3_config5}
python3 bench.py --workload stream --steps 980 --warmup 4 --no-cpu-baseline --no-exclusive > ${O}_stream_1M.json 2> ${O}_stream_1M.err
python3 tools/uds_bench.py --connections 16384 --rate 15000 --duration 70 > ${O}_uds_1M.json 2> ${O}_uds_1M.err
python3 tools/uds_bench.py --connections 16384 --rate 19000 --duration 55 --no-verify > ${O}_uds_prove_1M.json 2>> ${O}_uds_1M.err
