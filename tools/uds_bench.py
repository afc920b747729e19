#!/usr/bin/env python3
"""BASELINE.json configs[4] measured THROUGH the reference's IPC surface: starts bbp-uds-server on a temp socket, writes K
pre-encoded synthetic bids, runs bbp-uds-loadgen with C closed-loop connections (one op = prove then verify, the Go
BenchmarkProveVerify's shape), prints the load generator's JSON line plus the server's batching statistics.

    python tools/uds_bench.py [--connections 2048] [--ops 16384] [--items 8] [--window-us 300] [--max-batch 1024] [--stub]
    python tools/uds_bench.py --rate 12000 --duration 10 [--connections 8192]     # open loop: Poisson arrivals at a fixed offered load
    python tools/uds_bench.py --sweep 4000,8000,12000,16000,18000 --duration 8     # latency-vs-offered-load table (one JSON line per point)
"""
import argparse, json, os, re, signal, struct, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from tests import uds_client as uc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--connections", type=int, default=2048)
    ap.add_argument("--ops", type=int, default=16384)
    ap.add_argument("--items", type=int, default=8)
    ap.add_argument("--distinct", type=int, default=256)
    ap.add_argument("--window-us", type=int, default=300)
    ap.add_argument("--max-batch", type=int, default=4096)
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--rate", type=float, default=0.0, help="open loop: offered ops per second (Poisson arrivals)")
    ap.add_argument("--duration", type=float, default=10.0, help="open loop: seconds of arrivals")
    ap.add_argument("--sweep", default="", help="open loop: comma-separated offered loads, one run each against the same server")
    ap.add_argument("--io-threads", type=int, default=2)
    ap.add_argument("--gen-threads", type=int, default=2)
    ap.add_argument("--preconnect", type=int, default=0, help="open loop: connections opened before the clock starts")
    ap.add_argument("--verify-aggregate", type=int, default=0, help="server --verify-aggregate G (0 = one MSM per proof like the reference)")
    ap.add_argument("--hwq", type=int, default=int(os.environ.get("BBP_SERVER_HWQ", "8")), help="GPU_MAX_HW_QUEUES of the server process")
    ap.add_argument("--devices", default="0", help="server --devices list (a,b,..: device pool)")
    ap.add_argument("--stub", action="store_true", help="CPU box: the tests' stub engine (plumbing check, not a measurement)")
    a = ap.parse_args()
    ge.build_server()
    d = tempfile.mkdtemp(prefix="bbp-uds-bench-")
    sock, reqs = os.path.join(d, "sock"), os.path.join(d, "bids.bin")
    N = a.items
    if a.stub:
        engine = ge.build_stub_engine()
        bids = []
        for i in range(a.distinct):
            s7 = b"".join(bytes([(7 * i + k) & 0xff]) * 31 + b"\x01" for k in range(7))
            pub = b"".join(bytes([(11 * i + j) & 0xff]) * 31 + b"\x02" for j in range(N))
            bids.append((s7, pub, i % N, s7[128:160] + s7[160:192] + s7[192:224]))
    else:
        import torch  # noqa: F401  (its HIP runtime first, see tests/conftest.py)
        import dusk_blindbidproof_amd as bbp
        from bench_workloads import synth_bids
        engine = bbp.lib_path
        ctx = bbp.Context(0)
        ins, _, pubs, qz = synth_bids(ctx, a.distinct, N, seed=77)
        ctx.close()
        bids = [(ins[i][:224], pubs[i], int.from_bytes(ins[i][-8:], "little"), qz[i]) for i in range(a.distinct)]
    with open(reqs, "wb") as f:
        for s7, pub, toggle, qzs in bids:
            pf = uc.prove_request(s7, pub, toggle)
            vt = uc.tlv(qzs[:32]) + uc.tlv(qzs[32:64]) + uc.tlv(qzs[64:96]) + uc.tlv_list([pub[32 * j:32 * j + 32] for j in range(N)])
            f.write(struct.pack("<I", len(pf)) + pf + struct.pack("<I", len(vt)) + vt)
    log = open(os.path.join(d, "server.log"), "w+")
    srv = subprocess.Popen([ge.SERVER_BIN, "-b", sock, "-l", "info", "--engine", engine, "--window-us", str(a.window_us), "--max-batch",
                            str(a.max_batch), "--max-connections", str(max(4096, 2 * a.connections)), "--io-threads", str(a.io_threads),
                            "--devices", a.devices, "--reserve", str(N)] + (["--verify-aggregate", str(a.verify_aggregate)] if a.verify_aggregate else []), stderr=log, env=dict(os.environ, GPU_MAX_HW_QUEUES=str(a.hwq)))
    for _ in range(6000):
        if os.path.exists(sock) or srv.poll() is not None:
            break
        time.sleep(0.02)
    if not os.path.exists(sock):
        log.seek(0)
        sys.exit("server did not come up: " + log.read()[-500:])
    cmd = [ge.LOADGEN_BIN, "--socket", sock, "--requests", reqs, "--connections", str(a.connections), "--threads", str(a.gen_threads)]
    extra = ["--no-verify"] if a.no_verify else []
    warm = subprocess.run(cmd + ["--ops", str(min(a.ops, 2 * a.connections))], capture_output=True, text=True)  # compile the circuit, grow buffers
    runs = []
    if a.sweep or a.rate > 0:
        for r in ([float(x) for x in a.sweep.split(",")] if a.sweep else [a.rate]):
            runs.append(subprocess.run(cmd + ["--rate", str(r), "--duration", str(a.duration), "--preconnect", str(a.preconnect)] + extra,
                                       capture_output=True, text=True))
    else:
        runs.append(subprocess.run(cmd + ["--ops", str(a.ops)] + extra, capture_output=True, text=True))
    srv.send_signal(signal.SIGTERM)
    srv.wait(timeout=60)
    log.seek(0)
    if os.environ.get("BBP_TRACE"):
        sys.stderr.write("".join([ln for ln in log.read().splitlines(True) if "trace" in ln][:int(os.environ.get("BBP_TRACE_HEAD", "0")) or None][-int(os.environ.get("BBP_TRACE_TAIL", "40")):]))
        log.seek(0)
    text = log.read()
    m = re.search(r"served (\d+) requests \((\d+) errors\) in (\d+) device calls, largest batch (\d+)", text)
    for run in runs:
        out = json.loads(run.stdout.strip().splitlines()[-1]) if run.stdout.strip() else {"error": run.stderr[-300:], "warmup": warm.stderr[-300:]}
        out.update(workload="configs[4] through the UDS server: %s per connection" % ("prove only" if a.no_verify else "prove then verify"), bid_list_len=N,
                   window_us=a.window_us, max_batch=a.max_batch, io_threads=a.io_threads, devices=a.devices, server_hw_queues=a.hwq, verify_aggregate=a.verify_aggregate,
                   engine="stub (not a measurement)" if a.stub else "libbbp_hip.so")
        if m and len(runs) == 1:
            out["server"] = dict(requests=int(m.group(1)), errors=int(m.group(2)), device_calls=int(m.group(3)), largest_batch=int(m.group(4)))
        print(json.dumps(out), flush=True)
    if "ERROR" in text and os.environ.get("BBP_SHOW_SERVER_ERRORS"):
        sys.stderr.write(text[-2000:])


if __name__ == "__main__":
    main()
