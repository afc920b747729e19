#!/usr/bin/env python3
"""GPU box: which of the engine's streams lands on which HARDWARE QUEUE inside bbp-uds-server, and what that does to a closed-loop
prove-only run.  Starts the server under `rocprofv3 --kernel-trace` (the server binary itself follows `--`), drives it with
bbp-uds-loadgen, stops it, and prints from the trace: Stream_Id -> Queue_Id, launches, busy time and the kernels that identify the
stream (k_open_serial = an opening stream, k_msm_acc = a heavy-stage stream, k_vtranscript = a verifier lane ...), then per queue
the streams that share it.

    python tools/uds_queue_map.py --hwq 8  [--connections 3072] [--ops 49152] [--verify]
    python tools/uds_queue_map.py --hwq 16
"""
import argparse, collections, csv, glob, json, os, signal, struct, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from tests import uds_client as uc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hwq", type=int, default=8)
    ap.add_argument("--connections", type=int, default=3072)
    ap.add_argument("--ops", type=int, default=49152)
    ap.add_argument("--verify", action="store_true", help="prove then verify per connection (default: prove only)")
    ap.add_argument("--items", type=int, default=8)
    a = ap.parse_args()
    ge.build_server()
    import torch  # noqa: F401
    import dusk_blindbidproof_amd as bbp
    from bench_workloads import synth_bids
    N = a.items
    ctx = bbp.Context(0)
    ins, _, pubs, qz = synth_bids(ctx, 256, N, seed=77)
    ctx.close()
    d = tempfile.mkdtemp(prefix="bbp-qmap-")
    sock, reqs, out = os.path.join(d, "sock"), os.path.join(d, "bids.bin"), os.path.join(d, "trace")
    with open(reqs, "wb") as f:
        for i in range(256):
            s7, pub, toggle = ins[i][:224], pubs[i], int.from_bytes(ins[i][-8:], "little")
            pf = uc.prove_request(s7, pub, toggle)
            vt = uc.tlv(qz[i][:32]) + uc.tlv(qz[i][32:64]) + uc.tlv(qz[i][64:96]) + uc.tlv_list([pub[32 * j:32 * j + 32] for j in range(N)])
            f.write(struct.pack("<I", len(pf)) + pf + struct.pack("<I", len(vt)) + vt)
    env = dict(os.environ, GPU_MAX_HW_QUEUES=str(a.hwq), TMPDIR="/tmp")
    log = open(os.path.join(d, "server.log"), "w+")
    srv = subprocess.Popen(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "srv", "--",
                            ge.SERVER_BIN, "-b", sock, "-l", "info", "--engine", bbp.lib_path, "--window-us", "300", "--max-batch", "4096",
                            "--max-connections", str(max(4096, 2 * a.connections)), "--io-threads", "2", "--reserve", str(N)], stderr=log, env=env, cwd="/tmp",
                           start_new_session=True)  # own process group: the launcher starts the server as ITS child, SIGTERM must reach both
    for _ in range(9000):
        if os.path.exists(sock) or srv.poll() is not None:
            break
        time.sleep(0.02)
    if not os.path.exists(sock):
        log.seek(0)
        sys.exit("server did not come up: " + log.read()[-800:])
    cmd = [ge.LOADGEN_BIN, "--socket", sock, "--requests", reqs, "--connections", str(a.connections), "--threads", "2"] + ([] if a.verify else ["--no-verify"])
    subprocess.run(cmd + ["--ops", str(2 * a.connections)], capture_output=True, text=True)
    run = subprocess.run(cmd + ["--ops", str(a.ops)], capture_output=True, text=True)
    os.killpg(srv.pid, signal.SIGTERM)  # exactly the group started above: the server drains and exits, the tool writes its trace at exit
    try:
        srv.wait(timeout=120)
    except subprocess.TimeoutExpired:
        os.killpg(srv.pid, signal.SIGKILL)
        srv.wait(timeout=30)
    for _ in range(200):  # the server process may outlive the launcher by the time its trace takes to be written
        if glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
            break
        time.sleep(0.1)
    time.sleep(1.0)
    res = json.loads(run.stdout.strip().splitlines()[-1]) if run.stdout.strip() else {"error": run.stderr[-300:]}
    print("GPU_MAX_HW_QUEUES=%d: %s proofs/s, prove p50 %s ms" % (a.hwq, res.get("proofs_per_s"), (res.get("prove_latency_ms") or {}).get("p50")))
    traces = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
    if not traces:
        log.seek(0)
        sys.exit("no kernel trace: " + log.read()[-800:])
    per_stream = collections.defaultdict(lambda: {"q": collections.Counter(), "n": 0, "busy": 0, "k": collections.Counter()})
    for r in csv.DictReader(open(traces[0])):
        s = per_stream[r["Stream_Id"]]
        s["q"][r["Queue_Id"]] += 1
        s["n"] += 1
        s["busy"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        s["k"][r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bbp::", "")[:28]] += 1
    by_queue = collections.defaultdict(list)
    print("%-8s %-14s %8s %10s  %s" % ("stream", "queue(s)", "launches", "busy ms", "most frequent kernels"))
    for sid, s in sorted(per_stream.items(), key=lambda kv: int(kv[0])):
        qs = ",".join("%s" % q for q, _ in s["q"].most_common())
        for q in s["q"]:
            by_queue[q].append(sid)
        print("%-8s %-14s %8d %10.1f  %s" % (sid, qs, s["n"], s["busy"] / 1e6, ", ".join("%s x%d" % kv for kv in s["k"].most_common(3))))
    print("queues shared by several streams:", {q: v for q, v in sorted(by_queue.items(), key=lambda kv: int(kv[0])) if len(v) > 1} or "none")


if __name__ == "__main__":
    main()
