#!/usr/bin/env python3
"""Latency of small host-pointer prove calls when several run at once (what the combiner's batch threads do at low load):
python tools/small_concurrent.py [B] -- T threads each calling bbp_prove_batch(B) back to back; mean / p90 latency per call."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import dusk_blindbidproof_amd as bbp
from bench_workloads import synth_bids
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = bbp.Context(0)
N = 8
ins, ents, pubs, qz = synth_bids(ctx, B, N, seed=5)
blob = b"".join(ins)
for _ in range(4):
    ctx.prove_batch(B, N, blob, None)
for T in (1, 2, 3, 4):
    lat = []
    lock = threading.Lock()
    def work():
        mine = []
        for _ in range(40):
            t = time.perf_counter()
            ctx.prove_batch(B, N, blob, None)
            mine.append(time.perf_counter() - t)
        with lock:
            lat.extend(mine)
    th = [threading.Thread(target=work) for _ in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    lat.sort()
    print("B=%d threads %d: %.1f ms mean, %.1f ms p90 per call; %.0f calls/s" % (B, T, 1e3 * sum(lat) / len(lat), 1e3 * lat[int(0.9 * len(lat))], len(lat) / dt), flush=True)
