#!/usr/bin/env python3
"""Latency of small calls through the host-pointer API (what a one-request-at-a-time caller of Proof::prove / Verify::verify sees):
python tools/latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (its HIP runtime first)
import dusk_blindbidproof_amd as bbp
from bench_workloads import synth_bids

ctx = bbp.Context(0)
for B in (1, 8, 64, 256):
    N = 8
    ins, ents, pubs, qz = synth_bids(ctx, B, N, 3)
    rs = bbp.record_size(N)
    ts = []
    for it in range(7):
        t = time.perf_counter()
        out, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
        ts.append(time.perf_counter() - t)
    assert st == [0] * B
    vin = b"".join(out[i * rs:(i + 1) * rs] + qz[i] + pubs[i] for i in range(B))
    tv = []
    for it in range(7):
        t = time.perf_counter()
        sv = ctx.verify_batch(B, N, vin)
        tv.append(time.perf_counter() - t)
    assert sv == [0] * B
    print("B=%4d  prove %.1f ms (min %.1f)   verify %.1f ms (min %.1f)" % (B, sorted(ts)[3] * 1e3, min(ts) * 1e3, sorted(tv)[3] * 1e3, min(tv) * 1e3), flush=True)
