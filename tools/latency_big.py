#!/usr/bin/env python3
"""Latency of ONE isolated large prove call through the host-pointer API: python tools/latency_big.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import dusk_blindbidproof_amd as bbp
from bench_workloads import synth_bids

ctx = bbp.Context(0)
N = 8
ins, ents, pubs, qz = synth_bids(ctx, 256, N, 3)
for B in [int(x) for x in sys.argv[1:]] or [512, 1024, 2048]:
    bi, be = b"".join(ins[i % 256] for i in range(B)), b"".join(ents[i % 256] for i in range(B))
    ts = []
    for it in range(6):
        t = time.perf_counter()
        out, st = ctx.prove_batch(B, N, bi, be)
        ts.append(time.perf_counter() - t)
        time.sleep(0.05)
    assert st == [0] * B
    print("B=%5d  isolated prove call %.1f ms (min %.1f)" % (B, sorted(ts[1:])[2] * 1e3, min(ts[1:]) * 1e3), flush=True)
