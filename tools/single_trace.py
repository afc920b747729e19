#!/usr/bin/env python3
"""One-proof calls through the host API, for a rocprofv3 kernel trace of the small-call path: python tools/single_trace.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import dusk_blindbidproof_amd as bbp
from bench_workloads import synth_bids
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = bbp.Context(0)
ins, ents, pubs, qz = synth_bids(ctx, B, 8, 3)
for _ in range(reps):
    out, st = ctx.prove_batch(B, 8, b"".join(ins), b"".join(ents))
ctx.close()
