#!/usr/bin/env python3
"""Per-launch HIP-event timings of one prove / verify step (B proofs): python tools/prof_step.py [B] [N] [workload]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import make_workload, TAG_NAMES

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
name = sys.argv[3] if len(sys.argv) > 3 else "prove"
dev = torch.device("cuda", 0)
ctx = bbp.Context(0)
wl = make_workload(name, ctx, bbp, torch, dev, B, N, 1)
s = None  # the context's own stream (include/bbp.h BBP_STREAM_CONTEXT); callers synchronise the device
wl.step(s)
torch.cuda.synchronize()
ctx.set_profiling(True)
ctx.last_timings()
wl.step(s)
t = ctx.last_timings()
tot = sum(us for _, us in t)
print("total kernel time %.1f ms for B=%d (%s)" % (tot / 1e3, B, name))
for tag, us in t:
    print("%-24s %10.1f us" % (TAG_NAMES.get(tag, tag), us))
