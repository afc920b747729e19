#!/usr/bin/env python3
"""prove_batch_dev + verify_batch_dev back to back on one stream, inputs resident: ms per (prove, verify) pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import make_workload, VerifyWorkload
dev = torch.device("cuda", 0)
ctx = bbp.Context(0)
pw = make_workload("prove", ctx, bbp, torch, dev, 1024, 8, 1)
vw = VerifyWorkload(ctx, bbp, torch, dev, 1024, 8, 1, prove_wl=pw)
s = None  # the context's own stream (include/bbp.h BBP_STREAM_CONTEXT); callers synchronise the device
for mode in ("prove", "verify", "both"):
    for _ in range(3):
        if mode != "verify": pw.step(s)
        if mode != "prove": vw.step(s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 12
    for _ in range(n):
        if mode != "verify": pw.step(s)
        if mode != "prove": vw.step(s)
    torch.cuda.synchronize()
    print("%-7s %.1f ms per iteration" % (mode, (time.perf_counter() - t) / n * 1e3), flush=True)
