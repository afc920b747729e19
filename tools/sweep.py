#!/usr/bin/env python3
"""Prove / verify throughput over batch size and list length (device-resident inputs): python tools/sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import make_workload, VerifyWorkload

dev = torch.device("cuda", 0)
ctx = bbp.Context(0)
s = None  # the context's own stream (include/bbp.h BBP_STREAM_CONTEXT); callers synchronise the device
for B, N in ((1024, 1), (1024, 8), (1024, 64), (1024, 202), (256, 8), (512, 8), (2048, 8), (4096, 8)):
    pw = make_workload("prove", ctx, bbp, torch, dev, B, N, 1)
    vw = VerifyWorkload(ctx, bbp, torch, dev, B, N, 1, prove_wl=pw)
    res = []
    for w in (pw, vw):
        for _ in range(2):
            w.step(s)
        torch.cuda.synchronize()
        n = 6
        t = time.perf_counter()
        for _ in range(n):
            w.step(s)
        torch.cuda.synchronize()
        res.append(B * n / (time.perf_counter() - t))
    free, total = torch.cuda.mem_get_info()
    print("B=%5d N=%3d  prove %8.0f /s   verify %8.0f /s   device memory in use %.1f GB" % (B, N, res[0], res[1], (total - free) / 2**30), flush=True)
    del pw, vw
