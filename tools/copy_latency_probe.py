#!/usr/bin/env python3
"""How long does a small pinned H2D copy on its own stream take while the engine saturates the GPU?  (host-API pipelining)
    [HSA_ENABLE_SDMA=0|1] python tools/copy_latency_probe.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import make_workload
dev = torch.device("cuda", 0)
ctx = bbp.Context(0)
wl = make_workload("prove", ctx, bbp, torch, dev, 1024, 8, 1)
eng = torch.cuda.ExternalStream(ctx.stream, device=dev)
stop = False
def load():
    with torch.cuda.stream(eng):
        while not stop:
            for _ in range(4):
                wl.step(eng.cuda_stream)
            eng.synchronize()
h = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
d = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
def probe(stream, label):
    lat = []
    for _ in range(30):
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            d.copy_(h, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        while not ev.query():
            time.sleep(0.0001)
        lat.append((time.perf_counter() - t0) * 1e3)
        time.sleep(0.013)
    lat.sort()
    print("%-28s p50 %.2f ms  p90 %.2f ms  max %.2f ms" % (label, lat[15], lat[27], lat[-1]), flush=True)
cs = torch.cuda.ExternalStream(ctx.copy_stream, device=dev)
ts = torch.cuda.Stream()
probe(cs, "idle GPU, ctx copy stream")
t = threading.Thread(target=load); t.start(); time.sleep(0.5)
probe(cs, "busy GPU, ctx copy stream")
probe(ts, "busy GPU, torch stream")
# a tiny kernel instead of a copy
x = torch.zeros(1024, device=dev)
lat = []
for _ in range(30):
    t0 = time.perf_counter()
    with torch.cuda.stream(ts):
        x.add_(1)
        ev = torch.cuda.Event(); ev.record(ts)
    while not ev.query():
        time.sleep(0.0001)
    lat.append((time.perf_counter() - t0) * 1e3); time.sleep(0.013)
lat.sort(); print("%-28s p50 %.2f ms  p90 %.2f ms  max %.2f ms" % ("busy GPU, tiny kernel", lat[15], lat[27], lat[-1]))
stop = True; t.join()
