#!/usr/bin/env python3
"""Does an extra stream used before the engine (as RCCL's is in multi-rank runs) cost throughput?  EXTRA_STREAMS=n [GPU_MAX_HW_QUEUES=8] python tools/hwq_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import make_workload
dev = torch.device("cuda", 0)
# simulate what a multi-rank run does before the engine starts: another library's stream (RCCL's) has already been used
extra = [torch.cuda.Stream() for _ in range(int(os.environ.get("EXTRA_STREAMS", "1")))]
for st in extra:
    with torch.cuda.stream(st):
        torch.zeros(1024, device=dev).add_(1)
if os.environ.get("USE_NCCL"):  # the real thing: a one-rank RCCL group, barrier + all-reduce like bench.py's multi-rank prologue
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=dev)
    dist.barrier()
    t1 = torch.ones(8, device=dev); dist.all_reduce(t1)
torch.cuda.synchronize()
ctx = bbp.Context(0)
pw = make_workload("prove", ctx, bbp, torch, dev, 1024, 8, 1)
s = None  # the context's own stream (include/bbp.h BBP_STREAM_CONTEXT); callers synchronise the device
for _ in range(3): pw.step(s)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(12): pw.step(s)
torch.cuda.synchronize()
print("GPU_MAX_HW_QUEUES=%s extra=%s: %.1f ms per batch" % (os.environ.get("GPU_MAX_HW_QUEUES"), os.environ.get("EXTRA_STREAMS", "1"), (time.perf_counter() - t) / 12 * 1e3), flush=True)
