import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench import synth_scalars_device
dev = torch.device("cuda", 0)
ctx = bbp.Context(0)
s = None  # the context's own stream (include/bbp.h BBP_STREAM_CONTEXT); callers synchronise the device
n = 2049
for B in (64, 256, 512, 1024, 2048, 4096):
    sc = synth_scalars_device(torch, B, n, 5, dev)
    out = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
    ctx.msm_batch_dev(B, n, sc.data_ptr(), 0, out.data_ptr(), s); torch.cuda.synchronize()
    ctx.set_profiling(True); ctx.last_timings()
    for _ in range(3): ctx.msm_batch_dev(B, n, sc.data_ptr(), 0, out.data_ptr(), s)
    t = [us for tag, us in ctx.last_timings() if tag == 1]
    ctx.set_profiling(False)
    print("B=%5d  %.2f ms   -> %.3e madds/s" % (B, min(t) / 1e3, B * n * 19.85 / (min(t) * 1e-6)))
