import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench import synth_scalars_device
dev = torch.device("cuda", 0)
ctx = bbp.Context(0)
s = None  # the context's own stream (include/bbp.h BBP_STREAM_CONTEXT); callers synchronise the device
B, n = 2048, 2049
out = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
def run(sc, label):
    ctx.msm_batch_dev(B, n, sc.data_ptr(), 0, out.data_ptr(), s); torch.cuda.synchronize()
    ctx.set_profiling(True); ctx.last_timings()
    for _ in range(3): ctx.msm_batch_dev(B, n, sc.data_ptr(), 0, out.data_ptr(), s)
    t = [us for tag, us in ctx.last_timings() if tag == 1]
    ctx.set_profiling(False)
    print(label, ["%.1f" % (x / 1e3) for x in t], "ms")
base = synth_scalars_device(torch, B, n, 5, dev)
run(base, "random            ")
for cnt in (64, 256, 582, 1024):
    sc = base.clone()
    sc[0::2, 1:1 + cnt, :] = sc[0::2, 1:2, :]      # cnt identical scalars in every other MSM
    run(sc, "identical x%-4d   " % cnt)
sc = base.clone(); sc[:, 1:583, :] = 0
run(sc, "582 zeros         ")
