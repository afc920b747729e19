#!/usr/bin/env python3
"""BASELINE configs[3] per-GPU shard: one 8192-verification call (1024 distinct proofs tiled 8x, ~1 % corrupted at known indices)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import make_workload, VerifyWorkload
dev = torch.device("cuda", 0)
ctx = bbp.Context(0)
s = None  # the context's own stream (include/bbp.h BBP_STREAM_CONTEXT); callers synchronise the device
pw = make_workload("prove", ctx, bbp, torch, dev, 1024, 8, 1)
pw.step(s); torch.cuda.synchronize()
rec = pw.rec
recs = pw.out_dev.view(1024, rec)
# tile 1024 distinct proofs 8x into one 8192-verification call, corrupt ~1 %
B = 8192
stride = rec + 96 + 32 * 8
tail = torch.frombuffer(bytearray(b"".join(pw.qz[i] + pw.pubs[i] for i in range(1024))), dtype=torch.uint8).to(dev).view(1024, stride - rec)
vin = torch.empty((B, stride), dtype=torch.uint8, device=dev)
for t in range(8):
    vin[t * 1024:(t + 1) * 1024, :rec] = recs
    vin[t * 1024:(t + 1) * 1024, rec:] = tail
bad = sorted({(i * 101 + 7) % B for i in range(82)})
for i in bad:
    vin[i, 100 + (i % 900)] ^= 0x20
ent = torch.zeros(B * 32, dtype=torch.uint8, device=dev)
st = torch.full((B,), -1, dtype=torch.int32, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.verify_batch_dev(B, 8, vin.data_ptr(), ent.data_ptr(), st.data_ptr(), s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    got = st.cpu().tolist()
    assert [i for i, v in enumerate(got) if v != 0] == bad, "flags differ"
    print("8192 verifications: %.1f ms -> %.0f /s, %d corrupted all flagged" % (dt * 1e3, B / dt, len(bad)), flush=True)
