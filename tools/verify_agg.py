#!/usr/bin/env python3
"""Aggregated verification (bbp_verify_batch_aggregated_dev) against the per-proof path on one 8192-verification
call: honest batch and ~1 % corrupted, several group sizes, per-kernel device time of the aggregated pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import make_workload, _kernel_table
dev = torch.device("cuda", 0)
ctx = bbp.Context(0)
s = None  # the context's own stream (include/bbp.h BBP_STREAM_CONTEXT); callers synchronise the device
pw = make_workload("prove", ctx, bbp, torch, dev, 1024, 8, 1)
pw.step(s); torch.cuda.synchronize()
rec = pw.rec
recs = pw.out_dev.view(1024, rec)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
stride = rec + 96 + 32 * 8
tail = torch.frombuffer(bytearray(b"".join(pw.qz[i] + pw.pubs[i] for i in range(1024))), dtype=torch.uint8).to(dev).view(1024, stride - rec)
vin = torch.empty((B, stride), dtype=torch.uint8, device=dev)
for t in range(B // 1024):
    vin[t * 1024:(t + 1) * 1024, :rec] = recs
    vin[t * 1024:(t + 1) * 1024, rec:] = tail
ent = torch.frombuffer(bytearray(os.urandom(B * 32)), dtype=torch.uint8).to(dev)
st = torch.full((B,), -1, dtype=torch.int32, device=dev)

def run(label, fn, expect_bad, reps=3):
    best = 1e9
    for _ in range(reps):
        st.fill_(-1)
        torch.cuda.synchronize(); t = time.perf_counter()
        extra = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        got = st.cpu().tolist()
        assert [i for i, v in enumerate(got) if v != 0] == expect_bad, "flags differ (%s)" % label
    print("%-44s %7.1f ms  %8.0f verifications/s %s" % (label, best * 1e3, B / best, extra if extra is not None else ""), flush=True)

for corrupt in (0, B // 100):
    bad = sorted({(i * 101 + 7) % B for i in range(corrupt)})
    for i in bad:
        vin[i, 100 + (i % 900)] ^= 0x20
    print("--- %d proofs, %d corrupted" % (B, len(bad)))
    run("per proof", lambda: ctx.verify_batch_dev(B, 8, vin.data_ptr(), ent.data_ptr(), st.data_ptr(), s), bad)
    for G in (8, 16, 32, 64, 128, 256):
        run("aggregated, groups of %d" % G,
            lambda: "fallback %d" % ctx.verify_batch_aggregated_dev(B, 8, vin.data_ptr(), ent.data_ptr(), st.data_ptr(), G, s), bad)

# per-kernel time of one aggregated pass (honest batch would be nicer, but the corrupted one shows the fallback too)
ctx.set_profiling(True)
ctx.verify_batch_aggregated_dev(B, 8, vin.data_ptr(), ent.data_ptr(), st.data_ptr(), 64, s)
torch.cuda.synchronize()
for name, v in sorted(_kernel_table(ctx.last_timings()).items(), key=lambda kv: -kv[1]["total_us"]):
    print("  %-18s %3d launches %9.1f us" % (name, v["launches"], v["total_us"]))
