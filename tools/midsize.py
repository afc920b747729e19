"""Mid-size batches (B < 1024): time per call and proofs/s of back-to-back bbp_prove_batch_dev calls, plus a pipeline
check -- three different input sets are proven round-robin without host synchronisation and every output must equal the
output of the same input set proven alone (proofs are deterministic in their inputs + entropy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import make_workload

dev = torch.device("cuda", 0)
ctx = bbp.Context(0)
s = None  # the context's own stream (include/bbp.h BBP_STREAM_CONTEXT); callers synchronise the device
sizes = [int(x) for x in sys.argv[1:]] or [64, 128, 256, 384, 512, 768, 1024]
for B in sizes:
    wls = [make_workload("prove", ctx, bbp, torch, dev, B, 8, seed) for seed in (1, 2, 3)]
    ref = []
    for w in wls:
        w.step(s)
        torch.cuda.synchronize()
        ref.append(w.out_dev.clone())
        w.out_dev.zero_()
    torch.cuda.synchronize()
    for i in range(6):
        wls[i % 3].step(s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 12
    for i in range(reps):
        wls[i % 3].step(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    ok = all(torch.equal(w.out_dev, r) for w, r in zip(wls, ref))
    wls[0].check()
    print("B=%4d  %.1f ms/call  %.0f proofs/s  pipeline outputs %s" % (B, dt * 1e3, B / dt, "identical" if ok else "DIFFER"), flush=True)
    if not ok:
        sys.exit(1)
    del wls, ref
