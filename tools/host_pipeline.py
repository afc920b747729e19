#!/usr/bin/env python3
"""Host-pointer batch API from T threads at once: does the lock-only-while-enqueueing path pipeline consecutive calls?
    python tools/host_pipeline.py [--batch 1024] [--threads 1,2,3] [--iters 12]"""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import dusk_blindbidproof_amd as bbp
from bench_workloads import synth_bids

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--threads", default="1,2,3")
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--verify", action="store_true", help="each iteration proves then verifies its own batch")
a = ap.parse_args()
ctx = bbp.Context(0)
N, B = 8, a.batch
ins, ents, pubs, qz = synth_bids(ctx, B, N, seed=3)
blob_in, blob_ent = b"".join(ins), b"".join(ents)
rs = bbp.record_size(N)
ctx.prove_batch(B, N, blob_in, blob_ent)
for T in [int(x) for x in a.threads.split(",")]:
    def work():
        for _ in range(a.iters):
            out, st = ctx.prove_batch(B, N, blob_in, None)
            assert st == [0] * B
            if a.verify:
                vin = b"".join(out[i * rs:(i + 1) * rs] + qz[i] + pubs[i] for i in range(B))
                assert ctx.verify_batch(B, N, vin) == [0] * B
    th = [threading.Thread(target=work) for _ in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    print("threads %d: %.1f ms per batch, %.0f proofs/s%s" % (T, dt / (T * a.iters) * 1e3, T * a.iters * B / dt, " (+verify)" if a.verify else ""), flush=True)
