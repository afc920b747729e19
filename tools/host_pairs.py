#!/usr/bin/env python3
"""Two host threads proving batches of DIFFERENT sizes through the host-pointer API, the way the combiner's two batch threads do at
saturation (closed-loop pairs like 870 / 2202 or 1633 / 1439): python tools/host_pairs.py 870,2202 1022,2050 1093,1979 1536,1536
Prints proofs/s of each pair over --seconds."""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import dusk_blindbidproof_amd as bbp
from bench_workloads import synth_bids

ap = argparse.ArgumentParser()
ap.add_argument("pairs", nargs="*", default=["870,2202", "1022,2050", "1093,1979", "1536,1536"])
ap.add_argument("--seconds", type=float, default=4.0)
a = ap.parse_args()
ctx = bbp.Context(0)
N = 8
ins, ents, pubs, qz = synth_bids(ctx, 256, N, seed=3)
def blob(B):
    return b"".join(ins[i % 256] for i in range(B))
for pair in a.pairs:
    sizes = [int(x) for x in pair.split(",")]
    blobs = [blob(B) for B in sizes]
    for B, bl in zip(sizes, blobs):
        ctx.prove_batch(B, N, bl, None)
    done = [0] * len(sizes)
    stop = time.perf_counter() + a.seconds
    def work(i):
        while time.perf_counter() < stop:
            out, st = ctx.prove_batch(sizes[i], N, blobs[i], None)
            done[i] += sizes[i]
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(sizes))]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    print("pair %-10s %.0f proofs/s  (%s)" % (pair, sum(done) / dt, ", ".join("%d x %d" % (d // s, s) for d, s in zip(done, sizes))), flush=True)
