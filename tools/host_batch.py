#!/usr/bin/env python3
"""Host-pointer batch API on a batch larger than one engine call: bbp_prove_batch / bbp_verify_batch cut it into equal chunks
(BBP_HOST_CHUNK_PROVE / _VERIFY) whose calls pipeline.  Prints proofs/s and verifications/s end to end (host copies included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import synth_bids
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = 8
ctx = bbp.Context(0)
ins, ents, pubs, qz = synth_bids(ctx, 1024, N, 1)
rep = B // 1024
blob_in, blob_ent = b"".join(ins) * rep, b"".join(ents) * rep
rs_ = bbp.record_size(N)
for chunk in (os.environ.get("BBP_HOST_CHUNK_PROVE", "16384"),):
    ctx.prove_batch(1024, N, b"".join(ins), b"".join(ents))
    t = time.perf_counter()
    out, st = ctx.prove_batch(B, N, blob_in, blob_ent)
    dt = time.perf_counter() - t
    assert st == [0] * B and out[:1024 * rs_] == out[-1024 * rs_:]
    print("prove_batch  B=%d chunk=%s: %.1f ms  %.0f proofs/s" % (B, chunk, dt * 1e3, B / dt), flush=True)
vin = b"".join(out[i * rs_:(i + 1) * rs_] + qz[i % 1024] + pubs[i % 1024] for i in range(B))
for name, fn in (("verify_batch", lambda: ctx.verify_batch(B, N, vin)), ("verify_batch_aggregated", lambda: ctx.verify_batch_aggregated(B, N, vin)[0])):
    fn()
    t = time.perf_counter()
    got = fn()
    dt = time.perf_counter() - t
    assert got == [0] * B
    print("%s B=%d: %.1f ms  %.0f verifications/s" % (name, B, dt * 1e3, B / dt), flush=True)
