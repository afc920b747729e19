#!/usr/bin/env python3
"""BASELINE.json configs[4] shape on one GPU: streaming prove + verify at sustained ingest.

    python tools/stream_bench.py [--bids 65536] [--chunk 1024] [--items 8] [--depth 2]

Bids arrive in host memory in chunks; each chunk is copied H2D (pinned, async), proved, its records verified, and records +
flags copied back.  `depth` chunk slots (default 3): inputs are staged one chunk ahead, so the copies and the prover's opening stage overlap the
previous chunk's MSM stage.  Reports sustained proofs/s (= verifies/s) and chunk latency percentiles (host-visible: submit -> results on host).
The distinct inputs are a tile of 256 synthetic bids (building 1M witnesses in Python would dominate the run), with fresh
entropy per chunk so every proof is different.
"""
import argparse, hashlib, json, os, sys, time
# the engine itself keeps four streams busy (caller's + opening + two more slices) = HIP's default number of hardware queues;
# this tool adds a copy stream, and a fifth stream would share a queue with one of them and serialise behind it
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import synth_bids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bids", type=int, default=65536)
    ap.add_argument("--chunk", type=int, default=1024)
    ap.add_argument("--items", type=int, default=8)
    ap.add_argument("--depth", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = bbp.Context(0)
    N, C = a.items, a.chunk
    tile = 256
    ins, ents, pubs, qz = synth_bids(ctx, tile, N, seed=9)
    rec = bbp.record_size(N)
    in_stride, ent_stride, v_stride = len(ins[0]), len(ents[0]), rec + 96 + 32 * N
    host_in = torch.frombuffer(bytearray(b"".join(ins[i % tile] for i in range(C))), dtype=torch.uint8).pin_memory()
    vtail = b"".join(qz[i % tile] + pubs[i % tile] for i in range(C))
    n_chunks = a.bids // C
    eng = torch.cuda.Stream()      # ONE engine stream: calls on a context are ordered; overlap across chunks happens inside the engine
    cp = torch.cuda.Stream()       # input copies
    slots = []
    for s in range(a.depth):
        slots.append(dict(
            st=eng, h_ent=torch.empty(C * ent_stride, dtype=torch.uint8).pin_memory(),
            d_in=torch.empty(C * in_stride, dtype=torch.uint8, device=dev), d_ent=torch.empty(C * ent_stride, dtype=torch.uint8, device=dev),
            d_rec=torch.empty(C * rec, dtype=torch.uint8, device=dev), d_vin=torch.empty(C * v_stride, dtype=torch.uint8, device=dev),
            d_vtail=torch.frombuffer(bytearray(vtail), dtype=torch.uint8).to(dev), d_vent=torch.zeros(C * 32, dtype=torch.uint8, device=dev),
            d_st=torch.full((C,), -1, dtype=torch.int32, device=dev), h_rec=torch.empty(C * rec, dtype=torch.uint8).pin_memory(),
            h_st=torch.empty(C, dtype=torch.int32).pin_memory(), done=torch.cuda.Event(), ev_in=torch.cuda.Event(), t0=None, busy=False))
    lat, bad = [], 0
    ent_np = np.frombuffer(bytearray(b"".join(ents[i % tile] for i in range(C))), dtype=np.uint8).reshape(C, ent_stride).copy()

    def finish(sl):
        nonlocal bad
        sl["done"].synchronize()
        lat.append(time.perf_counter() - sl["t0"])
        bad += int((sl["h_st"] != 0).sum())
        sl["busy"] = False

    def stage(k, sl):
        # host side of the ingest, one chunk AHEAD of its prove call: fresh prover entropy per chunk (blindings stay those of the
        # tile, rng seed varies), then the async H2D copies.  With the GPU saturated a copy can wait tens of milliseconds for its
        # turn; issued a chunk early that wait is off the critical path (the engine needs complete inputs at call time).
        seeds = np.frombuffer(hashlib.shake_256(b"chunk%d" % k).digest(32 * C), dtype=np.uint8).reshape(C, 32)
        ent_np[:, ent_stride - 32:] = seeds
        sl["h_ent"].copy_(torch.from_numpy(ent_np.reshape(-1)))
        sl["t0"] = time.perf_counter()
        with torch.cuda.stream(cp):
            sl["d_in"].copy_(host_in, non_blocking=True)
            sl["d_ent"].copy_(sl["h_ent"], non_blocking=True)
            sl["ev_in"].record(cp)

    def submit(k, sl):
        sl["ev_in"].synchronize()
        with torch.cuda.stream(sl["st"]):
            s = sl["st"].cuda_stream
            ctx.prove_batch_dev(C, N, sl["d_in"].data_ptr(), sl["d_ent"].data_ptr(), sl["d_rec"].data_ptr(), s)
            vin = sl["d_vin"].view(C, v_stride)
            vin[:, :rec] = sl["d_rec"].view(C, rec)
            vin[:, rec:] = sl["d_vtail"].view(C, v_stride - rec)
            ctx.verify_batch_dev(C, N, sl["d_vin"].data_ptr(), sl["d_vent"].data_ptr(), sl["d_st"].data_ptr(), s)
            sl["h_rec"].copy_(sl["d_rec"], non_blocking=True)
            sl["h_st"].copy_(sl["d_st"], non_blocking=True)
            sl["done"].record(sl["st"])
        sl["busy"] = True

    # warm-up (allocations, circuit compile)
    stage(-1, slots[0]); submit(-1, slots[0]); finish(slots[0]); lat.clear()
    t_start = time.perf_counter()
    stage(0, slots[0])
    for k in range(n_chunks):
        if k + 1 < n_chunks:
            nxt = slots[(k + 1) % a.depth]
            if nxt["busy"]:
                finish(nxt)
            stage(k + 1, nxt)
        submit(k, slots[k % a.depth])
    for sl in slots:
        if sl["busy"]:
            finish(sl)
    wall = time.perf_counter() - t_start
    lat.sort()
    out = {"workload": "configs[4]-shaped streaming prove+verify, 1 GPU", "bids": n_chunks * C, "chunk": C, "depth": a.depth, "bid_list_len": N,
           "proofs_per_s": n_chunks * C / wall, "verifies_per_s": n_chunks * C / wall, "failed_verifications": bad,
           "chunk_latency_ms": {"p50": lat[len(lat) // 2] * 1e3, "p99": lat[min(len(lat) - 1, int(len(lat) * 0.99))] * 1e3, "max": lat[-1] * 1e3}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
