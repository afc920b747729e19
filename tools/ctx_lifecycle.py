#!/usr/bin/env python3
"""Create / use / destroy a context repeatedly: device memory in use must not grow from one iteration to the next."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dusk_blindbidproof_amd as bbp
from bench_workloads import synth_bids
free0, total = torch.cuda.mem_get_info()
for it in range(4):
    ctx = bbp.Context(0)
    ins, ents, pubs, qz = synth_bids(ctx, 96, 5, 3 + it)
    out, st = ctx.prove_batch(96, 5, b"".join(ins), b"".join(ents))
    assert st == [0] * 96
    rs = bbp.record_size(5)
    vin = b"".join(out[i * rs:(i + 1) * rs] + qz[i] + pubs[i] for i in range(96))
    assert ctx.verify_batch(96, 5, vin) == [0] * 96
    used = (total - torch.cuda.mem_get_info()[0]) / 2**20
    ctx.close()
    after = (total - torch.cuda.mem_get_info()[0]) / 2**20
    print("iteration %d: %.0f MiB in use with the context, %.0f MiB after close" % (it, used, after), flush=True)
