import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
import dusk_blindbidproof_amd as bbp
from tests import oracle_c
import tests.test_gpu_prove_verify as pv
oc = oracle_c.load(ge.build_oracle())
ctx = bbp.Context(0)
for N in (4, 6, 7, 9, 10, 11, 16, 31, 50, 64, 100, 150, 199, 200, 201):
    B = 3
    ins, ents, vins = pv._synth_batch(ctx, B, N, seed=1000 + N)
    out, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
    cout, cst = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=8)
    assert st == [0] * B and cst == [0] * B and out == cout, N
    rs_ = bbp.record_size(N)
    vin = b"".join(out[i * rs_:(i + 1) * rs_] + v[0] + v[1] + v[2] + v[3] for i, v in enumerate(vins))
    assert ctx.verify_batch(B, N, vin) == [0] * B
    got, nfb = ctx.verify_batch_aggregated(B, N, vin, 2)
    assert got == [0] * B and nfb == 0
    print("N=%d ok" % N, flush=True)
