#!/usr/bin/env python3
"""Randomised differential run over the paths round 3 added or rewired: python tools/fuzz_round3.py [iterations] [seed]
Every iteration draws a list length N, a batch size B (across the cooperative-rng / rotation / variable-base-kernel thresholds) and
a handle (context, two-member pool), proves through prove_batch OR the asynchronous calls OR concurrent single calls, compares every
record with the C oracle byte for byte, corrupts a few rows (proof bytes, public inputs, non-canonical scalars, undecodable points)
and compares the verdicts of verify_batch, verify_batch_aggregated and the asynchronous verify with the oracle's."""
import os, random, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import __graft_entry__ as ge
import dusk_blindbidproof_amd as bbp
from tests import oracle_c
from tests.test_gpu_prove_verify import _synth_batch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20260
oc = oracle_c.load(ge.build_oracle())
ctx = bbp.Context(0)
pool = bbp.Pool([0, 0])
t_start = time.time()
n_proofs = n_verdicts = 0
for it in range(iters):
    rnd = random.Random(seed0 + it)
    N = rnd.choice([1, 2, 3, 5, 8, 8, 8, 13, 30, 77, 202])
    B = rnd.choice([1, 2, 5, 17, 64, 65, 129, 257, 300]) if N <= 13 else rnd.choice([1, 3, 9, 33])
    h = pool if it % 3 == 2 else ctx
    ins, ents, vins = _synth_batch(ctx, B, N, seed=seed0 * 1000 + it)
    rs_ = bbp.record_size(N)
    exp, est = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=16)
    assert est == [0] * B
    mode = it % 4
    if mode == 0 or B > 64:
        out, st = h.prove_batch(B, N, b"".join(ins), b"".join(ents))
        assert st == [0] * B
    elif mode == 1:  # asynchronous calls, all queued from this thread
        got, keep, ev = {}, [], threading.Event()
        for i in range(B):
            keep.append(h.prove_async(ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"), ents[i],
                                      (lambda i: lambda s, r: (got.__setitem__(i, (s, r)), len(got) == B and ev.set()))(i)))
        assert ev.wait(300)
        assert all(got[i][0] == 0 for i in range(B))
        out = b"".join(got[i][1] for i in range(B))
    else:  # concurrent single calls from threads (the reference's calling pattern)
        res = [None] * B
        def work(lo, hi):
            for i in range(lo, hi):
                res[i] = h.prove(ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"), ents[i])
        T = min(8, B)
        th = [threading.Thread(target=work, args=(t * B // T, (t + 1) * B // T)) for t in range(T)]
        [x.start() for x in th]; [x.join() for x in th]
        out = b"".join(res)
    assert out == exp, (it, N, B, mode, "records differ from the oracle's")
    n_proofs += B
    rows = [bytearray(out[i * rs_:(i + 1) * rs_] + b"".join(vins[i])) for i in range(B)]
    bad = set(rnd.sample(range(B), min(B, rnd.choice([0, 1, 1, 2, 5]))))
    for i in bad:
        kind = rnd.randrange(4)
        if kind == 0:
            rows[i][1 + rnd.randrange(1120)] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            rows[i][rs_ + rnd.randrange(96 + 32 * N)] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            rows[i][1 + 32 * 9:1 + 32 * 10] = b"\xff" * 32
        else:
            j = rnd.randrange(4 + N)
            rows[i][rs_ - 32 * (4 + N) + 32 * j:rs_ - 32 * (4 + N) + 32 * (j + 1)] = b"\xff" * 32
    blob = b"".join(bytes(r) for r in rows)
    ost = oc.verify_many(blob, B, N, threads=16)
    plain = h.verify_batch(B, N, blob)
    assert [p != 0 for p in plain] == [o != 0 for o in ost], (it, N, B, plain, ost)
    assert all((p == 3) == (o == 3) for p, o in zip(plain, ost)), (it, "format classes differ")
    agg, _ = h.verify_batch_aggregated(B, N, blob, rnd.choice([0, 4, 16]))
    assert agg == plain, (it, N, B, "aggregated verdicts differ")
    if B <= 64:
        verd, ev = {}, threading.Event()
        stride = rs_ + 96 + 32 * N
        keep = [h.verify_async(blob[i * stride:i * stride + rs_], blob[i * stride + rs_:i * stride + rs_ + 32], blob[i * stride + rs_ + 32:i * stride + rs_ + 64],
                               blob[i * stride + rs_ + 64:i * stride + rs_ + 96], blob[i * stride + rs_ + 96:(i + 1) * stride],
                               (lambda i: lambda s: (verd.__setitem__(i, s), len(verd) == B and ev.set()))(i)) for i in range(B)]
        assert ev.wait(300)
        assert [verd[i] for i in range(B)] == plain, (it, N, B, "asynchronous verdicts differ")
    n_verdicts += 2 * B
    if it % 10 == 9:
        print("iteration %d: %d proofs byte-equal, %d verdicts equal so far (%.0f s)" % (it + 1, n_proofs, n_verdicts, time.time() - t_start), flush=True)
assert ctx.health() == 0 and pool.health() == 0
pool.close()
ctx.close()
print("fuzz_round3: %d iterations, %d proofs byte-equal to the C oracle, %d verdicts equal, health clean" % (iters, n_proofs, n_verdicts))
