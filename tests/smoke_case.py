"""One small invocation of the hot path on cuda:0, checked against the oracle (called by __graft_entry__.smoke())."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(bbp):
    import __graft_entry__ as ge
    from bench_workloads import synth_bids
    from tests import oracle_c
    oc = oracle_c.load(ge.build_oracle())
    ctx = bbp.Context(0)
    try:
        B, N = 2, 3
        ins, ents, pubs, qz = synth_bids(ctx, B, N, seed=42)
        out, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
        assert st == [0] * B, st
        rs = bbp.record_size(N)
        for i in range(B):
            rec = out[i * rs:(i + 1) * rs]
            rc, exp = oc.prove(ins[i][:224], ins[i][224:224 + 32 * N], int.from_bytes(ins[i][-8:], "little"), ents[i])
            assert rc == 0 and rec == exp, "device proof %d differs from the oracle's" % i
            assert oc.verify(rec, qz[i][:32], qz[i][32:64], qz[i][64:96], pubs[i]) == 0
            assert ctx.verify(rec, qz[i][:32], qz[i][32:64], qz[i][64:96], pubs[i]) == 0
            bad = bytearray(rec)
            bad[77] ^= 1
            assert ctx.verify(bytes(bad), qz[i][:32], qz[i][32:64], qz[i][64:96], pubs[i]) == 1
        print("smoke ok: %d blind-bid proofs (N=%d) proved on the GPU, byte-identical to the oracle, verified both ways" % (B, N))
    finally:
        ctx.close()


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    import dusk_blindbidproof_amd
    run(dusk_blindbidproof_amd)
