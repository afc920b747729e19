#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the big-int oracle (oracle/ref_py) under FIXED entropy.

The reference holds no fixtures of its own and cannot be built here (no Rust toolchain,
SURVEY.md F4), so these vectors are produced by the build's own restatement; the public KATs that pin the
restatement's primitives live in tests/test_oracle_kat.py.  Usage:  python tests/golden/make_golden.py [--full]
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_py import blindbid as bb, ristretto as rs  # noqa: E402

L = rs.L
HERE = os.path.dirname(os.path.abspath(__file__))


def stream(tag, n):
    return hashlib.shake_256(b"bbp-golden-v1/" + tag).digest(n)


def case(name, rounds, cap, N, toggle):
    cons = bb.mimc_constants(rounds)
    d = int.from_bytes(stream(name.encode() + b"/d", 8), "little")
    k = rs.sc_wide(stream(name.encode() + b"/k", 64))
    seed = rs.sc_wide(stream(name.encode() + b"/seed", 64))
    w = bb.witness(d, k, seed, cons)
    pub = [rs.sc_wide(stream(name.encode() + b"/pub%d" % i, 64)) for i in range(N)]
    pub[toggle] = w["x"]
    ent = stream(name.encode() + b"/ent", 32 * (4 + N) + 32)
    # blinding scalars must be canonical for the C-ABI: reduce them here and re-serialise
    ent = b"".join(rs.sc_bytes(rs.sc_wide(ent[32 * i:32 * i + 32] + bytes(32))) for i in range(4 + N)) + ent[32 * (4 + N):]
    trace = {}
    t0 = time.time()
    pr = bb.prove(d, k, w["y"], w["y_inv"], w["q"], w["z_img"], seed, pub, toggle, ent, rounds, cap, trace)
    t1 = time.time()
    rec = pr.to_record()
    assert bb.verify(bb.Proof.from_record(rec, N), w["q"], w["z_img"], seed, pub, bytes(32), rounds, cap)
    print(name, "prove %.1fs verify %.1fs" % (t1 - t0, time.time() - t1), "proof bytes", len(pr.proof.to_bytes()))
    sc = lambda v: rs.sc_bytes(v).hex()
    return dict(name=name, rounds=rounds, cap=cap, N=N, toggle=toggle,
                d=sc(d), k=sc(k), seed=sc(seed), m=sc(w["m"]), x=sc(w["x"]), y=sc(w["y"]), y_inv=sc(w["y_inv"]),
                q=sc(w["q"]), z_img=sc(w["z_img"]), pub_list=[sc(p) for p in pub], entropy=ent.hex(),
                proof_len=len(pr.proof.to_bytes()), record=rec.hex(), trace=trace)


def case_noncanonical(name, N, toggle):
    """bid-list entries are 32 raw bytes taken through Scalar::from_bits (src/blindbid/bid.rs:27,
    verify.rs:115) -- bit 255 cleared, NOT reduced, used mod l downstream.  The list holds l, l + 1, 2^255 - 1, a value with bit
    255 set and a value >= 2^255 + l beside the bid's own x; `pub_list` in the fixture is the RAW bytes the caller hands over."""
    cons = bb.mimc_constants()
    d = int.from_bytes(stream(name.encode() + b"/d", 8), "little")
    k = rs.sc_wide(stream(name.encode() + b"/k", 64))
    seed = rs.sc_wide(stream(name.encode() + b"/seed", 64))
    w = bb.witness(d, k, seed, cons)
    raw = [L, L + 1, 2**255 - 1, 2**255 + 12345, 2**256 - 1, 2**255 + L + 7][:N]
    raw = [v.to_bytes(32, "little") for v in raw]
    while len(raw) < N:
        raw.append(stream(name.encode() + b"/pub%d" % len(raw), 32))
    raw[toggle] = rs.sc_bytes(w["x"])
    pub = [rs.sc_from_bits(b) for b in raw]
    ent = stream(name.encode() + b"/ent", 32 * (4 + N) + 32)
    ent = b"".join(rs.sc_bytes(rs.sc_wide(ent[32 * i:32 * i + 32] + bytes(32))) for i in range(4 + N)) + ent[32 * (4 + N):]
    trace = {}
    pr = bb.prove(d, k, w["y"], w["y_inv"], w["q"], w["z_img"], seed, pub, toggle, ent, trace=trace)
    rec = pr.to_record()
    assert bb.verify(bb.Proof.from_record(rec, N), w["q"], w["z_img"], seed, pub, bytes(32))
    sc = lambda v: rs.sc_bytes(v).hex()
    return dict(name=name, rounds=90, cap=2048, N=N, toggle=toggle, d=sc(d), k=sc(k), seed=sc(seed), y=sc(w["y"]), y_inv=sc(w["y_inv"]),
                q=sc(w["q"]), z_img=sc(w["z_img"]), pub_list=[b.hex() for b in raw], entropy=ent.hex(),
                proof_len=len(pr.proof.to_bytes()), record=rec.hex(), trace=trace)


def main():
    if "--noncanonical" in sys.argv:
        out = {"noncanonical": [case_noncanonical("noncanon_n7", 7, 4)]}
        json.dump(out, open(os.path.join(HERE, "proofs_noncanonical.json"), "w"), indent=1)
        return
    full = "--full" in sys.argv
    out = {"small": [case("r2n3", 2, 64, 3, 1), case("r3n1", 3, 64, 1, 0), case("r1n5", 1, 64, 5, 4)]}
    json.dump(out, open(os.path.join(HERE, "proofs_small.json"), "w"), indent=1)
    if full:
        outf = {"full": [case("full_n1", 90, 2048, 1, 0), case("full_n8", 90, 2048, 8, 3)]}
        json.dump(outf, open(os.path.join(HERE, "proofs_full.json"), "w"), indent=1)
    # generator / constant digests
    pc, bp = bb.gens(2048)
    gd = hashlib.sha256(b"".join(rs.encode(p) for p in bp.G)).hexdigest()
    hd = hashlib.sha256(b"".join(rs.encode(p) for p in bp.H)).hexdigest()
    cons = bb.mimc_constants()
    kat = dict(B=rs.encode(pc.B).hex(), B_blinding=rs.encode(pc.B_blinding).hex(),
               G=[rs.encode(p).hex() for p in bp.G[:3]], H=[rs.encode(p).hex() for p in bp.H[:3]],
               G_last=rs.encode(bp.G[2047]).hex(), H_last=rs.encode(bp.H[2047]).hex(),
               G_sha256=gd, H_sha256=hd,
               mimc_c=[rs.sc_bytes(c).hex() for c in cons],
               mimc_sha256=hashlib.sha256(b"".join(rs.sc_bytes(c) for c in cons)).hexdigest())
    json.dump(kat, open(os.path.join(HERE, "setup_kat.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
