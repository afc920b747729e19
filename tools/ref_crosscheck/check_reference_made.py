#!/usr/bin/env python3
"""Feeds reference-made records (bbp-ref-crosscheck make K N out.txt) to this repository's verifiers: the C oracle always, the
GPU engine when a device is present.  Every record must be accepted; with the witness columns present the prover's byte layout is
also compared field by field (lengths, version byte)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from tests import oracle_c

oc = oracle_c.load(ge.build_oracle())
ctx = None
try:
    import torch
    if torch.cuda.is_available():
        import dusk_blindbidproof_amd as bbp
        ctx = bbp.Context(0)
except Exception:
    pass
bad = 0
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) < 7:
        continue
    n = int(f[1])
    rec, score, z, seed = (bytes.fromhex(x) for x in f[3:7])
    pub = b"".join(bytes.fromhex(x) for x in f[7:7 + n])
    o = oc.verify(rec, score, z, seed, pub)
    g = ctx.verify(rec, score, z, seed, pub) if ctx else None
    print(f[0], "record %d bytes, version byte 0x%02x" % (len(rec), rec[0]), "oracle:", o, "engine:", g)
    bad += (o != 0) + (g not in (None, 0))
sys.exit(1 if bad else 0)
