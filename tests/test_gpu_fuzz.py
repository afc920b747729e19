"""GPU tier: randomised differential runs against the C oracle -- MSM shapes / scalar patterns that stress the recoding, the split
into sub-MSMs and empty buckets, and prove / verify over random batch geometries and list lengths.  Seeds are fixed: a failure
reproduces."""
import hashlib
import random

import pytest

from oracle.ref_py import ristretto as rs
from tests import oracle_c
from tests.test_gpu_prove_verify import _synth_batch

pytestmark = pytest.mark.gpu
L = rs.L


@pytest.fixture(scope="module")
def oc(built):
    return oracle_c.load(built.build_oracle())


def _pattern_scalar(rnd):
    k = rnd.randrange(12)
    if k == 0:
        return 0
    if k == 1:
        return 1
    if k == 2:
        return L - 1 - rnd.randrange(3)
    if k == 3:
        return 1 << rnd.randrange(252)
    if k == 4:
        return ((1 << rnd.randrange(2, 252)) - 1) % L            # run of ones: one long carry chain in the NAF
    if k == 5:
        return int("10" * 126, 2) >> rnd.randrange(8)             # alternating bits
    if k == 6:
        return (0x7FF << rnd.randrange(0, 240)) % L               # a digit at the NAF magnitude limit
    if k == 7:
        return (0x801 << rnd.randrange(0, 240)) % L
    if k == 8:
        return rnd.getrandbits(rnd.randrange(1, 64))              # small values (witness-like)
    return rnd.randrange(L)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_msm_fuzz(ctx, bbp, oc, seed):
    rnd = random.Random(1000 + seed)
    for case in range(14):
        layout = rnd.choice([bbp.LAYOUT_BLIND_G_H, bbp.LAYOUT_BLIND_G])
        m = rnd.choice([1, 2, 3, 17, 63, 64, 65, 127, 128, 129, 300, 1023, 1466, 2048]) if case % 2 else rnd.randrange(1, 2049)
        n_terms = 1 + (2 * m if layout == bbp.LAYOUT_BLIND_G_H else m)
        B = rnd.choice([1, 2, 3, 5, 31, 64, 127, 128, 129, 200]) if n_terms < 700 else rnd.choice([1, 2, 3, 5, 9])
        repeated = _pattern_scalar(rnd)
        rows = []
        for b in range(B):
            style = rnd.randrange(4)
            if style == 0:
                row = [_pattern_scalar(rnd) for _ in range(n_terms)]
            elif style == 1:
                row = [repeated] * n_terms                           # every term in one bucket
            elif style == 2:
                row = [0] * n_terms                                  # empty MSM -> identity
                row[rnd.randrange(n_terms)] = _pattern_scalar(rnd)
            else:
                row = [rnd.randrange(L) for _ in range(n_terms)]
            rows.append(b"".join(rs.sc_bytes(v) for v in row))
        got = ctx.msm_batch(B, n_terms, b"".join(rows), layout)
        exp = oc.msm_layout_many(rows, [n_terms] * B, [layout] * B, threads=8)
        assert got == exp, (seed, case, layout, n_terms, B)


def test_msm_fuzz_half_wavefront_fold(bbp, oc):
    """k_msm_fold_half (two MSMs per wavefront, 32 lanes x 32 buckets) serves launches of 512 MSMs and more by default; with
    BBP_FOLD_HALF_FROM=1 every launch uses it: odd MSM counts (idle upper half), split MSMs, one-term and empty MSMs, every term in one
    bucket (runs of chunk-leading partial sums), and the full-width 4097-term shape -- all against the C oracle."""
    import os
    old = os.environ.get("BBP_FOLD_HALF_FROM")
    os.environ["BBP_FOLD_HALF_FROM"] = "1"
    try:
        c2 = bbp.Context(0)
    finally:
        if old is None:
            os.environ.pop("BBP_FOLD_HALF_FROM", None)
        else:
            os.environ["BBP_FOLD_HALF_FROM"] = old
    try:
        rnd = random.Random(77)
        shapes = [(bbp.LAYOUT_BLIND_G_H, 1, 1), (bbp.LAYOUT_BLIND_G_H, 3, 3), (bbp.LAYOUT_BLIND_G, 34, 5), (bbp.LAYOUT_BLIND_G_H, 129, 127),
                  (bbp.LAYOUT_BLIND_G_H, 257, 200), (bbp.LAYOUT_BLIND_G, 1467, 9), (bbp.LAYOUT_BLIND_G_H, 2933, 3), (bbp.LAYOUT_BLIND_G_H, 4097, 2),
                  (bbp.LAYOUT_BLIND_G_H, 2049, 65)]
        for layout, n_terms, B in shapes:
            repeated = _pattern_scalar(rnd)
            rows = []
            for b in range(B):
                style = (b + n_terms) % 4
                if style == 0:
                    row = [_pattern_scalar(rnd) for _ in range(n_terms)]
                elif style == 1:
                    row = [repeated] * n_terms
                elif style == 2:
                    row = [0] * n_terms
                    row[rnd.randrange(n_terms)] = _pattern_scalar(rnd)
                else:
                    row = [rnd.randrange(L) for _ in range(n_terms)]
                rows.append(b"".join(rs.sc_bytes(v) for v in row))
            got = c2.msm_batch(B, n_terms, b"".join(rows), layout)
            exp = oc.msm_layout_many(rows, [n_terms] * B, [layout] * B, threads=8)
            assert got == exp, (layout, n_terms, B)
        assert c2.health() == 0
    finally:
        c2.close()


@pytest.mark.parametrize("knobs", [{}, {"BBP_SORT_STAGED": "7"}, {"BBP_SORT_STAGED": "0"}])
def test_msm_sort_oversized_bucket(bbp, oc, knobs):
    """k_msm_sort_staged places the entries of a window of buckets in an LDS image of 16 384 entries (32 768 for MSMs wider than 3000
    terms with bit 2 of BBP_SORT_STAGED); a single bucket larger than the image takes a pass of its own with direct stores.  The scalar
    sum_j 2^(13 j) has 19 NAF digits of magnitude 1: with every term equal to it, bucket 1 holds 19 n entries (55 727 at n = 2933,
    77 843 at 4097) and every other bucket is empty; a second pattern puts two thirds of the terms there and spreads the rest."""
    import os
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        c2 = bbp.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        ones = sum(1 << (13 * j) for j in range(19))
        rnd = random.Random(5)
        for n_terms, B in ((2933, 3), (2049, 4), (4097, 2)):
            rows = []
            for b in range(B):
                if b % 2 == 0:
                    row = [ones] * n_terms
                else:
                    row = [ones if i % 3 else rnd.randrange(L) for i in range(n_terms)]
                rows.append(b"".join(rs.sc_bytes(v) for v in row))
            got = c2.msm_batch(B, n_terms, b"".join(rows), bbp.LAYOUT_BLIND_G_H)
            exp = oc.msm_layout_many(rows, [n_terms] * B, [bbp.LAYOUT_BLIND_G_H] * B, threads=8)
            assert got == exp, (knobs, n_terms)
        assert c2.health() == 0
    finally:
        c2.close()


@pytest.mark.parametrize("seed", [11, 12])
def test_prove_verify_fuzz(ctx, bbp, oc, seed):
    rnd = random.Random(seed)
    for case in range(5):
        N = rnd.choice([1, 2, 3, 5, 8, 13, 21, 40]) if case else rnd.choice([101, 202])
        B = rnd.choice([1, 2, 3, 7, 19]) if N < 100 else rnd.choice([1, 3])
        ins, ents, vins = _synth_batch(ctx, B, N, seed=seed * 100 + case)
        rs_ = bbp.record_size(N)
        out, st = ctx.prove_batch(B, N, b"".join(ins), b"".join(ents))
        assert st == [0] * B
        cout, cst = oc.prove_many(b"".join(ins), b"".join(ents), B, N, threads=8)
        assert cst == [0] * B and out == cout, (seed, case, N, B)
        vin = bytearray(b"".join(out[i * rs_:(i + 1) * rs_] + v[0] + v[1] + v[2] + v[3] for i, v in enumerate(vins)))
        bad = rnd.randrange(B)
        vin[bad * (rs_ + 96 + 32 * N) + 1 + rnd.randrange(1100)] ^= 1 << rnd.randrange(8)
        got = ctx.verify_batch(B, N, bytes(vin))
        exp = oc.verify_many(bytes(vin), B, N, threads=8)
        assert [g != 0 for g in got] == [e != 0 for e in exp] and got[bad] != 0, (seed, case, N, B, got, exp)
