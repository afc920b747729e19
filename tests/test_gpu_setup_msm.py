"""GPU parity: resident generator/constant tables and the batched Pippenger MSM (K1) against the big-int oracle.
All calls go through the C-ABI (include/bbp.h)."""
import hashlib
import random

import pytest

from oracle.ref_py import blindbid as bb, ristretto as rs

pytestmark = pytest.mark.gpu
L = rs.L


def test_generators_match_golden(ctx, bbp, golden):
    kat = golden("setup_kat.json")
    assert ctx.generator(bbp.BASE_B).hex() == kat["B"]
    assert ctx.generator(bbp.BASE_BBLIND).hex() == kat["B_blinding"]
    for i in range(3):
        assert ctx.generator(bbp.BASE_G0 + i).hex() == kat["G"][i]
        assert ctx.generator(bbp.BASE_H0 + i).hex() == kat["H"][i]
    assert ctx.generator(bbp.BASE_G0 + 2047).hex() == kat["G_last"]
    assert ctx.generator(bbp.BASE_H0 + 2047).hex() == kat["H_last"]
    g = hashlib.sha256(b"".join(ctx.generator(bbp.BASE_G0 + i) for i in range(2048))).hexdigest()
    h = hashlib.sha256(b"".join(ctx.generator(bbp.BASE_H0 + i) for i in range(2048))).hexdigest()
    assert g == kat["G_sha256"] and h == kat["H_sha256"]


def test_mimc_constants_match_golden(ctx, golden):
    kat = golden("setup_kat.json")
    got = [ctx.mimc_constant(i).hex() for i in range(90)]
    assert got == kat["mimc_c"]
    # reference recipe, src/blindbid/mod.rs:7-24, recomputed with hashlib
    assert got == [rs.sc_bytes(c).hex() for c in bb.mimc_constants()]


def _bases(layout, n_terms, bbp):
    pc, bp = bb.gens(2048)
    if layout == bbp.LAYOUT_BLIND_G_H:
        m = (n_terms - 1) // 2
        return [pc.B_blinding] + bp.G[:m] + bp.H[:m]
    return [pc.B_blinding] + bp.G[:n_terms - 1]


def _edge_scalars():
    return [0, 1, 2, L - 1, L - 2, 2**252, 2**252 - 1, 1024, 1025, 2047, 2048, (1 << 11) - 1, 1 << 242, (1 << 253) % L,
            sum(1024 << (11 * j) for j in range(23)) % L, sum(1025 << (11 * j) for j in range(22)) % L]


@pytest.mark.parametrize("layout_name,n_terms,B", [("gh", 1 + 2 * 24, 5), ("g", 1 + 33, 4), ("gh", 3, 3), ("g", 1, 2)])
def test_msm_small_vs_oracle(ctx, bbp, layout_name, n_terms, B):
    layout = bbp.LAYOUT_BLIND_G_H if layout_name == "gh" else bbp.LAYOUT_BLIND_G
    rnd = random.Random(n_terms * 131 + B)
    bases = _bases(layout, n_terms, bbp)
    edge = _edge_scalars()
    sc = [[(edge[(i + 7 * b) % len(edge)] if (i + b) % 3 == 0 else rnd.randrange(L)) for i in range(n_terms)] for b in range(B)]
    sc[0] = [0] * n_terms  # all-zero MSM -> identity
    buf = b"".join(rs.sc_bytes(s) for row in sc for s in row)
    out = ctx.msm_batch(B, n_terms, buf, layout)
    for b in range(B):
        assert out[32 * b:32 * b + 32] == rs.encode(rs.msm(sc[b], bases)), b
    assert out[:32] == bytes(32)


def test_msm_rejects_noncanonical(ctx, bbp):
    buf = (L).to_bytes(32, "little") * 3
    with pytest.raises(bbp.BbpError) as e:
        ctx.msm_batch(1, 3, buf, bbp.LAYOUT_BLIND_G_H)
    assert e.value.status == 3


@pytest.mark.parametrize("layout_name,n_terms", [("gh", 2933), ("g", 1467), ("gh", 4097)])
def test_msm_full_size_vs_oracle(ctx, bbp, layout_name, n_terms):
    """BASELINE.json configs[1] shapes (N = 8: 2933 / 1467 terms) plus the table's maximum width."""
    layout = bbp.LAYOUT_BLIND_G_H if layout_name == "gh" else bbp.LAYOUT_BLIND_G
    B = 3
    bases = _bases(layout, n_terms, bbp)
    sc = [[rs.sc_wide(hashlib.sha512(b"msm%d/%d/%d" % (n_terms, b, i)).digest()) for i in range(n_terms)] for b in range(B)]
    # witness-like row: mostly small / boolean scalars
    sc[1] = [(i % 2) if i % 5 else sc[1][i] for i in range(n_terms)]
    buf = b"".join(rs.sc_bytes(s) for row in sc for s in row)
    out = ctx.msm_batch(B, n_terms, buf, layout)
    for b in range(B):
        assert out[32 * b:32 * b + 32] == rs.encode(rs.msm(sc[b], bases)), b


def test_msm_linearity_large_batch(ctx, bbp):
    """Size-independent property at batch scale: MSM(s + t) == MSM(s) + MSM(t) for 256 MSMs of 2933 terms."""
    n_terms, B = 2933, 256
    rnd = random.Random(99)
    s = [rnd.randrange(L) for _ in range(n_terms)]
    rows, exp_rows = [], []
    for b in range(B):
        t = [(x * (b + 1) + b) % L for x in s]
        rows.append(t)
    buf = b"".join(rs.sc_bytes(v) for row in rows for v in row)
    out = ctx.msm_batch(B, n_terms, buf, bbp.LAYOUT_BLIND_G_H)
    pts = [rs.decode(out[32 * b:32 * b + 32]) for b in range(B)]
    assert all(p is not None for p in pts)
    # rows[b] = (b+1)*s + b*1  =>  P_b = (b+1)*P_s + b*P_1 ;  check P_b - P_{b-1} is constant (= P_s + P_1)
    d0 = rs.pt_add(pts[1], rs.pt_neg(pts[0]))
    for b in range(2, B):
        assert rs.pt_eq(rs.pt_add(pts[b], rs.pt_neg(pts[b - 1])), d0), b


def test_msm_dev_noncanonical_scalars_do_not_fault(ctx, bbp):
    """include/bbp.h: the device-pointer variant cannot screen scalars; non-canonical ones give an unspecified point but must
    never fault (row indices stay inside the table whatever the bits are).  All-ones words are the worst case for the recoding
    (every window carries); canonical rows in the same call must still be exact."""
    import torch
    dev = torch.device("cuda", 0)
    n_terms, B = 301, 4  # layout 0: B~ + m G's + m H's
    rnd = random.Random(5)
    good = [rnd.randrange(L) for _ in range(n_terms)]
    rows = [good, [2**256 - 1] * n_terms, [L + i for i in range(n_terms)], [rnd.getrandbits(256) for _ in range(n_terms)]]
    buf = b"".join(v.to_bytes(32, "little") for row in rows for v in row)
    d_in = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev)
    d_out = torch.zeros(B * 32, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.msm_batch_dev(B, n_terms, d_in.data_ptr(), bbp.LAYOUT_BLIND_G_H, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = bytes(d_out.cpu().numpy().tobytes())
    assert out[:32] == rs.encode(rs.msm(good, _bases(bbp.LAYOUT_BLIND_G_H, n_terms, bbp)))
    for b in range(1, B):
        assert rs.decode(out[32 * b:32 * b + 32]) is not None  # still a valid group element


def test_config2_msm_only_batch_1024(ctx, bbp, built):
    """BASELINE.json configs[1] AT ITS STATED SIZE: a batch of 1024 blind-bid proofs, Pippenger MSM kernel only -- per proof the
    three commitment MSMs A_I1 / A_O1 / S1 (2933 + 1467 + 2933 terms at N = 8, SURVEY.md 8d "Config 2"), through
    bbp_msm_batch_dev with the scalars resident in HBM, "bit-exact check vs CPU":
      * 16 rows of every launch (first, last, 14 spread) byte-equal to the C oracle's msm_layout (vartime Pippenger / Straus, own
        field arithmetic), and
      * a size-independent property over ALL 1024 rows of every launch: rows b and 512 + b differ by one fixed scalar vector t,
        so P[512 + b] - P[b] = MSM(t) for every b (linearity), with MSM(t) itself taken from the big-int oracle."""
    import torch
    from tests import oracle_c
    oc = oracle_c.load(built.build_oracle())
    dev = torch.device("cuda", 0)
    B, half = 1024, 512
    shapes = [(2933, bbp.LAYOUT_BLIND_G_H), (1467, bbp.LAYOUT_BLIND_G), (2933, bbp.LAYOUT_BLIND_G_H)]
    for which, (n_terms, layout) in enumerate(shapes):
        rnd = random.Random(1000 + which)
        t = [rnd.randrange(L) for _ in range(n_terms)]
        base = [[rnd.getrandbits(252) for _ in range(n_terms)] for _ in range(half)]  # < 2^252 < l: canonical
        if which == 0:  # witness-like rows as the prover's a_L / a_R are: booleans, zeros, small values among full-size scalars
            base[3] = [(i % 2) if i % 5 else base[3][i] for i in range(n_terms)]
            base[4] = [0] * n_terms
        rows = base + [[(a + b) % L for a, b in zip(row, t)] for row in base]
        buf = b"".join(v.to_bytes(32, "little") for row in rows for v in row)
        d_in = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev)
        d_out = torch.zeros(B * 32, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.msm_batch_dev(B, n_terms, d_in.data_ptr(), layout, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out = bytes(d_out.cpu().numpy().tobytes())
        # (a) sampled rows against the C oracle
        sample = sorted({0, 3, 4, B - 1, half - 1, half, half + 3, half + 4} | {(73 * j + 11) % B for j in range(10)})
        assert len(sample) >= 16
        stride = 32 * n_terms
        exp = oc.msm_layout_many([buf[r * stride:(r + 1) * stride] for r in sample], [n_terms] * len(sample), [layout] * len(sample), 8)
        for j, r in enumerate(sample):
            assert out[32 * r:32 * r + 32] == exp[32 * j:32 * j + 32], (which, r)
        assert out[32 * 4:32 * 5] == bytes(32) if which == 0 else True  # the all-zero row is the identity
        # (b) linearity over every row of the launch
        pt_t = rs.msm(t, _bases(layout, n_terms, bbp))
        pts = [rs.decode(out[32 * b:32 * b + 32]) for b in range(B)]
        assert all(p is not None for p in pts), which
        for b in range(half):
            assert rs.pt_eq(rs.pt_add(pts[half + b], rs.pt_neg(pts[b])), pt_t), (which, b)
