"""not-gpu tier: the PRODUCT's limb arithmetic headers (dusk_blindbidproof_amd/csrc/{field,scalar,point,keccak}.h),
compiled for the host by tests/host_check.cpp, against the big-int oracle.  Catches arithmetic bugs without a GPU;
the shipped library never runs this code on the CPU."""
import ctypes
import hashlib
import random

import pytest

from oracle.ref_py import merlin, ristretto as rs

P, L = rs.P, rs.L


def b32(x):
    return x.to_bytes(32, "little")


@pytest.fixture(scope="module")
def lib(built):
    return ctypes.CDLL(built.build_hostcheck())


EDGE = [0, 1, 2, 19, 38, P - 1, P, P + 1, 2 * P - 1, 2 * P, 2 * P + 1, 2**255 - 1, 2**255, 2**256 - 1, 2**256 - 38, 2**256 - 39,
        2**256 - 37, 2**32 - 1, 2**224, (2**256 - 1) ^ (2**128)]


def test_field_ops(lib):
    rnd = random.Random(1)

    def fe_op(op, a, b=0):
        out = ctypes.create_string_buffer(32)
        lib.hc_fe_op(op, b32(a), b32(b), out)
        return int.from_bytes(out.raw, "little")
    M = (1 << 255) - 1  # the loader ignores bit 255, like dalek's FieldElement::from_bytes
    limb_edges = [(1 << 26) - 1, ((1 << 25) - 1) << 26, M, M - 18, M - 19, M - 20, (1 << 255) - (1 << 230), sum(1 << o for o in (25, 50, 76, 101, 127, 152, 178, 203, 229, 254))]
    vals = EDGE + limb_edges + [rnd.getrandbits(256) for _ in range(200)]
    for a0 in vals:
        a = a0 & M
        for b0 in rnd.sample(vals, 5) + EDGE[:4] + [2**256 - 1, 2**256 - 38] + limb_edges[:3]:
            b = b0 & M
            assert fe_op(0, a0, b0) == (a + b) % P
            assert fe_op(1, a0, b0) == (a - b) % P
            assert fe_op(2, a0, b0) == (a * b) % P
            assert fe_op(8, a0, b0) == (a * (b0 & 0x3ffffff)) % P
            m1, m2, m3 = a * b % P, b * b % P, b * a * a % P
            r = (2 * m1 + m2) * (m3 - m2 - m1) % P
            assert fe_op(10, a0, b0) == pow(r + m1 - m3, 2, P)
            # round 4: the device path threads the carries through the column sums (field.h fe_chain_step / fe_chain_wrap, seed 2^25 +
            # 2^50 on even columns); the same functions over plain-C column sums must give the same field element, limbs carried
            assert fe_op(11, a0, b0) == (a * b) % P
            assert fe_op(12, a0, b0) == pow(r + m1 - m3, 2, P)
        assert fe_op(3, a0) == a * a % P
        assert fe_op(9, a0) == 2 * a * a % P
        assert fe_op(5, a0) == a % P
        assert fe_op(6, a0) == (-a) % P
    for a0 in vals[:40]:
        a = a0 & M
        assert fe_op(4, a0) == pow(a % P, P - 2, P)
        assert fe_op(7, a0) == pow(a % P, (P - 5) // 8, P)


def test_scalar_ops(lib):
    rnd = random.Random(2)

    def sc_op(op, a, b=b32(0)):
        out = ctypes.create_string_buffer(32)
        lib.hc_sc_op(op, a, b, out)
        return int.from_bytes(out.raw, "little")
    svals = [0, 1, 2, L - 1, L - 2, L // 2, 2**252, 2**252 - 1] + [rnd.randrange(L) for _ in range(200)]
    for a in svals:
        for b in rnd.sample(svals, 5) + [0, 1, L - 1]:
            assert sc_op(0, b32(a), b32(b)) == (a + b) % L
            assert sc_op(1, b32(a), b32(b)) == (a - b) % L
            assert sc_op(2, b32(a), b32(b)) == (a * b) % L
        assert sc_op(6, b32(a)) == (-a) % L
    for a in svals[:30]:
        if a:
            assert sc_op(7, b32(a)) == pow(a, L - 2, L)   # Fermat ladder (kept as a cross-check)
    # safegcd inversion: edge values, powers of two and their neighbours, small values, dense random sample; 0 -> 0
    inv_vals = svals + [3, 4, 5, 2**30 - 1, 2**30, 2**30 + 1, 2**60, 2**90 - 1, L - 3, (L + 1) // 2, (L - 1) // 2, 2**251, 2**252 + 1]
    inv_vals += [2**k for k in range(0, 252, 7)] + [L - 2**k for k in range(1, 252, 11)] + [rnd.randrange(L) for _ in range(3000)]
    inv_vals += [rnd.getrandbits(rnd.randrange(1, 252)) for _ in range(500)]
    for a in inv_vals:
        assert sc_op(3, b32(a)) == (pow(a, L - 2, L) if a else 0), hex(a)
    for w in [0, 2**512 - 1, 2**256 - 1, 2**256, L, L << 256] + [rnd.getrandbits(512) for _ in range(200)]:
        assert sc_op(4, w.to_bytes(64, "little")) == w % L
    for w in [2**256 - 1, 2**255, 2**255 - 1, L, L + 1, 15 * L + 7] + [rnd.getrandbits(256) for _ in range(100)]:
        assert sc_op(5, b32(w)) == (w & (2**255 - 1)) % L
        assert lib.hc_sc_is_canonical(b32(w)) == (1 if w < L else 0)
    assert lib.hc_sc_is_canonical(b32(L - 1)) == 1


# RFC 9496 appendix A.3: encodings that must be rejected
BAD_ENCODINGS = [
    "00ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff", "f3ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
    "edffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f", "0100000000000000000000000000000000000000000000000000000000000000",
    "01ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f", "ed57ffd8c914fb201471d1c3d245ce3c746fcbe63a3679d51b6a516ebebe0e20",
    "c34c4e1826e5d403b78e246e88aa051c36ccf0aafebffe137d148a2bf9104562", "c940e5a4404157cfb1628b108db051a8d439e1a421394ec4ebccb9ec92a8ac78",
    "47cfc5497c53dc8e61c91d17fd626ffb1c49e2bca94eed052281b510b1117a24", "f1c6165d33367351b0da8f6e4511010c68174a03b6581212c71c0e1d026c3c72",
    "87260f7a2f12495118360f02c26a470f450dadf34a413d21042b43b9d93e1309", "26948d35ca62e643e26a83177332e6b6afeb9d08e4268b650f1f5bbd8d81d371",
    "4eac077a713c57b4f4397629a4145982c661f48044dd3f96427d40b147d9742f", "de6a7b00deadc788eb6b6c8d20c0ae96c2f2019078fa604fee5b87d6e989ad7b",
    "bcab477be20861e01e4a0e295284146a510150d9817763caf1a6f4b422d67042", "2a292df7e32cababbd9de088d1d1abec9fc0440f637ed2fba145094dc14bea08",
    "f4a9e534fc0d216c44b218fa0c42d99635a0127ee2e53c712f70609649fdff22", "8268436f8c4126196cf64b3c7ddbda90746a378625f9813dd9b8457077256731",
    "2810e5cbc2cc4d4eece54f61c6f69758e289aa7ab440b3cbeaa21995c2f4232b", "3eb858e78f5a7254d8c9731174a94f76755fd3941c0ac93735c07ba14579630e",
    "a45fdc55c76448c049a1ab33f17023edfb2be3581e9c7aade8a6125215e04220", "d483fe813c6ba647ebbfd3ec41adca1c6130c2beeee9d9bf065c8d151c5f396e",
    "8a2e1d30050198c65a54483123960ccc38aef6848e1ec8f5f780e8523769ba32", "32888462f8b486c68ad7dd9610be5192bbeaf3b443951ac1a8118419d9fa097b",
    "227142501b9d4355ccba290404bde41575b037693cef1f438c47f8fbf35d1165", "5c37cc491da847cfeb9281d407efc41e15144c876e0170b499a96a22ed31e01e",
    "445425117cb8c90edcbc7c1cc0e74f747f2c1efa5630a967c64f287792a48a4b", "ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f"]


def test_naf_recoding(lib):
    """sc_for_each_naf_digit (the MSM kernels' scalar recoding): value, oddness, magnitude, spacing, position and count bounds"""
    rnd = random.Random(7)
    vals = [0, 1, 2, 3, L - 1, L - 2, 2**252, 2**253 - 1, 2**252 - 1, (2**253 - 1) // 3, 0xfff, 0x800, 0x7ff, 2**200 - 1,
            int("10" * 126, 2), int("01" * 126, 2), (1 << 253) - (1 << 241), sum(1 << (13 * i) for i in range(19))]
    vals += [rnd.getrandbits(253) for _ in range(3000)] + [rnd.getrandbits(rnd.randrange(1, 253)) for _ in range(500)]
    total = {12: 0, 9: 0}
    for width, max_digits in ((12, 22), (9, 29)):
        pos = (ctypes.c_int32 * 64)()
        dig = (ctypes.c_int32 * 64)()
        for v in vals:
            n = lib.hc_sc_naf(width, b32(v), pos, dig)
            assert n <= max_digits
            assert sum(dig[i] << pos[i] for i in range(n)) == v
            for i in range(n):
                assert dig[i] & 1 and abs(dig[i]) < (1 << (width - 1)) and 0 <= pos[i] <= 253
                assert i == 0 or pos[i] >= pos[i - 1] + width
            total[width] += n
    assert total[12] / len(vals) < 20.5 and total[9] / len(vals) < 26.5  # ~253/13 and ~253/10 digits on average


def test_point_ops_and_codec(lib):
    rnd = random.Random(3)
    out = ctypes.create_string_buffer(32)
    lib.hc_basepoint(out)
    assert out.raw == rs.encode(rs.BASEPOINT)
    pts = [rs.IDENT, rs.BASEPOINT] + [rs.pt_mul(rnd.randrange(L), rs.BASEPOINT) for _ in range(16)]
    enc = [rs.encode(p) for p in pts]

    def ge_op(op, a, b=bytes(32)):
        o = ctypes.create_string_buffer(32)
        return o.raw if lib.hc_ge_op(op, a, b, o) else None
    for i, (p, e) in enumerate(zip(pts, enc)):
        assert ge_op(0, e) == e
        assert ge_op(1, e) == rs.encode(rs.pt_dbl(p))
        for j in [0, 1, i, (i * 7 + 3) % len(pts)]:
            q, f = pts[j], enc[j]
            assert ge_op(2, e, f) == rs.encode(rs.pt_add(p, q))
            assert ge_op(3, e, f) == rs.encode(rs.pt_add(p, rs.pt_neg(q)))
            assert ge_op(4, e, f) == rs.encode(rs.pt_add(p, q))
            assert ge_op(5, e, f) == rs.encode(rs.pt_add(p, rs.pt_neg(q)))
    for h in BAD_ENCODINGS:
        b = bytes.fromhex(h)
        assert rs.decode(b) is None, h
        assert ge_op(0, b) is None, h
    for i in range(40):
        u = hashlib.sha512(b"u%d" % i).digest()
        lib.hc_from_uniform(u, out)
        assert out.raw == rs.encode(rs.from_uniform_bytes(u))
    for i in range(4):
        s = rnd.randrange(L)
        lib.hc_scalarmult(b32(s), enc[3 + i], out)
        assert out.raw == rs.encode(rs.pt_mul(s, pts[3 + i]))


def test_rng_bulk_draws_equal_generic_path(lib):
    """The register-resident steady-state TranscriptRng draw (keccak.h merlin_rng_fill64_bulk) == byte-wise STROBE."""
    for count, wlen in [(1, 32), (7, 32), (50, 0), (300, 100)]:
        n = 64 * (count + 2)
        a, b = ctypes.create_string_buffer(n), ctypes.create_string_buffer(n)
        w = hashlib.shake_256(b"w%d" % count).digest(wlen) if wlen else b""
        assert lib.hc_merlin_rng_bulk(w, wlen, b"\x21" * 32, count, a, b) == 1
        assert a.raw == b.raw
        t = merlin.Transcript(b"BlindBidProofGadget")
        r = t.build_rng([(b"v_blinding", w)], b"\x21" * 32)
        assert b"".join(r.fill_bytes(64) for _ in range(count + 2)) == a.raw


def test_gate_assignments_interpreted_native_and_oracle_agree(lib):
    """csrc/witness.h on the host: a_L, a_R, a_O of one proof from (a) the compiled gadget program interpreted, (b) the gadget wiring
    written out (what the cooperative opening launches run), (c) the big-int oracle's Prover driven through the same gadgets
    (oracle/ref_py, src/gadgets.rs in call order).  All three must be the same canonical scalars, for several list lengths, toggle
    positions (including one outside the list: no bit set) and an input above the group order."""
    import ctypes, random
    from oracle.ref_py import blindbid as bb, r1cs
    L = r1cs.L

    class ValuesOnly(r1cs.Prover):
        def __init__(self, values):
            r1cs._CSBase.__init__(self)
            self.v, self.vb = [x % L for x in values], []
            self.aL, self.aR, self.aO = [], [], []

    rnd = random.Random(81)
    mimc = bb.mimc_constants(bb.MIMC_ROUNDS)
    mimc_raw = b"".join(int(c).to_bytes(32, "little") for c in mimc)
    lib.hc_witness_gates.restype = ctypes.c_int
    for n, toggle in ((1, 0), (2, 1), (8, 3), (8, 2 ** 32 + 3), (13, 12), (40, 0)):
        seven = [rnd.randrange(L) for _ in range(7)]
        seven[0] = L + 5  # d is reduced on the way in
        items = [rnd.getrandbits(256) for _ in range(n)]
        raw = b"".join(x.to_bytes(32, "little") for x in seven) + b"".join(x.to_bytes(32, "little") for x in items) + toggle.to_bytes(8, "little")
        cap = 2048
        a, b = ctypes.create_string_buffer(3 * cap * 32), ctypes.create_string_buffer(3 * cap * 32)
        n_mul = lib.hc_witness_gates(n, raw, mimc_raw, a, b, cap)
        assert n_mul == 4 * 4 * bb.MIMC_ROUNDS + 3 * n + 2
        assert a.raw[:3 * n_mul * 32] == b.raw[:3 * n_mul * 32]
        d, k, y, y_inv, q, z_img, seed = [x % L for x in seven]
        cs = ValuesOnly([d, k, y, y_inv] + [1 if i == toggle else 0 for i in range(n)])
        t_v = [(r1cs.COMMITTED, 4 + i) for i in range(n)]
        bb.proof_gadget(cs, r1cs.LC.of((r1cs.COMMITTED, 0)), r1cs.LC.of((r1cs.COMMITTED, 1)), r1cs.LC.of((r1cs.COMMITTED, 3)), r1cs.LC.of(q),
                        r1cs.LC.of(z_img), r1cs.LC.of(seed), mimc, t_v, [r1cs.LC.of((x & ((1 << 255) - 1)) % L) for x in items])  # Scalar::from_bits (bid.rs:27): bit 255 cleared
        assert cs.n_mul == n_mul
        want = b"".join(int(x).to_bytes(32, "little") for x in cs.aL + cs.aR + cs.aO)
        assert a.raw[:3 * n_mul * 32] == want, n


def test_bit_interleaved_words_rotate_as_the_wave_keccak_assumes(lib):
    """keccak_wave.h keeps every 64-bit state word as (even bits, odd bits): a 64-bit rotation must be a 32-bit rotation of each
    half, with the halves changing places for odd amounts -- the rule the lane shifts and the rho-pi gather addresses are built from.
    All 64 amounts on random and edge words."""
    import ctypes, random
    lib.hc_kw_interleave.argtypes = [ctypes.c_uint64, ctypes.c_int]
    lib.hc_kw_interleave.restype = ctypes.c_int
    rnd = random.Random(50)
    words = [0, 1, 2, 1 << 63, (1 << 64) - 1, 0x5555555555555555, 0xAAAAAAAAAAAAAAAA, 0x0123456789ABCDEF] + [rnd.getrandbits(64) for _ in range(200)]
    for x in words:
        for r in range(64):
            assert lib.hc_kw_interleave(x, r) == 1, (hex(x), r)


def test_wave_keccak_model_permutes_like_keccak_f(lib):
    """tests/host_check.cpp hc_kw_keccak_f: the one-wavefront Keccak of keccak_wave.h restated on arrays of 64 lanes with the
    PRODUCT's lane tables (layout, theta / rho shift amounts, gather addresses, per-round iota vectors).  It must be Keccak-f[1600]."""
    import ctypes
    for seed in (b"a", b"bb", b"ccc", b""):
        st = bytearray(hashlib.shake_256(b"kw" + seed).digest(200)) if seed else bytearray(200)
        want = bytearray(st)
        merlin.keccak_f1600(want)
        buf = ctypes.create_string_buffer(bytes(st), 200)
        lib.hc_kw_keccak_f(buf)
        assert buf.raw == bytes(want)


def test_keccak_and_merlin(lib):
    st = bytearray(hashlib.shake_256(b"st").digest(200))
    st2 = bytearray(st)
    merlin.keccak_f1600(st2)
    buf = ctypes.create_string_buffer(bytes(st), 200)
    lib.hc_keccak_f(buf)
    assert buf.raw == bytes(st2)
    o = ctypes.create_string_buffer(32)
    lib.hc_merlin_kat(b"test protocol", 13, b"some label", 10, b"some data", 9, b"challenge", 9, o, 32)
    assert o.raw.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"  # merlin's own test vector
    for mlen in [0, 1, 150, 166, 167, 400]:
        msg = hashlib.shake_256(b"m%d" % mlen).digest(mlen) if mlen else b""
        n = 200
        o = ctypes.create_string_buffer(n)
        lib.hc_merlin_kat(b"BlindBidProofGadget", 19, b"lbl", 3, msg, mlen, b"ch", 2, o, n)
        t = merlin.Transcript(b"BlindBidProofGadget")
        t.append_message(b"lbl", msg)
        assert t.challenge_bytes(b"ch", n) == o.raw
        o = ctypes.create_string_buffer(2 * n)
        lib.hc_merlin_rng(b"BlindBidProofGadget", 19, b"v_blinding", 10, msg, mlen, b"\x07" * 32, o, n)
        t = merlin.Transcript(b"BlindBidProofGadget")
        r = t.build_rng([(b"v_blinding", msg)], b"\x07" * 32)
        assert r.fill_bytes(n) + r.fill_bytes(n) == o.raw
