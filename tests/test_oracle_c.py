"""not-gpu tier: the C restatement (oracle/c/bbp_oracle.c, the timed CPU baseline) against the big-int oracle and the golden
fixtures -- two independent restatements (Python big-ints; C 64-bit limbs) must agree byte for byte."""
import hashlib
import random

import pytest

from oracle.ref_py import blindbid as bb, ristretto as rs
from tests import oracle_c

L = rs.L
b32 = lambda x: x.to_bytes(32, "little")


@pytest.fixture(scope="module")
def oc(built):
    return oracle_c.load(built.build_oracle())


def test_scalar_primitives(oc):
    rnd = random.Random(5)
    for _ in range(200):
        a, b = rnd.randrange(L), rnd.randrange(L)
        assert oc.sc_op(0, b32(a), b32(b)) == (a + b) % L
        assert oc.sc_op(1, b32(a), b32(b)) == (a - b) % L
        assert oc.sc_op(2, b32(a), b32(b)) == a * b % L
        w = rnd.getrandbits(512)
        assert oc.sc_op(4, w.to_bytes(64, "little")) == w % L
        v = rnd.getrandbits(256)
        assert oc.sc_op(5, b32(v)) == (v & (2**255 - 1)) % L
    for a in [1, 2, L - 1, rnd.randrange(L)]:
        assert oc.sc_op(3, b32(a)) == pow(a, L - 2, L)


def test_setup_and_codec(oc, golden):
    kat = golden("setup_kat.json")
    assert oc.merlin_kat(b"test protocol", b"some label", b"some data", b"challenge", 32).hex() == \
        "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    assert oc.generator(0).hex() == kat["B_blinding"] and oc.generator(4097).hex() == kat["B"]
    assert hashlib.sha256(b"".join(oc.generator(1 + i) for i in range(2048))).hexdigest() == kat["G_sha256"]
    assert hashlib.sha256(b"".join(oc.generator(2049 + i) for i in range(2048))).hexdigest() == kat["H_sha256"]
    assert [oc.mimc_constant(i).hex() for i in range(90)] == kat["mimc_c"]
    rnd = random.Random(6)
    for i in range(10):
        u = hashlib.sha512(b"q%d" % i).digest()
        assert oc.from_uniform(u) == rs.encode(rs.from_uniform_bytes(u))
        s = rnd.randrange(L)
        assert oc.scalarmult(b32(s), oc.generator(5 + i)) == rs.encode(rs.pt_mul(s, rs.decode(oc.generator(5 + i))))


def test_msm_all_code_paths(oc):
    """Straus (< 190 terms) and each Pippenger window width against the big-int MSM."""
    pc, bp = bb.gens(2048)
    rnd = random.Random(7)
    for n, layout in [(1, 1), (3, 0), (41, 0), (191, 0), (401, 0), (600, 1), (901, 0)]:
        sc = [rnd.randrange(L) for _ in range(n)]
        sc[0] = 0
        m = (n - 1) // 2 if layout == 0 else n - 1
        bases = [pc.B_blinding] + bp.G[:m] + (bp.H[:m] if layout == 0 else [])
        assert oc.msm_layout(b"".join(b32(s) for s in sc), n, layout) == rs.encode(rs.msm(sc, bases)), n


def test_witness(oc):
    w = bb.witness(1, 2, 3)
    assert oc.witness(b32(1) + b32(2) + b32(3)) == b"".join(rs.sc_bytes(w[k]) for k in ["m", "x", "y", "y_inv", "q", "z_img"])


def _run_case(oc, c):
    s7 = b"".join(bytes.fromhex(c[k]) for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"])
    pub = b"".join(bytes.fromhex(p) for p in c["pub_list"])
    rc, rec, tr = oc.prove(s7, pub, c["toggle"], bytes.fromhex(c["entropy"]), c["rounds"], c["cap"], True)
    assert rc == 0
    for k in ["y", "z", "u", "x", "w"]:
        assert tr[k] == c["trace"][k], k
    assert tr["u_ipp"][:len(c["trace"]["u_ipp"])] == c["trace"]["u_ipp"]
    assert tr["n_mul"] == c["trace"]["n_mul"] and tr["n_constraints"] == c["trace"]["n_constraints"]
    assert rec.hex() == c["record"], c["name"]
    args = (bytes.fromhex(c["q"]), bytes.fromhex(c["z_img"]), bytes.fromhex(c["seed"]), pub, c["rounds"], c["cap"])
    assert oc.verify(rec, *args) == 0
    assert oc.verify(rec, *args, entropy32=b"\x11" * 32) == 0
    for pos in (1, 40, 1 + 32 * 8 + 3, c["proof_len"] - 40, c["proof_len"] + 5, len(rec) - 1):
        bad = bytearray(rec)
        bad[pos] ^= 1
        assert oc.verify(bytes(bad), *args) in (1, 3), pos
    assert oc.verify(rec, bytes.fromhex(c["z_img"]), *args[1:]) == 1          # wrong score
    assert oc.verify(rec[:-1], *args) == 3                                      # truncated
    assert oc.verify(b"\x02" + rec[1:], *args) == 3                             # bad version byte
    nc = bytearray(rec)
    nc[1 + 32 * 8:1 + 32 * 9] = b"\xff" * 32                                    # non-canonical t_x
    assert oc.verify(bytes(nc), *args) == 3


def test_small_golden(oc, golden):
    for c in golden("proofs_small.json")["small"]:
        _run_case(oc, c)


def test_full_golden(oc, golden):
    for c in golden("proofs_full.json")["full"]:
        _run_case(oc, c)


def test_noncanonical_bid_list_golden(oc, golden):
    """SURVEY.md 8a row a9 (Scalar::from_bits, src/blindbid/bid.rs:20-29, verify.rs:112-116): the fixture's list holds l, l + 1,
    2^255 - 1 and values with bit 255 set as RAW bytes; the C restatement reproduces the big-int oracle's bytes, and the list
    reduced by hand is the same statement."""
    c = golden("proofs_noncanonical.json")["noncanonical"][0]
    raw = [bytes.fromhex(p) for p in c["pub_list"]]
    vals = [int.from_bytes(b, "little") for b in raw]
    assert L in vals and L + 1 in vals and 2**255 - 1 in vals and any(v >> 255 for v in vals)
    _run_case(oc, c)
    s7 = b"".join(bytes.fromhex(c[k]) for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"])
    reduced = b"".join(rs.sc_bytes(rs.sc_from_bits(b)) for b in raw)
    rc, rec = oc.prove(s7, reduced, c["toggle"], bytes.fromhex(c["entropy"]))
    assert rc == 0 and rec.hex() == c["record"]


def test_bad_args(oc):
    s7 = bytes(7 * 32)
    assert oc.prove(s7, b"", 0, bytes(32 * 5))[0] == 4
    assert oc.prove(s7, bytes(64), 2, bytes(32 * 7))[0] == 4   # toggle >= N
    rc, _ = oc.prove(s7, bytes(32 * 203), 0, bytes(32 * (4 + 203) + 32))
    assert rc == 2                                             # N = 203 -> 2051 multipliers > 2048 (InvalidGeneratorsLength)


def test_threaded_batch_matches_single(oc, golden):
    c = golden("proofs_small.json")["small"][0]
    s7 = b"".join(bytes.fromhex(c[k]) for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"])
    pub = b"".join(bytes.fromhex(p) for p in c["pub_list"])
    one = s7 + pub + (c["toggle"]).to_bytes(8, "little")
    B = 6
    # oc_prove_many writes fixed 1121-byte proof slots; use the full-size circuit only in the GPU/bench tiers
    full = golden("proofs_full.json")["full"][1]
    s7f = b"".join(bytes.fromhex(full[k]) for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"])
    pubf = b"".join(bytes.fromhex(p) for p in full["pub_list"])
    inp = (s7f + pubf + (full["toggle"]).to_bytes(8, "little")) * B
    out, st = oc.prove_many(inp, bytes.fromhex(full["entropy"]) * B, B, full["N"], threads=4)
    assert st == [0] * B
    stride = 1121 + 32 * (4 + full["N"])
    assert all(out[i * stride:(i + 1) * stride].hex() == full["record"] for i in range(B))
    vin = b"".join(out[i * stride:(i + 1) * stride] + bytes.fromhex(full["q"]) + bytes.fromhex(full["z_img"]) + bytes.fromhex(full["seed"]) + pubf
                   for i in range(B))
    vin = bytearray(vin)
    vin[2 * (stride + 96 + 32 * full["N"]) + 50] ^= 4   # corrupt proof #2
    assert oc.verify_many(bytes(vin), B, full["N"], threads=4) == [0, 0, 1, 0, 0, 0]
    del one
