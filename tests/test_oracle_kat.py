"""not-gpu tier: pins the big-int oracle (oracle/ref_py) to PUBLIC known answers -- RFC 9496 ristretto255 vectors, merlin's
own test vector, hashlib SHA-3 -- and to the committed golden fixtures; then prove -> verify round trips and tamper tests.
The reference holds no tests or fixtures of its own and cannot be built here (F4)."""
import hashlib

import pytest

from oracle.ref_py import blindbid as bb, merlin, r1cs, ristretto as rs

L = rs.L

RFC9496_MULTIPLES = [
    "0000000000000000000000000000000000000000000000000000000000000000", "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
    "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919", "94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259",
    "da80862773358b466ffadfe0b3293ab3d9fd53c5ea6c955358f568322daf6a57", "e882b131016b52c1d3337080187cf768423efccbb517bb495ab812c4160ff44e",
    "f64746d3c92b13050ed8d80236a7f0007c3b3f962f5ba793d19a601ebb1df403", "44f53520926ec81fbd5a387845beb7df85a96a24ece18738bdcfa6a7822a176d",
    "903293d8f2287ebe10e2374dc1a53e0bc887e592699f02d077d5263cdd55601c", "02622ace8f7303a31cafc63f8fc48fdc16e1c8c8d234b2f0d6685282a9076031",
    "20706fd788b2720a1ed2a5dad4952b01f413bcf0e7564de8cdc816689e2db95f", "bce83f8ba5dd2fa572864c24ba1810f9522bc6004afe95877ac73241cafdab42",
    "e4549ee16b9aa03099ca208c67adafcafa4c3f3e4e5303de6026e3ca8ff84460", "aa52e000df2e16f55fb1032fc33bc42742dad6bd5a8fc0be0167436c5948501f",
    "46376b80f409b29dc2b5f6f0c52591990896e5716f41477cd30085ab7f10301e", "e0c418f7c8d9c4cdd7395b93ea124f3ad99021bb681dfc3302a9d99a2e53e64e"]


def test_rfc9496_multiples_of_generator():
    acc = rs.IDENT
    for e in RFC9496_MULTIPLES:
        assert rs.encode(acc).hex() == e
        assert rs.pt_eq(rs.decode(bytes.fromhex(e)), acc)
        acc = rs.pt_add(acc, rs.BASEPOINT)
    assert rs.encode(rs.pt_mul(L, rs.BASEPOINT)) == bytes(32)


def test_keccak_against_hashlib():
    def sha3_256(m):
        st, rate = bytearray(200), 136
        m = bytearray(m) + b"\x06"
        while len(m) % rate:
            m += b"\x00"
        m[-1] |= 0x80
        for i in range(0, len(m), rate):
            for j in range(rate):
                st[j] ^= m[i + j]
            merlin.keccak_f1600(st)
        return bytes(st[:32])
    for msg in [b"", b"abc", b"x" * 135, b"y" * 136, b"z" * 500]:
        assert sha3_256(msg) == hashlib.sha3_256(msg).digest()


def test_merlin_published_vector():
    t = merlin.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_pedersen_blinding_generator_published():
    assert rs.encode(r1cs.PedersenGens().B_blinding).hex() == "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"


def test_setup_matches_golden(golden):
    kat = golden("setup_kat.json")
    g = r1cs.BulletproofGens(3)
    assert [rs.encode(p).hex() for p in g.G] == kat["G"] and [rs.encode(p).hex() for p in g.H] == kat["H"]
    c = bb.mimc_constants()
    assert [rs.sc_bytes(x).hex() for x in c] == kat["mimc_c"]
    assert hashlib.sha256(b"".join(rs.sc_bytes(x) for x in c)).hexdigest() == kat["mimc_sha256"]
    # SURVEY.md App. B values
    assert kat["mimc_c"][0] == "cfff56ca78e2dd3e3fd7664f7568b578b02aafb564ad816afce960c98524520d"
    assert kat["mimc_sha256"] == "74f3f9e9ee3fa730557edfbcd4fb2c4c9ba13ce3699d92e2ec79d93398dc3f1a"


def test_witness_kat():
    w = bb.witness(1, 2, 3)
    assert rs.sc_bytes(w["m"]).hex() == "e9c12933df0565e65eabf6436296300b1a8f9eb7355ebc1af7d0661cd8f23805"
    assert rs.sc_bytes(w["z_img"]).hex() == "b9c78d935706cae2cc9ca48b2bca76884358a7d0a4a2366605959ade22a6be07"
    assert w["y"] * w["y_inv"] % L == 1 and w["q"] == w["y_inv"] % L  # d = 1


def _case_inputs(c):
    f = lambda k: int.from_bytes(bytes.fromhex(c[k]), "little")
    pub = [int.from_bytes(bytes.fromhex(p), "little") for p in c["pub_list"]]
    return f, pub


def test_small_circuits_regenerate_golden_and_tamper(golden):
    for c in golden("proofs_small.json")["small"]:
        f, pub = _case_inputs(c)
        tr = {}
        pr = bb.prove(f("d"), f("k"), f("y"), f("y_inv"), f("q"), f("z_img"), f("seed"), pub, c["toggle"], bytes.fromhex(c["entropy"]),
                      c["rounds"], c["cap"], tr)
        assert pr.to_record().hex() == c["record"]
        assert tr["n_mul"] == 4 * 4 * c["rounds"] + 3 * c["N"] + 2
        assert tr["n_constraints"] == 2 * tr["n_mul"] + 3 + 3 * c["N"]
        rec = bytes.fromhex(c["record"])
        p2 = bb.Proof.from_record(rec, c["N"])
        assert bb.verify(p2, f("q"), f("z_img"), f("seed"), pub, bytes(32), c["rounds"], c["cap"])
        # accept must not depend on the verifier's own randomness
        assert bb.verify(p2, f("q"), f("z_img"), f("seed"), pub, b"\x5a" * 32, c["rounds"], c["cap"])
        with pytest.raises(r1cs.VerificationError):
            bb.verify(p2, (f("q") + 1) % L, f("z_img"), f("seed"), pub, bytes(32), c["rounds"], c["cap"])
        with pytest.raises(r1cs.VerificationError):
            bb.verify(p2, f("q"), (f("z_img") + 1) % L, f("seed"), pub, bytes(32), c["rounds"], c["cap"])
        with pytest.raises(r1cs.VerificationError):
            bb.verify(p2, f("q"), f("z_img"), (f("seed") + 1) % L, pub, bytes(32), c["rounds"], c["cap"])
        bad_pub = list(pub)
        bad_pub[c["toggle"]] = (bad_pub[c["toggle"]] + 1) % L
        with pytest.raises(r1cs.VerificationError):
            bb.verify(p2, f("q"), f("z_img"), f("seed"), bad_pub, bytes(32), c["rounds"], c["cap"])
        for pos in (1, 1 + 32 * 3 + 7, 1 + 32 * 8 + 3, c["proof_len"] - 40, c["proof_len"] + 5):
            r2 = bytearray(rec)
            r2[pos] ^= 0x01
            with pytest.raises((r1cs.VerificationError, r1cs.FormatError)):
                bb.verify(bb.Proof.from_record(bytes(r2), c["N"]), f("q"), f("z_img"), f("seed"), pub, bytes(32), c["rounds"], c["cap"])


def test_full_size_golden_shapes(golden):
    """Full 90-round circuit fixtures (generated offline: ~20 s each in big-int Python) have the sizes SURVEY.md F7 states
    and verify under the oracle."""
    for c in golden("proofs_full.json")["full"]:
        assert c["trace"]["n_mul"] == 1442 + 3 * c["N"] and c["trace"]["n_constraints"] == 2 * c["trace"]["n_mul"] + 3 + 3 * c["N"]
        assert c["proof_len"] == 1121 and len(c["trace"]["u_ipp"]) == 11
    c = golden("proofs_full.json")["full"][0]
    f, pub = _case_inputs(c)
    assert bb.verify(bb.Proof.from_record(bytes.fromhex(c["record"]), c["N"]), f("q"), f("z_img"), f("seed"), pub)


def test_format_errors():
    with pytest.raises(r1cs.FormatError):
        r1cs.R1CSProof.from_bytes(b"")
    with pytest.raises(r1cs.FormatError):
        r1cs.R1CSProof.from_bytes(b"\x02" + bytes(32 * 13))
    with pytest.raises(r1cs.FormatError):
        r1cs.R1CSProof.from_bytes(b"\x00" + bytes(32 * 12))
    with pytest.raises(r1cs.FormatError):  # non-canonical t_x
        r1cs.R1CSProof.from_bytes(b"\x00" + bytes(32 * 8) + b"\xff" * 32 + bytes(32 * 4))
