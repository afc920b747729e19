"""not-gpu tier: the server's TLV / request parsers (server/tlv.h, server/wire.h) under AddressSanitizer + UBSan.
Well-formed requests built by the Python restatement (tests/uds_client.py) must parse to exactly the fields that went in; every
truncation of them and thousands of random mutations must be rejected or parsed WITHOUT a sanitizer report -- a malformed frame
from the socket must never read out of bounds in the process that owns the GPU context."""
import random
import subprocess

import pytest

from tests import uds_client as uc


@pytest.fixture(scope="module")
def fuzz_bin(built):
    return built.build_wire_fuzz()


def _run(fuzz_bin, lines):
    p = subprocess.run([fuzz_bin], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, "sanitizer report or crash:\n" + p.stderr[-3000:]
    out = p.stdout.splitlines()
    assert "BROKEN FRAME" not in out
    return [ln for ln in out if ln.startswith(("ok", "err"))]


def _prove_body(rnd, n):
    s7 = bytes(rnd.getrandbits(8) for _ in range(224))
    pub = bytes(rnd.getrandbits(8) for _ in range(32 * n))
    toggle = rnd.getrandbits(64)
    frame = uc.prove_request(s7, pub, toggle)
    body, _ = uc.parse(frame)
    return body[1:], (n, toggle, s7, pub)


def _verify_body(rnd, n, proof_len=1121):
    proof = bytes(rnd.getrandbits(8) for _ in range(proof_len))
    c = [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(4)]
    t = [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(n)]
    blob = uc.tlv(proof) + uc.tlv_list(c) + uc.tlv_list(t)
    sc, z, sd = (bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(3))
    pub = bytes(rnd.getrandbits(8) for _ in range(32 * n))
    body, _ = uc.parse(uc.verify_request(blob, sc, z, sd, pub))
    return body[1:], (n, proof + b"".join(c) + b"".join(t), sc, z, sd, pub)


def test_wellformed_requests_parse_to_their_fields(fuzz_bin):
    rnd = random.Random(1)
    lines, exp = [], []
    for n in (1, 2, 8, 57, 202):
        b, f = _prove_body(rnd, n)
        lines.append("P " + b.hex())
        exp.append("ok %d %d %s %s" % (f[0], f[1], f[2].hex(), f[3].hex()))
        for pl in (1121, 1217, 300):      # the proof element is opaque to the parser: any length goes through to the engine
            b, f = _verify_body(rnd, n, pl)
            lines.append("V " + b.hex())
            exp.append("ok %d %s %s %s %s %s" % (f[0], f[1].hex(), f[2].hex(), f[3].hex(), f[4].hex(), f[5].hex()))
    assert _run(fuzz_bin, lines) == exp
    b, _ = _prove_body(rnd, 203)              # one bid too many for 2048 generators
    assert _run(fuzz_bin, ["P " + b.hex()])[0].startswith("err")


def test_every_truncation_is_rejected_cleanly(fuzz_bin):
    rnd = random.Random(2)
    pb, _ = _prove_body(rnd, 3)
    vb, _ = _verify_body(rnd, 3)
    lines = ["P " + pb[:k].hex() for k in range(len(pb))] + ["V " + vb[:k].hex() for k in range(0, len(vb), 7)]
    out = _run(fuzz_bin, lines)
    assert len(out) == len(lines) and all(o.startswith("err") for o in out)


def test_random_mutations_never_trip_the_sanitizers(fuzz_bin):
    rnd = random.Random(3)
    lines = []
    for i in range(3000):
        kind = "PV"[i & 1]
        body = bytearray((_prove_body if kind == "P" else _verify_body)(rnd, rnd.choice((1, 2, 5, 9)))[0])
        for _ in range(rnd.randrange(1, 6)):
            op = rnd.randrange(4)
            pos = rnd.randrange(len(body))
            if op == 0:
                body[pos] = rnd.getrandbits(8)
            elif op == 1:
                del body[pos:pos + rnd.randrange(1, 40)]
            elif op == 2:
                body[pos:pos] = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 20)))
            else:
                body[pos] = rnd.choice((0, 1, 2, 4, 8, 0xff))     # plausible and implausible width bytes
            if not body:
                body = bytearray(b"\x01")
        lines.append(kind + " " + bytes(body).hex())
    lines += ["P " + bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 400))).hex() for _ in range(500)]
    lines += ["V 08" + "ff" * 8 + "00" * 10, "V 04ffffffff", "P 02ffff" + "00" * 100]          # lengths far beyond the buffer
    out = _run(fuzz_bin, lines)
    assert len(out) == len(lines)
