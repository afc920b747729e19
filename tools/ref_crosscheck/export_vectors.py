#!/usr/bin/env python3
"""Writes tools/ref_crosscheck/vectors.txt from the committed golden fixtures (full-size cases only), one line per vector in the
format src/main.rs documents, plus tampered variants with the verdict this repository's verifiers give (third column of
expected.txt); and frames.txt: the same golden cases as WIRE bytes -- opcode-1 request body, proof blob, opcode-2 request body and
reply frame exactly as this repository's encoders emit them (tests/uds_client.py, the Python twin of server/tlv.h + server/wire.h) --
for `bbp-ref-crosscheck frames`, which compares them with what the reference's own TlvWriter produces and feeds them to the
reference's parsers.  Run anywhere: it reads tests/golden/*.json only."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tests import uds_client as uc
G = os.path.join(HERE, "..", "..", "tests", "golden")
lines, expected, frames, traces = [], [], [], []


def emit(name, c, record, score, verdict):
    pubs = " ".join(c["pub_list"])
    lines.append("%s %d %d %s %s %s %s %s" % (name, c["N"], c["toggle"], record, score, c["z_img"], c["seed"], pubs))
    expected.append("%s %s" % (name, verdict))


for fn, key in (("proofs_full.json", "full"), ("proofs_noncanonical.json", "noncanonical")):
    for c in json.load(open(os.path.join(G, fn)))[key]:
        emit(c["name"], c, c["record"], c["q"], "accept")
        if key == "full":  # wire vectors: canonical inputs only (the reference's serde Scalar refuses others before any framing matters)
            n, rec = c["N"], bytes.fromhex(c["record"])
            s7 = b"".join(bytes.fromhex(c[k]) for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"])
            pub = b"".join(bytes.fromhex(p) for p in c["pub_list"])
            plen = len(rec) - 32 * (4 + n)
            blob = uc.tlv(rec[:plen]) + uc.tlv_list([rec[plen + 32 * i:plen + 32 * i + 32] for i in range(4)]) + \
                uc.tlv_list([rec[plen + 128 + 32 * i:plen + 128 + 32 * i + 32] for i in range(n)])
            prove_body = uc.parse(uc.prove_request(s7, pub, c["toggle"]))[0][1:]                      # request[1..], proof.rs:97
            verify_body = uc.parse(uc.verify_request(blob, bytes.fromhex(c["q"]), bytes.fromhex(c["z_img"]), bytes.fromhex(c["seed"]), pub))[0][1:]
            frames.append(" ".join(["wire", c["name"], str(n), str(c["toggle"])] + [c[k] for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"]] +
                                   c["pub_list"] + [c["record"], prove_body.hex(), blob.hex(), verify_body.hex(), uc.tlv(blob).hex()]))
            # `trace`: inputs + the injected entropy + the record + this repository's Fiat-Shamir challenges in transcript order
            t = c.get("trace") or {}
            chal = [t.get(k, "") for k in ("y", "z", "u", "x", "w")] + list(t.get("u_ipp", []))
            traces.append(" ".join(["trace", c["name"], str(n), str(c["toggle"])] + [c[k] for k in ["d", "k", "y", "y_inv", "q", "z_img", "seed"]] + c["pub_list"] +
                                   [c["entropy"], c["record"]] + [x for x in chal if x]))
        bad = bytearray(bytes.fromhex(c["record"]))
        bad[200] ^= 1
        emit(c["name"] + "_flipped_bit", c, bytes(bad).hex(), c["q"], "reject")
        emit(c["name"] + "_wrong_score", c, c["record"], c["z_img"], "reject")
        nc = bytearray(bytes.fromhex(c["record"]))
        nc[1 + 32 * 8:1 + 32 * 9] = b"\xff" * 32
        emit(c["name"] + "_noncanonical_t_x", c, bytes(nc).hex(), c["q"], "format-error")
open(os.path.join(HERE, "vectors.txt"), "w").write("\n".join(lines) + "\n")
open(os.path.join(HERE, "expected.txt"), "w").write("\n".join(expected) + "\n")
open(os.path.join(HERE, "frames.txt"), "w").write("\n".join(frames) + "\n")
open(os.path.join(HERE, "trace.txt"), "w").write("\n".join(traces) + "\n")
print("wrote %d vectors, %d wire cases, %d prover traces" % (len(lines), len(frames), len(traces)))
