#!/usr/bin/env python3
"""Writes tools/ref_crosscheck/vectors.txt from the committed golden fixtures (full-size cases only), one line per vector in the
format src/main.rs documents, plus tampered variants with the verdict this repository's verifiers give (third column of
expected.txt).  Run anywhere: it reads tests/golden/*.json only."""
import json, os
HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "..", "..", "tests", "golden")
lines, expected = [], []


def emit(name, c, record, score, verdict):
    pubs = " ".join(c["pub_list"])
    lines.append("%s %d %d %s %s %s %s %s" % (name, c["N"], c["toggle"], record, score, c["z_img"], c["seed"], pubs))
    expected.append("%s %s" % (name, verdict))


for fn, key in (("proofs_full.json", "full"), ("proofs_noncanonical.json", "noncanonical")):
    for c in json.load(open(os.path.join(G, fn)))[key]:
        emit(c["name"], c, c["record"], c["q"], "accept")
        bad = bytearray(bytes.fromhex(c["record"]))
        bad[200] ^= 1
        emit(c["name"] + "_flipped_bit", c, bytes(bad).hex(), c["q"], "reject")
        emit(c["name"] + "_wrong_score", c, c["record"], c["z_img"], "reject")
        nc = bytearray(bytes.fromhex(c["record"]))
        nc[1 + 32 * 8:1 + 32 * 9] = b"\xff" * 32
        emit(c["name"] + "_noncanonical_t_x", c, bytes(nc).hex(), c["q"], "format-error")
open(os.path.join(HERE, "vectors.txt"), "w").write("\n".join(lines) + "\n")
open(os.path.join(HERE, "expected.txt"), "w").write("\n".join(expected) + "\n")
print("wrote %d vectors" % len(lines))
