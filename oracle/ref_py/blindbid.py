"""ORACLE (test infrastructure, never shipped): restatement of the in-tree reference files
  src/gadgets.rs (entire), src/blindbid/mod.rs:7-40, src/blindbid/proof.rs:36-91,
  src/blindbid/verify.rs:47-89, src/blindbid/bid.rs:20-29
on top of the oracle's r1cs / ristretto / merlin restatements.
`rounds` / `cap` are parameters only so tests can run reduced circuits; the reference fixes
MIMC_ROUNDS = 90 (src/gadgets.rs:4) and BulletproofGens::new(2048, 1) (src/blindbid/mod.rs:36).
"""
import hashlib

from . import ristretto as rs
from .merlin import Transcript
from .r1cs import (LC, BulletproofGens, PedersenGens, Prover, R1CSProof, Verifier, COMMITTED,
                   VerificationError, FormatError, InvalidGeneratorsLength)

L = rs.L
MIMC_ROUNDS = 90
GENS_CAPACITY = 2048


def mimc_constants(rounds=MIMC_ROUNDS):
    """src/blindbid/mod.rs:7-24: SHA-512 chain seeded by b"blind bid"."""
    out = []
    h = hashlib.sha512(b"blind bid").digest()
    for _ in range(rounds):
        c = rs.sc_wide(h)
        out.append(c)
        h = hashlib.sha512(rs.sc_bytes(c)).digest()
    return out


_gens_cache = {}


def gens(cap=GENS_CAPACITY):
    """src/blindbid/mod.rs:34-40 generate_cs_transcript (generators part), cached: they are constants."""
    if cap not in _gens_cache:
        _gens_cache[cap] = (PedersenGens(), BulletproofGens(cap))
    return _gens_cache[cap]


def new_transcript():
    return Transcript(b"BlindBidProofGadget")


# ---- native (non-circuit) arithmetic: what the Go caller computes upstream ---------------------------
def mimc_hash(left, right, constants):
    """Native image of mimc_gadget (src/gadgets.rs:45-67)."""
    x = left % L
    for c in constants:
        a = (x + right + c) % L
        x = pow(a, 7, L)
    return (x + right) % L


def witness(d, k, seed, constants=None):
    """m, x, y, y_inv, q, z_img as wired by proof_gadget (src/gadgets.rs:20-33)."""
    constants = constants or mimc_constants()
    m = mimc_hash(k, 0, constants)
    x = mimc_hash(d, m, constants)
    y = mimc_hash(seed, x, constants)
    z = mimc_hash(seed, m, constants)
    y_inv = rs.sc_inv(y)
    q = d * y_inv % L
    return dict(m=m, x=x, y=y, y_inv=y_inv, q=q, z_img=z)


# ---- gadgets (src/gadgets.rs) ------------------------------------------------------------------------
def mimc_gadget(cs, left, right, constants):
    x = LC.of(left)
    key = LC.of(right)
    for c in constants:
        a = x + key + c
        _, _, a2 = cs.multiply(a, a)
        _, _, a3 = cs.multiply(LC.of(a2), a)
        _, _, a4 = cs.multiply(LC.of(a2), LC.of(a2))
        _, _, a7 = cs.multiply(LC.of(a4), LC.of(a3))
        x = LC.of(a7)
    return x + key


def boolean_gadget(cs, a1):
    a = LC.of(a1)
    _, _, c = cs.multiply(a, LC.of(1) - a)
    cs.constrain(LC.of(c))


def one_of_many_gadget(cs, x, toggle, items):
    n = len(toggle)
    for tv in toggle:
        boolean_gadget(cs, LC.of(tv))
    tsum = [LC.of(toggle[0])]
    for i in range(1, n):
        tsum.append(tsum[i - 1] + toggle[i])
    for i in range(1, n):
        prev, cur, cur_sum = tsum[i - 1], toggle[i], tsum[i]
        tsum[i] = tsum[i - 1] + toggle[i]
        cs.constrain(prev + cur - cur_sum)
    cs.constrain(tsum[n - 1] - 1)
    for i in range(n):
        _, _, left = cs.multiply(items[i], LC.of(toggle[i]))
        _, _, right = cs.multiply(LC.of(toggle[i]), x)
        cs.constrain(LC.of(left) - right)


def score_gadget(cs, d, y, y_inv, q):
    _, _, one_var = cs.multiply(y, y_inv)
    cs.constrain(LC.of(one_var) - 1)
    _, _, q_var = cs.multiply(d, y_inv)
    cs.constrain(LC.of(q) - q_var)


def proof_gadget(cs, d, k, y_inv, q, z_img, seed, constants, toggle, items):
    m = mimc_gadget(cs, k, LC.of(0), constants)
    x = mimc_gadget(cs, d, m, constants)
    one_of_many_gadget(cs, x, toggle, items)
    y = mimc_gadget(cs, seed, x, constants)
    z = mimc_gadget(cs, seed, m, constants)
    cs.constrain(LC.of(z_img) - z)
    score_gadget(cs, d, y, y_inv, q)


# ---- drivers -------------------------------------------------------------------------------------------
class Proof:
    def __init__(self, proof, commitments, t_c):
        self.proof, self.commitments, self.t_c = proof, commitments, t_c

    def to_record(self):
        """Raw record of SURVEY.md 8b: R1CSProof bytes || 4x32 commitments || Nx32 t_c."""
        return self.proof.to_bytes() + b"".join(self.commitments) + b"".join(self.t_c)

    @staticmethod
    def from_record(buf, n_items, proof_len=None):
        tail = 32 * (4 + n_items)
        if len(buf) < tail + 1:
            raise FormatError("record")
        pl = len(buf) - tail if proof_len is None else proof_len
        pr = R1CSProof.from_bytes(buf[:pl])
        c = [buf[pl + 32 * i:pl + 32 * i + 32] for i in range(4)]
        t = [buf[pl + 128 + 32 * i:pl + 128 + 32 * i + 32] for i in range(n_items)]
        return Proof(pr, c, t)


def prove(d, k, y, y_inv, q, z_img, seed, pub_list, toggle, entropy, rounds=MIMC_ROUNDS, cap=GENS_CAPACITY,
          trace=None):
    """src/blindbid/proof.rs:36-91.  `entropy` = (4+N) blinding scalars (32 B each) || 32 B rng seed
     replacing thread_rng."""
    n = len(pub_list)
    assert len(entropy) == 32 * (4 + n) + 32
    pc, bp = gens(cap)
    constants = mimc_constants(rounds)
    prover = Prover(pc, new_transcript())
    bl = [rs.sc_wide(entropy[32 * i:32 * i + 32] + bytes(32)) for i in range(4 + n)]
    commitments, vars_ = [], []
    for i, v in enumerate([d, k, y, y_inv]):
        V, var = prover.commit(v, bl[i])
        commitments.append(V)
        vars_.append(var)
    t_c, t_v = [], []
    for i in range(n):
        V, var = prover.commit(1 if i == toggle else 0, bl[4 + i])
        t_c.append(V)
        t_v.append(var)
    items = [LC.of(b % L) for b in pub_list]
    proof_gadget(prover, LC.of(vars_[0]), LC.of(vars_[1]), LC.of(vars_[3]), LC.of(q), LC.of(z_img), LC.of(seed),
                 constants, t_v, items)
    pr = prover.prove(bp, entropy[32 * (4 + n):], trace)
    if trace is not None:
        trace["a_L_first"] = [rs.sc_bytes(v).hex() for v in prover.aL[:4]]
    return Proof(pr, commitments, t_c)


def verify(proof: Proof, score, z_img, seed, pub_list, entropy32=bytes(32), rounds=MIMC_ROUNDS, cap=GENS_CAPACITY):
    """src/blindbid/verify.rs:47-89. Raises VerificationError / InvalidGeneratorsLength; returns True on accept."""
    pc, bp = gens(cap)
    constants = mimc_constants(rounds)
    ver = Verifier(new_transcript())
    vars_ = [ver.commit(c) for c in proof.commitments]
    t_v = [ver.commit(c) for c in proof.t_c]
    items = [LC.of(b % L) for b in pub_list]
    proof_gadget(ver, LC.of(vars_[0]), LC.of(vars_[1]), LC.of(vars_[3]), LC.of(score), LC.of(z_img), LC.of(seed),
                 constants, t_v, items)
    return ver.verify(proof.proof, pc, bp, entropy32)
