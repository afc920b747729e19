"""ORACLE (test infrastructure, never shipped): restatement of bulletproofs (git develop @4a05305,
v1.0.4, feature yoloproofs; Cargo.toml:23-26, Cargo.lock:65-67) R1CS prover / verifier and the
inner-product argument, as SURVEY.md App. A.3-A.8 restates them.  Crate source is not under
/root/reference; call sites: src/blindbid/proof.rs:50,57,62,88, src/blindbid/verify.rs:51,57,63,88,
src/gadgets.rs:53-62,80,84,119,123,128-130,138-139, src/blindbid/mod.rs:35-36.

parity unpinned by the reference itself (it holds no tests/fixtures): pinned instead by
public KATs of every primitive plus prove->verify round trips and tamper tests.
"""
import hashlib

from . import ristretto as rs
from .merlin import Transcript

L = rs.L


# ---- generators (A.2) ------------------------------------------------------------------------------
class PedersenGens:
    def __init__(self):
        self.B = rs.BASEPOINT
        self.B_blinding = rs.from_uniform_bytes(hashlib.sha3_512(rs.encode(rs.BASEPOINT)).digest())

    def commit(self, v, vb):
        return rs.pt_add(rs.pt_mul(v, self.B), rs.pt_mul(vb, self.B_blinding))


class BulletproofGens:
    """BulletproofGens::new(cap, 1): party 0 chains 'G' and 'H' (SHAKE256 'GeneratorsChain' || label)."""

    def __init__(self, cap):
        self.cap = cap
        self.G = self._chain(b"G", cap)
        self.H = self._chain(b"H", cap)

    @staticmethod
    def _chain(tag, n):
        label = tag + (0).to_bytes(4, "little")
        stream = hashlib.shake_256(b"GeneratorsChain" + label).digest(64 * n)
        return [rs.from_uniform_bytes(stream[64 * i:64 * i + 64]) for i in range(n)]


# ---- transcript protocol (A.1 tail) ----------------------------------------------------------------
class VerificationError(Exception):
    pass


class FormatError(Exception):
    pass


class InvalidGeneratorsLength(Exception):
    pass


def t_point(t, label, pbytes):
    t.append_message(label, pbytes)


def t_validate_point(t, label, pbytes):
    if pbytes == bytes(32):
        raise VerificationError("identity point")
    t.append_message(label, pbytes)


def t_scalar(t, label, s):
    t.append_message(label, rs.sc_bytes(s))


def t_challenge(t, label):
    return rs.sc_wide(t.challenge_bytes(label, 64))


# ---- constraint system (A.3) -----------------------------------------------------------------------
COMMITTED, MUL_L, MUL_R, MUL_O, ONE = "V", "L", "R", "O", "1"


class LC:
    """LinearCombination: list of (variable, coeff); variable = (kind, index)."""

    __slots__ = ("terms",)

    def __init__(self, terms=None):
        self.terms = list(terms or [])

    @staticmethod
    def of(x):
        if isinstance(x, LC):
            return LC(x.terms)
        if isinstance(x, tuple):
            return LC([(x, 1)])
        return LC([((ONE, 0), x % L)])

    def __add__(self, o):
        return LC(self.terms + LC.of(o).terms)

    def __sub__(self, o):
        return LC(self.terms + [(v, (-c) % L) for v, c in LC.of(o).terms])

    def __neg__(self):
        return LC([(v, (-c) % L) for v, c in self.terms])


class _CSBase:
    def __init__(self):
        self.constraints = []
        self.n_mul = 0

    def constrain(self, lc):
        self.constraints.append(lc)

    def flatten(self, z, m, with_wc):
        n = self.n_mul
        wL, wR, wO, wV, wc = [0] * n, [0] * n, [0] * n, [0] * m, 0
        e = z
        for lc in self.constraints:
            for (kind, i), c in lc.terms:
                if kind == MUL_L:
                    wL[i] = (wL[i] + e * c) % L
                elif kind == MUL_R:
                    wR[i] = (wR[i] + e * c) % L
                elif kind == MUL_O:
                    wO[i] = (wO[i] + e * c) % L
                elif kind == COMMITTED:
                    wV[i] = (wV[i] - e * c) % L
                elif with_wc:
                    wc = (wc - e * c) % L
            e = e * z % L
        return wL, wR, wO, wV, wc


def _exp_iter(x, n):
    out, e = [], 1
    for _ in range(n):
        out.append(e)
        e = e * x % L
    return out


def _ip(a, b):
    return sum(x * y for x, y in zip(a, b)) % L


# ---- inner-product proof (A.6) ---------------------------------------------------------------------
def ipp_create(t, Q, Gf, Hf, G, H, a, b, trace=None):
    n = len(a)
    assert n & (n - 1) == 0 and len(G) == len(H) == len(b) == len(Gf) == len(Hf) == n
    t.append_message(b"dom-sep", b"ipp v1")
    t.append_u64(b"n", n)
    G, H, a, b = list(G), list(H), list(a), list(b)
    Ls, Rs = [], []
    first = True
    while n != 1:
        n //= 2
        aL, aR, bL, bR = a[:n], a[n:], b[:n], b[n:]
        GL, GR, HL, HR = G[:n], G[n:], H[:n], H[n:]
        cL, cR = _ip(aL, bR), _ip(aR, bL)
        if first:
            Lp = rs.msm([x * g for x, g in zip(aL, Gf[n:])] + [x * h for x, h in zip(bR, Hf[:n])] + [cL], GR + HL + [Q])
            Rp = rs.msm([x * g for x, g in zip(aR, Gf[:n])] + [x * h for x, h in zip(bL, Hf[n:])] + [cR], GL + HR + [Q])
        else:
            Lp = rs.msm(aL + bR + [cL], GR + HL + [Q])
            Rp = rs.msm(aR + bL + [cR], GL + HR + [Q])
        Lb, Rb = rs.encode(Lp), rs.encode(Rp)
        Ls.append(Lb)
        Rs.append(Rb)
        t_point(t, b"L", Lb)
        t_point(t, b"R", Rb)
        u = t_challenge(t, b"u")
        ui = rs.sc_inv(u)
        if trace is not None:
            trace.setdefault("u_ipp", []).append(rs.sc_bytes(u).hex())
        a = [(x * u + ui * y) % L for x, y in zip(aL, aR)]
        b = [(x * ui + u * y) % L for x, y in zip(bL, bR)]
        if first:
            G = [rs.msm([ui * Gf[i], u * Gf[n + i]], [GL[i], GR[i]]) for i in range(n)]
            H = [rs.msm([u * Hf[i], ui * Hf[n + i]], [HL[i], HR[i]]) for i in range(n)]
        else:
            G = [rs.msm([ui, u], [GL[i], GR[i]]) for i in range(n)]
            H = [rs.msm([u, ui], [HL[i], HR[i]]) for i in range(n)]
        first = False
    return Ls, Rs, a[0], b[0]


def ipp_to_bytes(Ls, Rs, a, b):
    return b"".join(x + y for x, y in zip(Ls, Rs)) + rs.sc_bytes(a) + rs.sc_bytes(b)


def ipp_from_bytes(buf):
    """InnerProductProof::from_bytes."""
    if len(buf) % 32 != 0 or len(buf) // 32 < 2 or (len(buf) // 32 - 2) % 2 != 0:
        raise FormatError("ipp length")
    lg_n = (len(buf) // 32 - 2) // 2
    if lg_n >= 32:
        raise FormatError("ipp lg_n")
    Ls = [buf[64 * i:64 * i + 32] for i in range(lg_n)]
    Rs = [buf[64 * i + 32:64 * i + 64] for i in range(lg_n)]
    a = rs.sc_canonical(buf[64 * lg_n:64 * lg_n + 32])
    b = rs.sc_canonical(buf[64 * lg_n + 32:64 * lg_n + 64])
    if a is None or b is None:
        raise FormatError("ipp scalar")
    return Ls, Rs, a, b


def ipp_verification_scalars(Ls, Rs, n, t):
    lg_n = len(Ls)
    if lg_n >= 32 or n != (1 << lg_n):
        raise VerificationError("ipp size")
    t.append_message(b"dom-sep", b"ipp v1")
    t.append_u64(b"n", n)
    ch = []
    for Lb, Rb in zip(Ls, Rs):
        t_validate_point(t, b"L", Lb)
        t_validate_point(t, b"R", Rb)
        ch.append(t_challenge(t, b"u"))
    ch_inv = [rs.sc_inv(c) for c in ch]
    allinv = 1
    for c in ch_inv:
        allinv = allinv * c % L
    ch_sq = [c * c % L for c in ch]
    ch_inv_sq = [c * c % L for c in ch_inv]
    s = [allinv]
    for i in range(1, n):
        lg_i = i.bit_length() - 1
        s.append(s[i - (1 << lg_i)] * ch_sq[lg_n - 1 - lg_i] % L)
    return ch_sq, ch_inv_sq, s, ch


# ---- R1CS proof container (A.8) --------------------------------------------------------------------
class R1CSProof:
    FIELDS_P = ["A_I1", "A_O1", "S1", "A_I2", "A_O2", "S2", "T_1", "T_3", "T_4", "T_5", "T_6"]

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to_bytes(self):
        one_phase = self.A_I2 == bytes(32) and self.A_O2 == bytes(32) and self.S2 == bytes(32)
        out = bytearray()
        if one_phase:
            out += b"\x00" + self.A_I1 + self.A_O1 + self.S1
        else:
            out += b"\x01" + self.A_I1 + self.A_O1 + self.S1 + self.A_I2 + self.A_O2 + self.S2
        out += self.T_1 + self.T_3 + self.T_4 + self.T_5 + self.T_6
        out += rs.sc_bytes(self.t_x) + rs.sc_bytes(self.t_x_blinding) + rs.sc_bytes(self.e_blinding)
        out += ipp_to_bytes(self.L_vec, self.R_vec, self.a, self.b)
        return bytes(out)

    @staticmethod
    def from_bytes(buf):
        if len(buf) < 1:
            raise FormatError("empty")
        ver, body = buf[0], buf[1:]
        if len(body) % 32 != 0:
            raise FormatError("length")
        if ver == 0:
            minel = 3 + 5 + 3 + 2
        elif ver == 1:
            minel = 6 + 5 + 3 + 2
        else:
            raise FormatError("version")
        if len(body) // 32 < minel:
            raise FormatError("short")
        pos = [0]

        def take():
            v = body[pos[0]:pos[0] + 32]
            pos[0] += 32
            return v

        kw = {}
        kw["A_I1"], kw["A_O1"], kw["S1"] = take(), take(), take()
        if ver == 0:
            kw["A_I2"] = kw["A_O2"] = kw["S2"] = bytes(32)
        else:
            kw["A_I2"], kw["A_O2"], kw["S2"] = take(), take(), take()
        for k in ["T_1", "T_3", "T_4", "T_5", "T_6"]:
            kw[k] = take()
        for k in ["t_x", "t_x_blinding", "e_blinding"]:
            v = rs.sc_canonical(take())
            if v is None:
                raise FormatError("scalar")
            kw[k] = v
        kw["L_vec"], kw["R_vec"], kw["a"], kw["b"] = ipp_from_bytes(body[pos[0]:])
        return R1CSProof(**kw)


# ---- prover (A.4, A.5) -----------------------------------------------------------------------------
class Prover(_CSBase):
    def __init__(self, pc_gens, transcript):
        super().__init__()
        self.pc = pc_gens
        self.t = transcript
        self.t.append_message(b"dom-sep", b"r1cs v1")
        self.v, self.vb = [], []
        self.aL, self.aR, self.aO = [], [], []

    def commit(self, v, vb):
        i = len(self.v)
        self.v.append(v % L)
        self.vb.append(vb % L)
        V = rs.encode(self.pc.commit(v, vb))
        t_point(self.t, b"V", V)
        return V, (COMMITTED, i)

    def eval(self, lc):
        acc = 0
        for (kind, i), c in lc.terms:
            val = {MUL_L: self.aL, MUL_R: self.aR, MUL_O: self.aO, COMMITTED: self.v}.get(kind)
            acc += c * (1 if kind == ONE else val[i])
        return acc % L

    def multiply(self, left, right):
        left, right = LC.of(left), LC.of(right)
        l, r = self.eval(left), self.eval(right)
        i = self.n_mul
        self.n_mul += 1
        self.aL.append(l)
        self.aR.append(r)
        self.aO.append(l * r % L)
        self.constrain(left - (MUL_L, i))
        self.constrain(right - (MUL_R, i))
        return (MUL_L, i), (MUL_R, i), (MUL_O, i)

    def prove(self, bp_gens, entropy32, trace=None):
        t, pc = self.t, self.pc
        m = len(self.v)
        t.append_u64(b"m", m)
        rng = t.build_rng([(b"v_blinding", rs.sc_bytes(x)) for x in self.vb], entropy32)
        rnd = lambda: rs.sc_wide(rng.fill_bytes(64))
        n1 = self.n_mul
        if bp_gens.cap < n1:
            raise InvalidGeneratorsLength()
        ib, ob, sb = rnd(), rnd(), rnd()
        sL = [rnd() for _ in range(n1)]
        sR = [rnd() for _ in range(n1)]
        G, H = bp_gens.G, bp_gens.H
        A_I1 = rs.encode(rs.msm([ib] + self.aL + self.aR, [pc.B_blinding] + G[:n1] + H[:n1]))
        A_O1 = rs.encode(rs.msm([ob] + self.aO, [pc.B_blinding] + G[:n1]))
        S1 = rs.encode(rs.msm([sb] + sL + sR, [pc.B_blinding] + G[:n1] + H[:n1]))
        t_point(t, b"A_I1", A_I1)
        t_point(t, b"A_O1", A_O1)
        t_point(t, b"S1", S1)
        # no randomized constraints in the blind-bid circuit -> 1-phase
        t.append_message(b"dom-sep", b"r1cs-1phase")
        n = n1
        padded = 1
        while padded < n:
            padded *= 2
        pad = padded - n
        if bp_gens.cap < padded:
            raise InvalidGeneratorsLength()
        ident = bytes(32)
        t_point(t, b"A_I2", ident)
        t_point(t, b"A_O2", ident)
        t_point(t, b"S2", ident)
        y = t_challenge(t, b"y")
        z = t_challenge(t, b"z")
        wL, wR, wO, wV, _ = self.flatten(z, m, False)
        yinv = rs.sc_inv(y)
        Y, Yi = _exp_iter(y, padded + 1), _exp_iter(yinv, padded)
        l1 = [(self.aL[i] + Yi[i] * wR[i]) % L for i in range(n)]
        l2 = list(self.aO)
        l3 = sL
        r0 = [(wO[i] - Y[i]) % L for i in range(n)]
        r1 = [(Y[i] * self.aR[i] + wL[i]) % L for i in range(n)]
        r3 = [Y[i] * sR[i] % L for i in range(n)]
        t1 = _ip(l1, r0)
        t2 = (_ip(l1, r1) + _ip(l2, r0)) % L
        t3 = (_ip(l2, r1) + _ip(l3, r0)) % L
        t4 = (_ip(l1, r3) + _ip(l3, r1)) % L
        t5 = _ip(l2, r3)
        t6 = _ip(l3, r3)
        tb1, tb3, tb4, tb5, tb6 = rnd(), rnd(), rnd(), rnd(), rnd()
        T = {}
        for k, tv, tbv in [(1, t1, tb1), (3, t3, tb3), (4, t4, tb4), (5, t5, tb5), (6, t6, tb6)]:
            T[k] = rs.encode(pc.commit(tv, tbv))
            t_point(t, b"T_%d" % k, T[k])
        u = t_challenge(t, b"u")
        x = t_challenge(t, b"x")
        tb2 = _ip(wV, self.vb)
        xs = _exp_iter(x, 7)
        t_x = sum(c * xs[k] for k, c in zip(range(1, 7), [t1, t2, t3, t4, t5, t6])) % L
        t_xb = sum(c * xs[k] for k, c in zip(range(1, 7), [tb1, tb2, tb3, tb4, tb5, tb6])) % L
        l_vec = [(l1[i] * xs[1] + l2[i] * xs[2] + l3[i] * xs[3]) % L for i in range(n)] + [0] * pad
        r_vec = [(r0[i] + r1[i] * xs[1] + r3[i] * xs[3]) % L for i in range(n)] + [(-Y[i]) % L for i in range(n, padded)]
        e_bl = x * (ib + x * (ob + x * sb)) % L
        t_scalar(t, b"t_x", t_x)
        t_scalar(t, b"t_x_blinding", t_xb)
        t_scalar(t, b"e_blinding", e_bl)
        w = t_challenge(t, b"w")
        Q = rs.pt_mul(w, pc.B)
        Gf = [1] * n1 + [u] * pad
        Hf = [Yi[i] * Gf[i] % L for i in range(padded)]
        if trace is not None:
            trace.update(y=rs.sc_bytes(y).hex(), z=rs.sc_bytes(z).hex(), u=rs.sc_bytes(u).hex(),
                         x=rs.sc_bytes(x).hex(), w=rs.sc_bytes(w).hex(), n_mul=n1, n_constraints=len(self.constraints),
                         t_coeffs=[rs.sc_bytes(c).hex() for c in (t1, t2, t3, t4, t5, t6)])
        Ls, Rs, a, b = ipp_create(t, Q, Gf, Hf, G[:padded], H[:padded], l_vec, r_vec, trace)
        return R1CSProof(A_I1=A_I1, A_O1=A_O1, S1=S1, A_I2=ident, A_O2=ident, S2=ident,
                         T_1=T[1], T_3=T[3], T_4=T[4], T_5=T[5], T_6=T[6],
                         t_x=t_x, t_x_blinding=t_xb, e_blinding=e_bl, L_vec=Ls, R_vec=Rs, a=a, b=b)


# ---- verifier (A.7) --------------------------------------------------------------------------------
class Verifier(_CSBase):
    def __init__(self, transcript):
        super().__init__()
        self.t = transcript
        self.t.append_message(b"dom-sep", b"r1cs v1")
        self.V = []

    def commit(self, Vbytes):
        i = len(self.V)
        self.V.append(Vbytes)
        t_point(self.t, b"V", Vbytes)
        return (COMMITTED, i)

    def multiply(self, left, right):
        left, right = LC.of(left), LC.of(right)
        i = self.n_mul
        self.n_mul += 1
        self.constrain(left - (MUL_L, i))
        self.constrain(right - (MUL_R, i))
        return (MUL_L, i), (MUL_R, i), (MUL_O, i)

    def verify(self, proof, pc, bp_gens, entropy32=bytes(32)):
        t = self.t
        m = len(self.V)
        t.append_u64(b"m", m)
        n1 = self.n_mul
        t_validate_point(t, b"A_I1", proof.A_I1)
        t_validate_point(t, b"A_O1", proof.A_O1)
        t_validate_point(t, b"S1", proof.S1)
        t.append_message(b"dom-sep", b"r1cs-1phase")
        n = n1
        padded = 1
        while padded < n:
            padded *= 2
        pad = padded - n
        if bp_gens.cap < padded:
            raise InvalidGeneratorsLength()
        t_point(t, b"A_I2", proof.A_I2)
        t_point(t, b"A_O2", proof.A_O2)
        t_point(t, b"S2", proof.S2)
        y = t_challenge(t, b"y")
        z = t_challenge(t, b"z")
        for k in (1, 3, 4, 5, 6):
            t_validate_point(t, b"T_%d" % k, getattr(proof, "T_%d" % k))
        u = t_challenge(t, b"u")
        x = t_challenge(t, b"x")
        t_scalar(t, b"t_x", proof.t_x)
        t_scalar(t, b"t_x_blinding", proof.t_x_blinding)
        t_scalar(t, b"e_blinding", proof.e_blinding)
        w = t_challenge(t, b"w")
        wL, wR, wO, wV, wc = self.flatten(z, m, True)
        u_sq, u_inv_sq, s, _ = ipp_verification_scalars(proof.L_vec, proof.R_vec, padded, t)
        a, b = proof.a, proof.b
        Yi = _exp_iter(rs.sc_inv(y), padded)
        yneg_wR = [wR[i] * Yi[i] % L for i in range(n)] + [0] * pad
        delta = _ip(yneg_wR[:n], wL)
        uf = [1] * n1 + [u] * pad
        wLp = wL + [0] * pad
        wOp = wO + [0] * pad
        g_sc = [uf[i] * (x * yneg_wR[i] - a * s[i]) % L for i in range(padded)]
        h_sc = [uf[i] * (Yi[i] * (x * wLp[i] + wOp[i] - b * s[padded - 1 - i]) - 1) % L for i in range(padded)]
        rng = t.build_rng([], entropy32)
        r = rs.sc_wide(rng.fill_bytes(64))
        xx = x * x % L
        rxx = r * xx % L
        xxx = x * xx % L
        scal = [x, xx, xxx, u * x, u * xx, u * xxx] + [c * rxx for c in wV]
        scal += [r * x, rxx * x, rxx * xx, rxx * xxx, rxx * xx * xx]
        scal += [w * (proof.t_x - a * b) + r * (xx * (wc + delta) - proof.t_x), -proof.e_blinding - r * proof.t_x_blinding]
        scal += g_sc + h_sc + u_sq + u_inv_sq
        pbytes = [proof.A_I1, proof.A_O1, proof.S1, proof.A_I2, proof.A_O2, proof.S2] + self.V
        pbytes += [proof.T_1, proof.T_3, proof.T_4, proof.T_5, proof.T_6]
        pts = []
        for pb in pbytes:
            p = rs.decode(pb)
            if p is None:
                raise VerificationError("decompress")
            pts.append(p)
        pts += [pc.B, pc.B_blinding] + bp_gens.G[:padded] + bp_gens.H[:padded]
        for pb in proof.L_vec + proof.R_vec:
            p = rs.decode(pb)
            if p is None:
                raise VerificationError("decompress")
            pts.append(p)
        mega = rs.msm(scal, pts)
        if not rs.pt_eq(mega, rs.IDENT):
            raise VerificationError("mega check")
        return True
