"""ORACLE (test infrastructure, never shipped): big-int restatement of curve25519-dalek 1.2.3
as used by dusk-blindbidproof's hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows (published algorithms; crate source is NOT under /root/reference):
  * curve25519-dalek 1.2.3 (Cargo.lock:141-143) `Scalar`, `FieldElement`, `RistrettoPoint`,
    `CompressedRistretto` -- call sites src/blindbid/mod.rs:16, src/blindbid/bid.rs:27,
    src/blindbid/verify.rs:115, src/blindbid/proof.rs:57-64.
  * ristretto255 encode/decode/one-way-map exactly as RFC 9496 section 4.3 (derived from dalek).
Pinned by tests/test_oracle_kat.py against RFC 9496 appendix A vectors.
"""
import hashlib

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)
SQRT_AD_MINUS_ONE = 25063068953384623474111414158702152701244531502492656460079210482610430750235
INVSQRT_A_MINUS_D = 54469307008909316920995813868745141605393597292927456921205312896311721017578
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = ((D - 1) * (D - 1)) % P


def _is_neg(x):
    return x & 1


def _abs(x):
    return (P - x) % P if x & 1 else x


def sqrt_ratio_m1(u, v):
    """RFC 9496 4.2 SQRT_RATIO_M1."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = (u * v3) % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (P - u) % P
    flipped_i = check == (P - u) % P * SQRT_M1 % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    return (correct or flipped), _abs(r)


# ---- Edwards points, extended coordinates (X, Y, Z, T), a = -1 --------------------------------
IDENT = (0, 1, 1, 0)


def pt_add(p, q):
    x1, y1, z1, t1 = p
    x2, y2, z2, t2 = q
    a = (y1 - x1) * (y2 - x2) % P
    b = (y1 + x1) * (y2 + x2) % P
    c = 2 * D * t1 % P * t2 % P
    d = 2 * z1 * z2 % P
    e, f, g, h = b - a, d - c, d + c, b + a
    return (e * f % P, g * h % P, f * g % P, e * h % P)


def pt_dbl(p):
    x1, y1, z1, _ = p
    a = x1 * x1 % P
    b = y1 * y1 % P
    c = 2 * z1 * z1 % P
    h = a + b
    e = h - (x1 + y1) * (x1 + y1) % P
    g = a - b
    f = c + g
    return (e * f % P, g * h % P, f * g % P, e * h % P)


def pt_neg(p):
    return ((P - p[0]) % P, p[1], p[2], (P - p[3]) % P)


def pt_mul(k, p):
    k %= L
    r = IDENT
    for bit in bin(k)[2:] if k else "":
        r = pt_dbl(r)
        if bit == "1":
            r = pt_add(r, p)
    return r


def pt_eq(p, q):
    """Ristretto equality (RFC 9496 4.3.3)."""
    x1, y1, _, _ = p
    x2, y2, _, _ = q
    return (x1 * y2 - y1 * x2) % P == 0 or (y1 * y2 - x1 * x2) % P == 0


def msm(scalars, points):
    """Pippenger over big-ints; any evaluation order yields the same group element."""
    scalars = [s % L for s in scalars]
    n = len(scalars)
    if n == 0:
        return IDENT
    if n < 8:
        r = IDENT
        for s, p in zip(scalars, points):
            r = pt_add(r, pt_mul(s, p))
        return r
    c = 4 if n < 64 else 6 if n < 512 else 8
    nwin = (253 + c - 1) // c
    acc = IDENT
    for w in reversed(range(nwin)):
        for _ in range(c):
            acc = pt_dbl(acc)
        buckets = [None] * (1 << c)
        sh = w * c
        for s, p in zip(scalars, points):
            dgt = (s >> sh) & ((1 << c) - 1)
            if dgt:
                buckets[dgt] = p if buckets[dgt] is None else pt_add(buckets[dgt], p)
        run, tot = IDENT, IDENT
        for b in reversed(buckets[1:]):
            if b is not None:
                run = pt_add(run, b)
            tot = pt_add(tot, run)
        acc = pt_add(acc, tot)
    return acc


# ---- ristretto255 -------------------------------------------------------------------------------
def decode(b):
    """RFC 9496 4.3.1; returns None on failure (dalek CompressedRistretto::decompress)."""
    if len(b) != 32:
        return None
    s = int.from_bytes(b, "little")
    if s >= P or _is_neg(s):
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2_sqr = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2_sqr) % P
    was_square, invsqrt = sqrt_ratio_m1(1, v * u2_sqr % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = _abs(2 * s * den_x % P)
    y = u1 * den_y % P
    t = x * y % P
    if (not was_square) or _is_neg(t) or y == 0:
        return None
    return (x, y, 1, t)


def encode(p):
    """RFC 9496 4.3.2 (dalek RistrettoPoint::compress)."""
    x0, y0, z0, t0 = p
    u1 = (z0 + y0) * (z0 - y0) % P
    u2 = x0 * y0 % P
    _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
    den1 = invsqrt * u1 % P
    den2 = invsqrt * u2 % P
    z_inv = den1 * den2 % P * t0 % P
    ix0 = x0 * SQRT_M1 % P
    iy0 = y0 * SQRT_M1 % P
    ench = den1 * INVSQRT_A_MINUS_D % P
    if _is_neg(t0 * z_inv % P):
        x, y, den_inv = iy0, ix0, ench
    else:
        x, y, den_inv = x0, y0, den2
    if _is_neg(x * z_inv % P):
        y = (P - y) % P
    s = _abs(den_inv * ((z0 - y) % P) % P)
    return s.to_bytes(32, "little")


def elligator(t):
    """RFC 9496 4.3.4 MAP."""
    r = SQRT_M1 * t % P * t % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    v = (-1 - r * D) % P * ((r + D) % P) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = (P - _abs(s * t % P)) % P
    if not was_square:
        s = s_prime
    c = (P - 1) if was_square else r
    n = (c * ((r - 1) % P) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = n * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return (w0 * w3 % P, w2 * w1 % P, w1 * w3 % P, w0 * w2 % P)


def from_uniform_bytes(b64):
    """dalek RistrettoPoint::from_uniform_bytes: two field elements (bit 255 masked), MAP each, add."""
    assert len(b64) == 64
    r0 = (int.from_bytes(b64[:32], "little") & ((1 << 255) - 1)) % P
    r1 = (int.from_bytes(b64[32:], "little") & ((1 << 255) - 1)) % P
    return pt_add(elligator(r0), elligator(r1))


_BY = 4 * pow(5, P - 2, P) % P
_BX = sqrt_ratio_m1((_BY * _BY - 1) % P, (D * _BY * _BY + 1) % P)[1]
if _BX & 1:  # ed25519 basepoint has even ("positive") x
    _BX = P - _BX
BASEPOINT = (_BX, _BY, 1, _BX * _BY % P)


# ---- scalars --------------------------------------------------------------------------------------
def sc_wide(b64):
    """Scalar::from_bytes_mod_order_wide."""
    return int.from_bytes(b64, "little") % L


def sc_from_bits(b32):
    """Scalar::from_bits (src/blindbid/bid.rs:27, verify.rs:115): clear bit 255, NOT reduced;
    downstream arithmetic is mod l, so the oracle reduces here."""
    return (int.from_bytes(b32, "little") & ((1 << 255) - 1)) % L


def sc_bytes(s):
    return (s % L).to_bytes(32, "little")


def sc_canonical(b32):
    """Scalar::from_canonical_bytes: None if not < l."""
    v = int.from_bytes(b32, "little")
    return v if v < L else None


def sc_inv(s):
    return pow(s % L, L - 2, L)


def sha3_512(b):
    return hashlib.sha3_512(b).digest()
