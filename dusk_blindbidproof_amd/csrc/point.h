// Edwards25519 points (a = -1) in extended coordinates and the ristretto255 codec, for gfx950.
//
// Replaces the role of curve25519-dalek 1.2.3 `RistrettoPoint` / `CompressedRistretto`
// (un-vendored; call sites src/blindbid/proof.rs:57-64,129,165,
// src/blindbid/verify.rs:70).  Codec and one-way map are RFC 9496 section 4.3 step for step.
// Group elements are canonical under `ge_encode`, so evaluation order never changes bytes.
#pragma once
#include "field.h"

namespace bbp {

struct ge {  // extended: x = X/Z, y = Y/Z, T = XY/Z
    fe X, Y, Z, T;
};

struct ge_niels {  // affine (Z = 1) cached form for mixed addition, in registers
    fe ypx, ymx, xy2d;
};

struct niels_packed {  // the same three field elements as canonical 8-word values: 96 bytes per table entry in HBM
    u32 w[24];
};

// MSM table row, one cache line of canonical limbs: y+x at words 0..9, y-x at words 16..25 (byte offsets 0 and 64: a negative digit
// swaps them by swapping two load offsets), 2dxy in the gaps (words 10..15 and 26..29), 2 pad words
struct alignas(128) niels_row {
    i32 v[32];
};

BBP_HD niels_packed niels_pack(const ge_niels& n) {
    niels_packed p;
    fe_towords(p.w, n.ypx);
    fe_towords(p.w + 8, n.ymx);
    fe_towords(p.w + 16, n.xy2d);
    return p;
}

BBP_HD niels_row niels_to_row(const ge_niels& n) {
    niels_row r;
    u32 w[8];
    const fe* src[3] = {&n.ypx, &n.ymx, &n.xy2d};
    fe c[3];
    for (int k = 0; k < 3; k++) {  // through the canonical words: limbs come out non-negative and fully carried
        fe_towords(w, *src[k]);
        c[k] = fe_fromwords(w);
    }
    for (int i = 0; i < 10; i++) {
        r.v[i] = c[0].v[i];
        r.v[16 + i] = c[1].v[i];
        r.v[i < 6 ? 10 + i : 20 + i] = c[2].v[i];  // 10..15, then 26..29
    }
    r.v[30] = r.v[31] = 0;
    return r;
}

BBP_HD ge ge_identity() {
    ge r;
    r.X = fe_zero();
    r.Y = fe_one();
    r.Z = fe_one();
    r.T = fe_zero();
    return r;
}

BBP_HD ge ge_basepoint() {
    ge r;
    r.X = BBP_FE_LIT(0x8f25d51au, 0xc9562d60u, 0x9525a7b2u, 0x692cc760u, 0xfdd6dc5cu, 0xc0a4e231u, 0xcd6e53feu, 0x216936d3u);
    r.Y = BBP_FE_LIT(0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u);
    r.Z = fe_one();
    r.T = BBP_FE_LIT(0xa5b7dda3u, 0x6dde8ab3u, 0x775152f5u, 0x20f09f80u, 0x64abe37du, 0x66ea4e8eu, 0xd78b7665u, 0x67875f0fu);
    return r;
}

BBP_HD ge ge_neg(const ge& p) {
    ge r;
    r.X = fe_neg(p.X);
    r.Y = p.Y;
    r.Z = p.Z;
    r.T = fe_neg(p.T);
    return r;
}

// unified addition, complete on the curve (add-2008-hwcd-3): 9M
BBP_HD ge ge_add(const ge& p, const ge& q) {
    fe a = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    fe b = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    fe c = fe_mul(fe_mul(p.T, q.T), fe_d2());
    fe zz = fe_mul(p.Z, q.Z);
    fe d = fe_add(zz, zz);
    fe e = fe_sub(b, a), f = fe_sub(d, c), g = fe_add(d, c), h = fe_add(b, a);
    ge r;
    r.X = fe_mul(e, f);
    r.Y = fe_mul(g, h);
    r.Z = fe_mul(g, f);  // (g first: the four products then share 2 x {e, g} and 19 x {f, h})
    r.T = fe_mul(e, h);
    return r;
}

BBP_HD ge ge_sub(const ge& p, const ge& q) { return ge_add(p, ge_neg(q)); }

// mixed addition with a cached affine point: 7M
BBP_HD ge ge_madd(const ge& p, const ge_niels& q) {
    fe a = fe_mul(fe_sub(p.Y, p.X), q.ymx);
    fe b = fe_mul(fe_add(p.Y, p.X), q.ypx);
    fe c = fe_mul(p.T, q.xy2d);
    fe d = fe_add(p.Z, p.Z);
    fe e = fe_sub(b, a), f = fe_sub(d, c), g = fe_add(d, c), h = fe_add(b, a);
    ge r;
    r.X = fe_mul(e, f);
    r.Y = fe_mul(g, h);
    r.Z = fe_mul(g, f);  // (g first: the four products then share 2 x {e, g} and 19 x {f, h})
    r.T = fe_mul(e, h);
    return r;
}

// p - q for a cached affine q
BBP_HD ge ge_msub(const ge& p, const ge_niels& q) {
    fe a = fe_mul(fe_sub(p.Y, p.X), q.ypx);
    fe b = fe_mul(fe_add(p.Y, p.X), q.ymx);
    fe c = fe_mul(p.T, q.xy2d);
    fe d = fe_add(p.Z, p.Z);
    fe e = fe_sub(b, a), f = fe_add(d, c), g = fe_sub(d, c), h = fe_add(b, a);
    ge r;
    r.X = fe_mul(e, f);
    r.Y = fe_mul(g, h);
    r.Z = fe_mul(g, f);  // (g first: the four products then share 2 x {e, g} and 19 x {f, h})
    r.T = fe_mul(e, h);
    return r;
}

// dbl-2008-hwcd: 4S + 4M
BBP_HD ge ge_dbl(const ge& p) {
    fe a = fe_sq(p.X);
    fe b = fe_sq(p.Y);
    fe c = fe_sq2(p.Z);  // carried, so that f = c + g stays inside the multiply bounds
    fe h = fe_add(a, b);
    fe xy = fe_add(p.X, p.Y);
    fe e = fe_sub(h, fe_sq(xy));
    fe g = fe_sub(a, b);
    fe f = fe_add(c, g);
    ge r;
    r.X = fe_mul(e, f);
    r.Y = fe_mul(g, h);
    r.Z = fe_mul(g, f);  // (g first: the four products then share 2 x {e, g} and 19 x {f, h})
    r.T = fe_mul(e, h);
    return r;
}

#if defined(__HIPCC__)
// ---- 128-byte limb rows in registers (device only): the MSM row table (msm.hip) and the per-proof tail tables (prover.hip) --------
// one row as the three field elements of a cached point, y+x and y-x already swapped for a negative digit (the swap is two load
// offsets, not twenty selects)
struct row_regs {
    fe ypx, ymx, xy2d;
};

__device__ __forceinline__ fe load_fe40(const u8* p) {  // 10 limbs at a 16-byte aligned address
    const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 16);
    const uint2 c = *reinterpret_cast<const uint2*>(p + 32);
    return fe{{(i32)a.x, (i32)a.y, (i32)a.z, (i32)a.w, (i32)b.x, (i32)b.y, (i32)b.z, (i32)b.w, (i32)c.x, (i32)c.y}};
}

__device__ __forceinline__ row_regs load_row_at(const niels_row* __restrict__ row, u32 neg) {  // neg: 0 or 1
    const u8* p = reinterpret_cast<const u8*>(row);
    const u32 swap = neg << 6;  // 64 for a negative digit
    row_regs r;
    r.ypx = load_fe40(p + swap);
    r.ymx = load_fe40(p + (swap ^ 64u));
    const uint2 x0 = *reinterpret_cast<const uint2*>(p + 40);
    const uint4 x1 = *reinterpret_cast<const uint4*>(p + 48);
    const uint2 x2 = *reinterpret_cast<const uint2*>(p + 104), x3 = *reinterpret_cast<const uint2*>(p + 112);
    r.xy2d = fe{{(i32)x0.x, (i32)x0.y, (i32)x1.x, (i32)x1.y, (i32)x1.z, (i32)x1.w, (i32)x2.x, (i32)x2.y, (i32)x3.x, (i32)x3.y}};
    return r;
}

// acc +/- row: mixed addition (7M); for a negative digit y+x / y-x arrive swapped and D - C / D + C swap roles
__device__ __forceinline__ ge ge_madd_row(const ge& p, const row_regs& q, bool neg) {
    fe a = fe_mul(fe_sub(p.Y, p.X), q.ymx);
    fe b = fe_mul(fe_add(p.Y, p.X), q.ypx);
    fe c = fe_mul(p.T, q.xy2d);
    fe d = fe_add(p.Z, p.Z);
    fe e = fe_sub(b, a), h = fe_add(b, a);
    fe f0 = fe_sub(d, c), g0 = fe_add(d, c);
    fe f = fe_select(f0, g0, neg), g = fe_select(g0, f0, neg);
    ge r;
    r.X = fe_mul(e, f);
    r.Y = fe_mul(g, h);
    r.Z = fe_mul(g, f);  // (g first: the four products then share 2 x {e, g} and 19 x {f, h})
    r.T = fe_mul(e, h);
    return r;
}
#endif

// affine cached form given 1/Z
BBP_HD ge_niels ge_to_niels(const ge& p, const fe& zinv) {
    fe x = fe_mul(p.X, zinv), y = fe_mul(p.Y, zinv);
    ge_niels r;
    r.ypx = fe_add(y, x);
    r.ymx = fe_sub(y, x);
    r.xy2d = fe_mul(fe_mul(x, y), fe_d2());
    return r;
}

BBP_HD ge ge_from_niels(const ge_niels& n) {  // recover (x, y, 1, xy): halve via (p+1)/2
    // x = (ypx - ymx)/2, y = (ypx + ymx)/2 ; 1/2 = 2^254 - 9 ... use multiply by inverse of 2
    const fe half = BBP_FE_LIT(0xfffffff7u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu);
    ge r;
    r.X = fe_mul(fe_sub(n.ypx, n.ymx), half);
    r.Y = fe_mul(fe_add(n.ypx, n.ymx), half);
    r.Z = fe_one();
    r.T = fe_mul(r.X, r.Y);
    return r;
}

// Ristretto equality (RFC 9496 4.3.3)
BBP_HD bool ge_eq(const ge& p, const ge& q) {
    return fe_eq(fe_mul(p.X, q.Y), fe_mul(p.Y, q.X)) || fe_eq(fe_mul(p.Y, q.Y), fe_mul(p.X, q.X));
}

BBP_HD bool ge_is_identity(const ge& p) { return fe_iszero(p.X) || fe_iszero(p.Y); }

// RFC 9496 4.3.2 Encode
BBP_HD_NOINLINE fe ge_encode_s(const ge& p) {
    fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
    fe u2 = fe_mul(p.X, p.Y);
    fe invsqrt;
    fe_sqrt_ratio_m1(invsqrt, fe_one(), fe_mul(u1, fe_sq(u2)));
    fe den1 = fe_mul(invsqrt, u1);
    fe den2 = fe_mul(invsqrt, u2);
    fe z_inv = fe_mul(fe_mul(den1, den2), p.T);
    fe ix0 = fe_mul(p.X, fe_sqrt_m1());
    fe iy0 = fe_mul(p.Y, fe_sqrt_m1());
    fe ench = fe_mul(den1, fe_invsqrt_a_minus_d());
    bool rotate = fe_isneg(fe_mul(p.T, z_inv));
    fe x = fe_select(p.X, iy0, rotate);
    fe y = fe_select(p.Y, ix0, rotate);
    fe den_inv = fe_select(den2, ench, rotate);
    y = fe_select(y, fe_neg(y), fe_isneg(fe_mul(x, z_inv)));
    return fe_abs(fe_mul(den_inv, fe_sub(p.Z, y)));
}

BBP_HD void ge_encode_words(u32* out8, const ge& p) { fe_towords(out8, ge_encode_s(p)); }

BBP_HD void ge_encode(uint8_t* out32, const ge& p) { fe_tobytes(out32, ge_encode_s(p)); }

// RFC 9496 4.3.1 Decode from 8 LE words; false on any failure (non-canonical, negative, not on curve)
BBP_HD_NOINLINE bool ge_decode_words(ge& out, const u32* w) {
    fe s = fe_fromwords(w);
    // canonical (< p, bit 255 clear) and non-negative (even): re-encoding must give the same 8 words
    u32 chk[8];
    fe_towords(chk, s);
    u32 diff = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) diff |= chk[i] ^ w[i];
    bool ok = (diff == 0) && !(w[0] & 1u);
    fe ss = fe_sq(s);
    fe u1 = fe_sub(fe_one(), ss);
    fe u2 = fe_add(fe_one(), ss);
    fe u2_sqr = fe_sq(u2);
    fe v = fe_sub(fe_neg(fe_mul(fe_d(), fe_sq(u1))), u2_sqr);
    fe invsqrt;
    bool was_square = fe_sqrt_ratio_m1(invsqrt, fe_one(), fe_mul(v, u2_sqr));
    fe den_x = fe_mul(invsqrt, u2);
    fe den_y = fe_mul(fe_mul(invsqrt, den_x), v);
    fe x = fe_abs(fe_mul(fe_add(s, s), den_x));
    fe y = fe_mul(u1, den_y);
    fe t = fe_mul(x, y);
    ok = ok && was_square && !fe_isneg(t) && !fe_iszero(y);
    out.X = x;
    out.Y = y;
    out.Z = fe_one();
    out.T = t;
    return ok;
}

// RFC 9496 4.3.4 MAP (Elligator 2, ristretto flavour)
BBP_HD_NOINLINE ge ge_elligator(const fe& t) {
    fe r = fe_mul(fe_sqrt_m1(), fe_sq(t));
    fe u = fe_mul(fe_add(r, fe_one()), fe_one_minus_d_sq());
    fe v = fe_mul(fe_sub(fe_neg(fe_one()), fe_mul(r, fe_d())), fe_add(r, fe_d()));
    fe s;
    bool was_square = fe_sqrt_ratio_m1(s, u, v);
    fe s_prime = fe_neg(fe_abs(fe_mul(s, t)));
    s = fe_select(s_prime, s, was_square);
    fe c = fe_select(r, fe_neg(fe_one()), was_square);
    fe n = fe_sub(fe_mul(fe_mul(c, fe_sub(r, fe_one())), fe_d_minus_one_sq()), v);
    fe w0 = fe_mul(fe_add(s, s), v);
    fe w1 = fe_mul(n, fe_sqrt_ad_minus_one());
    fe ss = fe_sq(s);
    fe w2 = fe_sub(fe_one(), ss);
    fe w3 = fe_add(fe_one(), ss);
    ge p;
    p.X = fe_mul(w0, w3);
    p.Y = fe_mul(w2, w1);
    p.Z = fe_mul(w1, w3);
    p.T = fe_mul(w0, w2);
    return p;
}

// dalek RistrettoPoint::from_uniform_bytes: 16 LE words -> point
BBP_HD ge ge_from_uniform_words(const u32* w16) {
    fe r0 = fe_fromwords(w16);
    fe r1 = fe_fromwords(w16 + 8);
    return ge_add(ge_elligator(r0), ge_elligator(r1));
}

}  // namespace bbp
