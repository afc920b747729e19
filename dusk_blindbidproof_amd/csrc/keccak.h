// Keccak-f[1600], STROBE-128 and the Merlin transcript (host + gfx950 device).
//
// Replaces the role of merlin 1.3.0 (+ keccak 0.1.0), un-vendored;
// reference call site: src/blindbid/mod.rs:37 `Transcript::new(b"BlindBidProofGadget")`, everything else
// is driven from inside bulletproofs' prover / verifier.  On the device one lane owns one proof's
// transcript (200-byte sponge + 3 counters); the 25 lanes of the sponge are 64-bit register pairs
// during the permutation.
#pragma once
#include "field.h"

namespace bbp {

// 64-bit rotate left.  On the device a rotation by a constant is two v_alignbit_b32 (the compiler's own lowering of the shift /
// or form costs three instructions: 64-bit shift, 32-bit shift, or) -- 27 of the ~300 instructions of a Keccak round, and the
// prover's 2935-permutation rng chain runs at exactly the single-wave issue limit.
BBP_HD u64 rotl64(u64 x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    if (n == 32) return ((u64)lo << 32) | hi;
    const u32 a = n < 32 ? lo : hi, b = n < 32 ? hi : lo;  // rotate {b:a} left by n mod 32
    const u32 s = 32u - ((u32)n & 31u);
    const u32 rl = __builtin_amdgcn_alignbit(a, b, s), rh = __builtin_amdgcn_alignbit(b, a, s);
    return ((u64)rh << 32) | rl;
#else
    return (x << n) | (x >> (64 - n));
#endif
}

// forceinline body: with `s` a local array indexed statically the 25 lanes live in registers across calls
BBP_HD void keccak_f1600_body(u64* s) {
    const u64 RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                        0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                        0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                        0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    u64 a00 = s[0], a01 = s[1], a02 = s[2], a03 = s[3], a04 = s[4];
    u64 a05 = s[5], a06 = s[6], a07 = s[7], a08 = s[8], a09 = s[9];
    u64 a10 = s[10], a11 = s[11], a12 = s[12], a13 = s[13], a14 = s[14];
    u64 a15 = s[15], a16 = s[16], a17 = s[17], a18 = s[18], a19 = s[19];
    u64 a20 = s[20], a21 = s[21], a22 = s[22], a23 = s[23], a24 = s[24];
    for (int r = 0; r < 24; r++) {
        // theta
        u64 c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20;
        u64 c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21;
        u64 c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22;
        u64 c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23;
        u64 c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
        u64 d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1);
        u64 d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);
        a00 ^= d0; a05 ^= d0; a10 ^= d0; a15 ^= d0; a20 ^= d0;
        a01 ^= d1; a06 ^= d1; a11 ^= d1; a16 ^= d1; a21 ^= d1;
        a02 ^= d2; a07 ^= d2; a12 ^= d2; a17 ^= d2; a22 ^= d2;
        a03 ^= d3; a08 ^= d3; a13 ^= d3; a18 ^= d3; a23 ^= d3;
        a04 ^= d4; a09 ^= d4; a14 ^= d4; a19 ^= d4; a24 ^= d4;
        // rho + pi : B[y][2x+3y] = rot(A[x][y])
        u64 b00 = a00;
        u64 b10 = rotl64(a01, 1), b20 = rotl64(a02, 62), b05 = rotl64(a03, 28), b15 = rotl64(a04, 27);
        u64 b16 = rotl64(a05, 36), b01 = rotl64(a06, 44), b11 = rotl64(a07, 6), b21 = rotl64(a08, 55), b06 = rotl64(a09, 20);
        u64 b07 = rotl64(a10, 3), b17 = rotl64(a11, 10), b02 = rotl64(a12, 43), b12 = rotl64(a13, 25), b22 = rotl64(a14, 39);
        u64 b23 = rotl64(a15, 41), b08 = rotl64(a16, 45), b18 = rotl64(a17, 15), b03 = rotl64(a18, 21), b13 = rotl64(a19, 8);
        u64 b14 = rotl64(a20, 18), b24 = rotl64(a21, 2), b09 = rotl64(a22, 61), b19 = rotl64(a23, 56), b04 = rotl64(a24, 14);
        // chi
        a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
        a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
        a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
        a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
        a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
        a00 ^= RC[r];
    }
    s[0] = a00; s[1] = a01; s[2] = a02; s[3] = a03; s[4] = a04;
    s[5] = a05; s[6] = a06; s[7] = a07; s[8] = a08; s[9] = a09;
    s[10] = a10; s[11] = a11; s[12] = a12; s[13] = a13; s[14] = a14;
    s[15] = a15; s[16] = a16; s[17] = a17; s[18] = a18; s[19] = a19;
    s[20] = a20; s[21] = a21; s[22] = a22; s[23] = a23; s[24] = a24;
}

BBP_HD_NOINLINE void keccak_f1600(u64* s) { keccak_f1600_body(s); }

// ---- STROBE-128 (rate 166) -------------------------------------------------------------------------
struct merlin_transcript {
    u64 st[25];
    u32 pos, pos_begin, cur_flags;
    u32 wave;  // device, prover.hip only: all 64 lanes of the wavefront hold THIS transcript and run it in lockstep; the permutation is then
               // spread over them (keccak_wave.h keccak_f1600_wave: 2.4 us instead of 15).  Set by the kernel after loading, meaningless in memory.
};
#if defined(__HIP_DEVICE_COMPILE__) && defined(BBP_KECCAK_WAVE)
__device__ void keccak_f1600_wave(u64* s);  // keccak_wave.h
#endif

#define BBP_STROBE_R 166u
#define BBP_FLAG_I 1u
#define BBP_FLAG_A 2u
#define BBP_FLAG_C 4u
#define BBP_FLAG_M 16u
#define BBP_FLAG_K 32u

BBP_HD void strobe_xor_byte(merlin_transcript& t, u32 pos, u32 b) { t.st[pos >> 3] ^= (u64)b << (8 * (pos & 7)); }
BBP_HD void strobe_set_byte(merlin_transcript& t, u32 pos, u32 b) {
    u32 sh = 8 * (pos & 7);
    t.st[pos >> 3] = (t.st[pos >> 3] & ~((u64)0xff << sh)) | ((u64)b << sh);
}
BBP_HD u32 strobe_get_byte(const merlin_transcript& t, u32 pos) { return (u32)(t.st[pos >> 3] >> (8 * (pos & 7))) & 0xffu; }

BBP_HD void strobe_run_f(merlin_transcript& t) {
    strobe_xor_byte(t, t.pos, t.pos_begin);
    strobe_xor_byte(t, t.pos + 1, 0x04);
    strobe_xor_byte(t, BBP_STROBE_R + 1, 0x80);
#if defined(__HIP_DEVICE_COMPILE__) && defined(BBP_KECCAK_WAVE)
    if (t.wave) keccak_f1600_wave(t.st);
    else
#endif
        keccak_f1600(t.st);
    t.pos = 0;
    t.pos_begin = 0;
}

BBP_HD void strobe_absorb(merlin_transcript& t, const uint8_t* d, u32 n) {
    for (u32 i = 0; i < n; i++) {
        strobe_xor_byte(t, t.pos, d[i]);
        if (++t.pos == BBP_STROBE_R) strobe_run_f(t);
    }
}

BBP_HD void strobe_overwrite(merlin_transcript& t, const uint8_t* d, u32 n) {
    for (u32 i = 0; i < n; i++) {
        strobe_set_byte(t, t.pos, d[i]);
        if (++t.pos == BBP_STROBE_R) strobe_run_f(t);
    }
}

BBP_HD void strobe_squeeze(merlin_transcript& t, uint8_t* d, u32 n) {
    for (u32 i = 0; i < n; i++) {
        d[i] = (uint8_t)strobe_get_byte(t, t.pos);
        strobe_set_byte(t, t.pos, 0);
        if (++t.pos == BBP_STROBE_R) strobe_run_f(t);
    }
}

BBP_HD void strobe_begin_op(merlin_transcript& t, u32 flags, bool more) {
    if (more) return;  // caller guarantees flags == cur_flags
    uint8_t hdr[2] = {(uint8_t)t.pos_begin, (uint8_t)flags};
    t.pos_begin = t.pos + 1;
    t.cur_flags = flags;
    strobe_absorb(t, hdr, 2);
    if ((flags & (BBP_FLAG_C | BBP_FLAG_K)) && t.pos != 0) strobe_run_f(t);
}

BBP_HD void strobe_meta_ad(merlin_transcript& t, const uint8_t* d, u32 n, bool more) {
    strobe_begin_op(t, BBP_FLAG_M | BBP_FLAG_A, more);
    strobe_absorb(t, d, n);
}
BBP_HD void strobe_ad(merlin_transcript& t, const uint8_t* d, u32 n, bool more) {
    strobe_begin_op(t, BBP_FLAG_A, more);
    strobe_absorb(t, d, n);
}
BBP_HD void strobe_prf(merlin_transcript& t, uint8_t* d, u32 n, bool more) {
    strobe_begin_op(t, BBP_FLAG_I | BBP_FLAG_A | BBP_FLAG_C, more);
    strobe_squeeze(t, d, n);
}
BBP_HD void strobe_key(merlin_transcript& t, const uint8_t* d, u32 n, bool more) {
    strobe_begin_op(t, BBP_FLAG_A | BBP_FLAG_C, more);
    strobe_overwrite(t, d, n);
}

BBP_HD void le32(uint8_t* o, u32 x) {
    o[0] = (uint8_t)x;
    o[1] = (uint8_t)(x >> 8);
    o[2] = (uint8_t)(x >> 16);
    o[3] = (uint8_t)(x >> 24);
}

// ---- Merlin ----------------------------------------------------------------------------------------
BBP_HD void merlin_append(merlin_transcript& t, const uint8_t* label, u32 llen, const uint8_t* msg, u32 mlen) {
    uint8_t len4[4];
    le32(len4, mlen);
    strobe_meta_ad(t, label, llen, false);
    strobe_meta_ad(t, len4, 4, true);
    strobe_ad(t, msg, mlen, false);
}

BBP_HD void merlin_append_u64(merlin_transcript& t, const uint8_t* label, u32 llen, u64 x) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
    merlin_append(t, label, llen, b, 8);
}

BBP_HD void merlin_init(merlin_transcript& t, const uint8_t* label, u32 llen) {
    for (int i = 0; i < 25; i++) t.st[i] = 0;
    const uint8_t hdr[18] = {1, 168, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    for (u32 i = 0; i < 18; i++) strobe_xor_byte(t, i, hdr[i]);
    keccak_f1600(t.st);
    t.pos = t.pos_begin = t.cur_flags = t.wave = 0;
    const uint8_t proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
    strobe_meta_ad(t, proto, 11, false);
    const uint8_t ds[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'};
    merlin_append(t, ds, 7, label, llen);
}

BBP_HD void merlin_challenge(merlin_transcript& t, const uint8_t* label, u32 llen, uint8_t* out, u32 n) {
    uint8_t len4[4];
    le32(len4, n);
    strobe_meta_ad(t, label, llen, false);
    strobe_meta_ad(t, len4, 4, true);
    strobe_prf(t, out, n, false);
}

// TranscriptRngBuilder on a COPY of the transcript
BBP_HD void merlin_rng_rekey(merlin_transcript& t, const uint8_t* label, u32 llen, const uint8_t* w, u32 wlen) {
    uint8_t len4[4];
    le32(len4, wlen);
    strobe_meta_ad(t, label, llen, false);
    strobe_meta_ad(t, len4, 4, true);
    strobe_key(t, w, wlen, false);
}

BBP_HD void merlin_rng_finalize(merlin_transcript& t, const uint8_t* ent32) {
    const uint8_t l[3] = {'r', 'n', 'g'};
    strobe_meta_ad(t, l, 3, false);
    strobe_key(t, ent32, 32, false);
}

BBP_HD void merlin_rng_fill(merlin_transcript& t, uint8_t* out, u32 n) {
    uint8_t len4[4];
    le32(len4, n);
    strobe_meta_ad(t, len4, 4, false);
    strobe_prf(t, out, n, false);
}

// `count` consecutive TranscriptRng::fill_bytes(64) calls (one per Scalar::random) in steady state.
// After any 64-byte fill the sponge sits at pos = 64, pos_begin = 0, and the next fill is always the same byte script:
//   meta_ad(u32_le(64))  -> absorb [0x00, M|A = 0x12] at 64,65 and [0x40,0,0,0] at 66..69
//   prf begin            -> absorb [65, I|A|C = 0x07] at 70,71 ; run_f: st[72] ^= 71, st[73] ^= 0x04, st[167] ^= 0x80
//   squeeze 64           -> output bytes 0..63, zero them, pos = 64
// so each draw is three constant lane XORs, one Keccak-f[1600] and a copy of lanes 0..7 -- no byte loops, and the 25
// lanes stay in registers for the whole run.  Returns false (and does nothing) if the sponge is not in that state.
BBP_HD bool merlin_rng_fill64_bulk(merlin_transcript& t, u32 count, u32* out_words /* count * 16 */) {
    if (t.pos != 64 || t.pos_begin != 0) return false;
    u64 a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = t.st[i];
    for (u32 c = 0; c < count; c++) {
        a[8] ^= 0x0741000000401200ull;
        a[9] ^= 0x0000000000000447ull;
        a[20] ^= 0x8000000000000000ull;
        keccak_f1600_body(a);
        u32* o = out_words + (size_t)c * 16;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            o[2 * i] = (u32)a[i];
            o[2 * i + 1] = (u32)(a[i] >> 32);
            a[i] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < 25; i++) t.st[i] = a[i];
    t.cur_flags = BBP_FLAG_I | BBP_FLAG_A | BBP_FLAG_C;
    return true;
}

}  // namespace bbp
