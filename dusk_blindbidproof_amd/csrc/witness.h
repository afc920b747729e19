// The prover's gate assignments a_L, a_R, a_O of ONE proof (bulletproofs Prover::multiply under the reference's gadgets,
// src/gadgets.rs:6-140), two ways, host + device:
//   witness_gates_interpret  walks the program circuit.h compiles from the generic gadget synthesis (any wiring);
//   witness_gates_native     is the gadget wiring of the reference written out, values kept in registers along a MiMC chain.
// Both must produce the same canonical scalars (tests/test_host_arith.py compares them with each other and with the big-int
// oracle's Prover on the host; on the device every record test compares the proof bytes under each).  Callers: prover.hip.
#pragma once
#include "circuit.h"
#include "scalar.h"

namespace bbp {

BBP_HD sc wt_ld(const sc* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    return BBP_SC_LIT(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
#else
    return *p;
#endif
}
BBP_HD void wt_st(sc* p, const sc& s) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(s.v[0], s.v[1], s.v[2], s.v[3]);
    q[1] = make_uint4(s.v[4], s.v[5], s.v[6], s.v[7]);
#else
    *p = s;
#endif
}

// cst: the proof's constant table (circuit.h CST_*), v: its committed values (d, k, y, y_inv, toggle bits);
// term word: [31:29] VarKind, [28] negative, [27:0] index (circuit.h pack_term)
BBP_HD void witness_gates_interpret(u32 n_mul, const u32* w_terms, const u32* w_loff, const u32* w_roff, const sc* cst, const sc* v,
                                    sc* aL, sc* aR, sc* aO) {
    for (u32 i = 0; i < n_mul; i++) {
        sc lr[2];
        const u32 bounds[3] = {w_loff[i], w_roff[i], w_loff[i + 1]};
        for (int side = 0; side < 2; side++) {
            sc acc = sc_zero();
            for (u32 t = bounds[side]; t < bounds[side + 1]; t++) {
                const u32 w = w_terms[t], kind = w >> 29, idx = w & 0x0fffffffu;
                const sc* src = kind == 4 ? &cst[idx] : kind == 3 ? &aO[idx] : kind == 1 ? &aL[idx] : kind == 2 ? &aR[idx] : &v[idx];
                sc val = wt_ld(src);
                acc = ((w >> 28) & 1u) ? sc_sub(acc, val) : sc_add(acc, val);
            }
            lr[side] = acc;
        }
        wt_st(&aL[i], lr[0]);
        wt_st(&aR[i], lr[1]);
        wt_st(&aO[i], sc_mul(lr[0], lr[1]));
    }
}

// src/gadgets.rs:37-68: 90 rounds of a = x + key + c_i; a^2, a^3 = a^2 a, a^4 = a^2 a^2, a^7 = a^4 a^3 (four multipliers, in this order)
BBP_HD sc witness_mimc_native(sc x, const sc& key, const sc* cst, sc* aL, sc* aR, sc* aO, u32 base) {
    for (u32 i = 0; i < (u32)circuit::MIMC_ROUNDS; i++) {
        const sc a = sc_add(sc_add(x, key), wt_ld(&cst[circuit::CST_MIMC0 + i]));
        const sc a2 = sc_mul(a, a), a3 = sc_mul(a2, a), a4 = sc_mul(a2, a2), a7 = sc_mul(a4, a3);
        const u32 j = base + 4 * i;
        wt_st(&aL[j], a);      wt_st(&aR[j], a);      wt_st(&aO[j], a2);
        wt_st(&aL[j + 1], a2); wt_st(&aR[j + 1], a);  wt_st(&aO[j + 1], a3);
        wt_st(&aL[j + 2], a2); wt_st(&aR[j + 2], a2); wt_st(&aO[j + 2], a4);
        wt_st(&aL[j + 3], a4); wt_st(&aR[j + 3], a3); wt_st(&aO[j + 3], a7);
        x = a7;
    }
    return sc_add(x, key);
}

// src/gadgets.rs:6-34 proof_gadget in call order; returns the number of multipliers written (4 * 4 * 90 + 3 N + 2)
BBP_HD u32 witness_gates_native(u32 n_items, const sc* cst, const sc* v, sc* aL, sc* aR, sc* aO) {
    const u32 N = n_items, R4 = 4 * (u32)circuit::MIMC_ROUNDS;
    const sc d = wt_ld(&v[0]), k = wt_ld(&v[1]), y_inv = wt_ld(&v[3]), seed = wt_ld(&cst[circuit::CST_SEED]);
    const sc mm = witness_mimc_native(k, sc_zero(), cst, aL, aR, aO, 0);      // m = H(k)
    const sc x = witness_mimc_native(d, mm, cst, aL, aR, aO, R4);             // x = H(d, m)
    u32 j = 2 * R4;
    for (u32 i = 0; i < N; i++, j++) {                                       // :134-140 toggle bits are bits: t (1 - t) = 0
        const sc t = wt_ld(&v[4 + i]), nt = sc_sub(sc_one(), t);
        wt_st(&aL[j], t);
        wt_st(&aR[j], nt);
        wt_st(&aO[j], sc_mul(t, nt));
    }
    for (u32 i = 0; i < N; i++, j += 2) {                                    // :88-132 item_i t_i = t_i x
        const sc t = wt_ld(&v[4 + i]), item = wt_ld(&cst[circuit::CST_ITEM0 + i]);
        wt_st(&aL[j], item);
        wt_st(&aR[j], t);
        wt_st(&aO[j], sc_mul(item, t));
        wt_st(&aL[j + 1], t);
        wt_st(&aR[j + 1], x);
        wt_st(&aO[j + 1], sc_mul(t, x));
    }
    const sc y = witness_mimc_native(seed, x, cst, aL, aR, aO, j);            // y = H(seed, x)
    witness_mimc_native(seed, mm, cst, aL, aR, aO, j + R4);                   // z = H(seed, m): constrained against z_img, not multiplied further
    j += 2 * R4;
    wt_st(&aL[j], y);                                                        // :70-86 score: y y_inv = 1, d y_inv = q
    wt_st(&aR[j], y_inv);
    wt_st(&aO[j], sc_mul(y, y_inv));
    wt_st(&aL[j + 1], d);
    wt_st(&aR[j + 1], y_inv);
    wt_st(&aO[j + 1], sc_mul(d, y_inv));
    return j + 2;
}

}  // namespace bbp
