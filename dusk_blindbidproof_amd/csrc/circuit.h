// Host-side mirror of the reference's constraint-system / gadget interface, used to COMPILE the blind-bid circuit
// (once per bid-list length N) into flat device tables.  Same names, call order and semantics as
//   src/gadgets.rs (proof_gadget, mimc_gadget, score_gadget, one_of_many_gadget, boolean_gadget)
//   bulletproofs::r1cs::{ConstraintSystem, LinearCombination, Variable} (un-vendored)
// The circuit's structure depends only on N; per-proof public inputs (seed, z_img, q/score, pub_list) and the MiMC
// round constants enter linear combinations as *symbolic* constants (an index into a per-proof constant table), so
// one compiled circuit serves every proof of the batch and both roles (prover and verifier synthesise identical
// constraint lists: src/blindbid/proof.rs:74 and src/blindbid/verify.rs:74 call the same gadget).
#pragma once
#include <stdint.h>

#include <stdexcept>
#include <utility>
#include <vector>

namespace bbp {
namespace circuit {

// ---- per-proof constant table layout (device array of scalars) -------------------------------------------------
enum : uint32_t {
    CST_ONE = 0,
    CST_ZERO = 1,
    CST_MIMC0 = 2,              // c_0 .. c_89 (src/blindbid/mod.rs:7-24)
    CST_SEED = CST_MIMC0 + 90,  // 92
    CST_ZIMG = 93,
    CST_Q = 94,                 // q for the prover, `score` for the verifier
    CST_ITEM0 = 95              // pub_list[0..N)
};
inline uint32_t cst_count(uint32_t n_items) { return CST_ITEM0 + n_items; }

struct Scalar {  // symbolic constant: sign * table[cst]
    int sign;
    uint32_t cst;
    static Scalar one() { return {1, CST_ONE}; }
    static Scalar zero() { return {1, CST_ZERO}; }
    Scalar operator-() const { return {-sign, cst}; }
};

enum class VarKind : uint8_t { Committed = 0, MultiplierLeft = 1, MultiplierRight = 2, MultiplierOutput = 3, One = 4 };

struct Variable {
    VarKind kind;
    uint32_t index;
};

struct LinearCombination {
    std::vector<std::pair<Variable, Scalar>> terms;
    LinearCombination() {}
    LinearCombination(Variable v) { terms.push_back({v, Scalar::one()}); }                          // From<Variable>
    LinearCombination(Scalar s) { terms.push_back({Variable{VarKind::One, 0}, s}); }                  // From<Scalar>
    LinearCombination operator+(const LinearCombination& o) const {
        LinearCombination r = *this;
        r.terms.insert(r.terms.end(), o.terms.begin(), o.terms.end());
        return r;
    }
    LinearCombination operator-(const LinearCombination& o) const {
        LinearCombination r = *this;
        for (auto& t : o.terms) r.terms.push_back({t.first, -t.second});
        return r;
    }
};

// bulletproofs::r1cs::ConstraintSystem
class ConstraintSystem {
  public:
    virtual ~ConstraintSystem() {}
    struct Mul {
        Variable left, right, out;
    };
    virtual Mul multiply(LinearCombination left, LinearCombination right) = 0;
    virtual void constrain(LinearCombination lc) = 0;
};

constexpr int MIMC_ROUNDS = 90;  // src/gadgets.rs:4

// src/gadgets.rs:37-68
template <class CS>
LinearCombination mimc_gadget(CS& cs, LinearCombination left, LinearCombination right, const std::vector<Scalar>& constants) {
    if ((int)constants.size() != MIMC_ROUNDS) throw std::logic_error("mimc constants");
    LinearCombination x = left;
    LinearCombination key = right;
    for (int i = 0; i < MIMC_ROUNDS; i++) {
        LinearCombination a = x + key + LinearCombination(constants[i]);
        Variable a_2 = cs.multiply(a, a).out;
        Variable a_3 = cs.multiply(LinearCombination(a_2), a).out;
        Variable a_4 = cs.multiply(LinearCombination(a_2), LinearCombination(a_2)).out;
        Variable a_7 = cs.multiply(LinearCombination(a_4), LinearCombination(a_3)).out;
        x = LinearCombination(a_7);
    }
    return x + key;
}

// src/gadgets.rs:134-140
template <class CS>
void boolean_gadget(CS& cs, LinearCombination a1) {
    LinearCombination a = a1;
    LinearCombination one(Scalar::one());
    Variable c_var = cs.multiply(a, one - a1).out;
    cs.constrain(LinearCombination(c_var));
}

// src/gadgets.rs:88-132
template <class CS>
void one_of_many_gadget(CS& cs, LinearCombination x, const std::vector<Variable>& toggle, const std::vector<LinearCombination>& items) {
    const size_t toggle_len = toggle.size();
    for (auto& t : toggle) boolean_gadget(cs, LinearCombination(t));
    std::vector<LinearCombination> toggle_sum;
    toggle_sum.push_back(LinearCombination(toggle[0]));
    for (size_t i = 1; i < toggle_len; i++) toggle_sum.push_back(toggle_sum[i - 1] + LinearCombination(toggle[i]));
    for (size_t i = 1; i < toggle_len; i++) {
        LinearCombination prev = toggle_sum[i - 1], cur_sum = toggle_sum[i];
        toggle_sum[i] = toggle_sum[i - 1] + LinearCombination(toggle[i]);
        cs.constrain(prev + LinearCombination(toggle[i]) - cur_sum);  // vacuous, but consumes a power of z
    }
    cs.constrain(toggle_sum[toggle_len - 1] - LinearCombination(Scalar::one()));
    for (size_t i = 0; i < toggle_len; i++) {
        Variable left = cs.multiply(items[i], LinearCombination(toggle[i])).out;
        Variable right = cs.multiply(LinearCombination(toggle[i]), x).out;
        cs.constrain(LinearCombination(left) - LinearCombination(right));
    }
}

// src/gadgets.rs:70-86
template <class CS>
void score_gadget(CS& cs, LinearCombination d, LinearCombination y, LinearCombination y_inv, LinearCombination q) {
    Variable one_var = cs.multiply(y, y_inv).out;
    cs.constrain(LinearCombination(one_var) - LinearCombination(Scalar::one()));
    Variable q_var = cs.multiply(d, y_inv).out;
    cs.constrain(q - LinearCombination(q_var));
}

// src/gadgets.rs:6-34
template <class CS>
void proof_gadget(CS& cs, LinearCombination d, LinearCombination k, LinearCombination y_inv, LinearCombination q,
                  LinearCombination z_img, LinearCombination seed, const std::vector<Scalar>& constants,
                  const std::vector<Variable>& toggle, const std::vector<LinearCombination>& items) {
    LinearCombination m = mimc_gadget(cs, k, LinearCombination(Scalar::zero()), constants);
    LinearCombination x = mimc_gadget(cs, d, m, constants);
    one_of_many_gadget(cs, x, toggle, items);
    LinearCombination y = mimc_gadget(cs, seed, x, constants);
    LinearCombination z = mimc_gadget(cs, seed, m, constants);
    cs.constrain(z_img - z);
    score_gadget(cs, d, y, y_inv, q);
}

// ---- compiled form -------------------------------------------------------------------------------------------
// term word: [31:29] VarKind, [28] negative, [27:0] index (variable index, or constant-table index for One)
inline uint32_t pack_term(VarKind k, bool neg, uint32_t idx) { return ((uint32_t)k << 29) | ((uint32_t)neg << 28) | idx; }

struct Compiled {
    uint32_t n_items = 0, m = 0, n_mul = 0, n_cons = 0, padded = 0;
    // witness program: multiplier i evaluates left = terms[l_off[i]..r_off[i]), right = terms[r_off[i]..l_off[i+1])
    std::vector<uint32_t> w_terms, w_loff, w_roff;
    // flatten gather: target t in [wL(n_mul) | wR(n_mul) | wO(n_mul) | wV(m)] sums sign * z^(q+1) over f_ent[f_off[t]..f_off[t+1])
    std::vector<uint32_t> f_off, f_ent;  // entry: [31] negative, [30:0] constraint index q
    // constant part: wc = - sum sign * cst * z^(q+1)
    std::vector<uint32_t> c_q, c_cst;    // c_q: [31] negative, [30:0] q
};

// ConstraintSystem that records structure only (the role bulletproofs' Prover/Verifier play during synthesis)
class Recorder : public ConstraintSystem {
  public:
    std::vector<LinearCombination> constraints;
    std::vector<std::pair<LinearCombination, LinearCombination>> muls;
    Mul multiply(LinearCombination left, LinearCombination right) override {
        uint32_t i = (uint32_t)muls.size();
        Variable l{VarKind::MultiplierLeft, i}, r{VarKind::MultiplierRight, i}, o{VarKind::MultiplierOutput, i};
        muls.push_back({left, right});
        // A.3: each multiply pushes two constraints, left - L_i = 0 and right - R_i = 0
        left.terms.push_back({l, -Scalar::one()});
        right.terms.push_back({r, -Scalar::one()});
        constrain(std::move(left));
        constrain(std::move(right));
        return {l, r, o};
    }
    void constrain(LinearCombination lc) override { constraints.push_back(std::move(lc)); }
};

inline uint32_t next_pow2(uint32_t n) {
    uint32_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

// Runs the driver exactly as Proof::prove / Verify::verify do (src/blindbid/proof.rs:55-85, verify.rs:54-85):
// commits d,k,y,y_inv then N toggle bits; gadget wired with vars[0], vars[1], vars[3].
inline Compiled compile(uint32_t n_items) {
    if (n_items == 0) throw std::invalid_argument("empty bid list (reference panics at src/gadgets.rs:103)");
    Recorder cs;
    std::vector<Variable> vars, t_v;
    for (uint32_t i = 0; i < 4; i++) vars.push_back({VarKind::Committed, i});
    for (uint32_t i = 0; i < n_items; i++) t_v.push_back({VarKind::Committed, 4 + i});
    std::vector<LinearCombination> l_v;
    for (uint32_t i = 0; i < n_items; i++) l_v.push_back(LinearCombination(Scalar{1, CST_ITEM0 + i}));
    std::vector<Scalar> constants;
    for (uint32_t i = 0; i < MIMC_ROUNDS; i++) constants.push_back(Scalar{1, CST_MIMC0 + i});
    proof_gadget(cs, LinearCombination(vars[0]), LinearCombination(vars[1]), LinearCombination(vars[3]),
                 LinearCombination(Scalar{1, CST_Q}), LinearCombination(Scalar{1, CST_ZIMG}), LinearCombination(Scalar{1, CST_SEED}),
                 constants, t_v, l_v);

    Compiled c;
    c.n_items = n_items;
    c.m = 4 + n_items;
    c.n_mul = (uint32_t)cs.muls.size();
    c.n_cons = (uint32_t)cs.constraints.size();
    c.padded = next_pow2(c.n_mul);
    auto emit = [&](const LinearCombination& lc) {
        for (auto& t : lc.terms) {
            if (t.first.kind == VarKind::One) {
                c.w_terms.push_back(pack_term(VarKind::One, t.second.sign < 0, t.second.cst));
            } else {
                if (t.second.cst != CST_ONE) throw std::logic_error("variable coefficient is not +-1");
                c.w_terms.push_back(pack_term(t.first.kind, t.second.sign < 0, t.first.index));
            }
        }
    };
    for (auto& mu : cs.muls) {
        c.w_loff.push_back((uint32_t)c.w_terms.size());
        emit(mu.first);
        c.w_roff.push_back((uint32_t)c.w_terms.size());
        emit(mu.second);
    }
    c.w_loff.push_back((uint32_t)c.w_terms.size());
    const uint32_t n_tgt = 3 * c.n_mul + c.m;
    std::vector<std::vector<uint32_t>> per(n_tgt);
    for (uint32_t q = 0; q < c.n_cons; q++) {
        for (auto& t : cs.constraints[q].terms) {
            bool neg = t.second.sign < 0;
            switch (t.first.kind) {
                case VarKind::MultiplierLeft: per[t.first.index].push_back(q | ((uint32_t)neg << 31)); break;
                case VarKind::MultiplierRight: per[c.n_mul + t.first.index].push_back(q | ((uint32_t)neg << 31)); break;
                case VarKind::MultiplierOutput: per[2 * c.n_mul + t.first.index].push_back(q | ((uint32_t)neg << 31)); break;
                case VarKind::Committed: per[3 * c.n_mul + t.first.index].push_back(q | ((uint32_t)(!neg) << 31)); break;  // wV -= e*c
                case VarKind::One:
                    c.c_q.push_back(q | ((uint32_t)neg << 31));
                    c.c_cst.push_back(t.second.cst);
                    break;
            }
            if (t.first.kind != VarKind::One && t.second.cst != CST_ONE) throw std::logic_error("variable coefficient is not +-1");
        }
    }
    for (uint32_t t = 0; t < n_tgt; t++) {
        c.f_off.push_back((uint32_t)c.f_ent.size());
        c.f_ent.insert(c.f_ent.end(), per[t].begin(), per[t].end());
    }
    c.f_off.push_back((uint32_t)c.f_ent.size());
    return c;
}

}  // namespace circuit
}  // namespace bbp
