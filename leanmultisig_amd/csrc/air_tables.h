// AIR constraint evaluators of the three leanVM tables, generic over the value type T of a column value:
//   T = kb::u32  (base field: the first sumcheck round runs on the committed base columns)
//   T = kb::EF   (extension field: after the first fold)
// Each evaluator returns sum_k alpha^k * C_k(flat, shift) in EF, constraint order identical to the reference
// (alpha power index = order of the assert_* calls):
//   execution     crates/lean_vm/src/tables/execution/air.rs:56-129
//   extension_op  crates/lean_vm/src/tables/extension_op/air.rs:59-163
//   poseidon_16   crates/lean_vm/src/tables/poseidon_16/mod.rs:316-548   (sparse partial rounds, poseidon16.h tables)
//   bus column    crates/lean_vm/src/tables/utils.rs:5-21
#pragma once
#include "kb.h"
#include "poseidon16.h"

namespace air {

using kb::EF;
using kb::u32;
using kb::u64;

static constexpr int T_EXECUTION = 0, T_EXTENSION_OP = 1, T_POSEIDON16 = 2;
static constexpr int MAX_ALPHA = 101;  // max_air_constraints() + 1 (prove_execution.rs:155)

KB_HD constexpr int n_columns(int t) { return t == T_EXECUTION ? 20 : t == T_EXTENSION_OP ? 29 : 109; }
KB_HD constexpr int n_shift(int t) { return t == T_EXECUTION ? 2 : t == T_EXTENSION_OP ? 13 : 0; }
KB_HD constexpr int degree(int t) { return t == T_EXECUTION ? 5 : t == T_EXTENSION_OP ? 6 : 10; }

// ExtraDataForBuses (tables/table_trait.rs:74-95), Montgomery words
struct Extra {
    EF alpha_powers[MAX_ALPHA];
    EF logup_eq[16];  // logup_alphas_eq_poly
    EF bus_beta;
    // Poseidon table, segments 0 / 1 / 3 (see POS_VIRT_O): beta[s][j] = sum_i alpha^(k_s + i) * MDS[i][j]
    EF out_beta[3][16];
};

// ---- small algebra over T ------------------------------------------------------------------------------------------
KB_HD u32 a_add(u32 a, u32 b) { return kb::add(a, b); }
KB_HD u32 a_sub(u32 a, u32 b) { return kb::sub(a, b); }
KB_HD u32 a_mul(u32 a, u32 b) { return kb::mul(a, b); }
KB_HD u32 a_neg(u32 a) { return kb::neg(a); }
KB_HD u32 a_mulc(u32 a, u32 c) { return kb::mul(a, c); }  // times a base constant (Montgomery)
KB_HD u32 a_addc(u32 a, u32 c) { return kb::add(a, c); }
KB_HD EF a_scale(const EF& e, u32 x) { return kb::ef_mul_base(e, x); }  // EF * T
KB_HD EF a_lift(u32 x) { return kb::ef_from_base(x); }
KB_HD u32 a_from_base(u32 c, u32) { return c; }

KB_HD EF a_add(const EF& a, const EF& b) { return kb::ef_add(a, b); }
KB_HD EF a_sub(const EF& a, const EF& b) { return kb::ef_sub(a, b); }
KB_HD EF a_mul(const EF& a, const EF& b) { return kb::ef_mul(a, b); }
KB_HD EF a_neg(const EF& a) { return kb::ef_neg(a); }
KB_HD EF a_mulc(const EF& a, u32 c) { return kb::ef_mul_base(a, c); }
KB_HD EF a_addc(const EF& a, u32 c) { return kb::ef_add_base(a, c); }
KB_HD EF a_scale(const EF& e, const EF& x) { return kb::ef_mul(e, x); }
KB_HD EF a_lift(const EF& x) { return x; }
KB_HD EF a_from_base(u32 c, const EF&) { return kb::ef_from_base(c); }

// small canonical integers in Montgomery form, folded at compile time by the optimiser
KB_HD u32 mc(u32 canon) { return kb::to_monty(canon); }

// keeps the scheduler from hoisting every plane's column loads above the first plane's arithmetic (register pressure)
#if defined(__HIP_DEVICE_COMPILE__)
#define AIR_SCHED_FENCE() ((void)0)
#else
#define AIR_SCHED_FENCE() ((void)0)
#endif

// plane access: an EF value is 5 base-field planes, a base value is 1
template <class T>
struct Planes;
template <>
struct Planes<u32> {
    static constexpr int N = 1;
    static KB_HD u32& at(u32& v, int) { return v; }
};
template <>
struct Planes<EF> {
    static constexpr int N = 5;
    static KB_HD u32& at(EF& v, int k) { return v.v[k]; }
};
using kb::IntC;
using kb::static_for;

// sum_k alpha^k * C_k.
// Base-field rounds (T = u32): DELAYED reduction — five 64-bit accumulators (one per coefficient plane) take the raw
// products alpha^k[i] * value; kb::fold32 keeps them below 2^64 (`room` = products per plane that still fit), one Montgomery
// reduction per plane at the end (5 multiply-adds per constraint instead of 5 multiplications + 5 additions: -15 % VALU).
// Extension-field rounds keep plain ef_mul / ef_add: the same scheme there (25 mads + 2 folds per plane) was measured
// SLOWER (90.5 k vs 87.3 k VALU per Poseidon evaluation, +16 % time: ten more live VGPRs in kernels that already spill).
template <class T>
struct Folder;
template <>
struct Folder<u32> {
    const Extra& x;
    u64 a[5];
    int k, room;
    KB_HD explicit Folder(const Extra& e) : x(e), k(0), room(4) {
#pragma unroll
        for (int i = 0; i < 5; i++) a[i] = 0;
    }
    KB_HD void ensure(int n) {
        if (room < n) {
#pragma unroll
            for (int i = 0; i < 5; i++) a[i] = kb::fold32(a[i]);
            room = 3;
        }
    }
    KB_HD void assert_zero(const u32& v) {
        ensure(1);
#pragma unroll
        for (int i = 0; i < 5; i++) a[i] += (u64)x.alpha_powers[k].v[i] * v;
        room -= 1;
        k++;
    }
    KB_HD void assert_zero_ef(const EF& v) {  // (the bus column is an EF value even in the base round)
        const EF t = kb::ef_mul(x.alpha_powers[k], v);
        ensure(1);
#pragma unroll
        for (int i = 0; i < 5; i++) a[i] += (u64)t.v[i] * kb::ONE;  // = t[i] * R: the final reduction divides by R again
        room -= 1;
        k++;
    }
    KB_HD EF result() const {
        EF r;
#pragma unroll
        for (int i = 0; i < 5; i++) r.v[i] = kb::reduce(kb::fold32(a[i]));
        return r;
    }
};
template <>
struct Folder<EF> {
    const Extra& x;
    EF acc;
    int k;
    KB_HD explicit Folder(const Extra& e) : x(e), acc(kb::ef_zero()), k(0) {}
    KB_HD void assert_zero(const EF& v) {
        acc = kb::ef_add(acc, kb::ef_mul(x.alpha_powers[k], v));
        k++;
    }
    KB_HD void assert_zero_ef(const EF& v) { assert_zero(v); }
    KB_HD EF result() const { return acc; }
};

// (sum_{i<4} eq[i] * data[i] + eq[15] * DOMAINSEP) * beta + flag        (LOGUP_PRECOMPILE_DOMAINSEP = 1)
template <class T>
KB_HD EF bus_column(const Extra& x, const T& flag, const T& d0, const T& d1, const T& d2, const T& d3) {
    EF s = a_scale(x.logup_eq[0], d0);
    s = kb::ef_add(s, a_scale(x.logup_eq[1], d1));
    s = kb::ef_add(s, a_scale(x.logup_eq[2], d2));
    s = kb::ef_add(s, a_scale(x.logup_eq[3], d3));
    s = kb::ef_add(s, x.logup_eq[15]);
    return kb::ef_add(kb::ef_mul(s, x.bus_beta), a_lift(flag));
}

template <class T>
KB_HD T bool_check(const T& v) {  // (1 - v) * v  (backend/field/src/field.rs:197-210)
    return a_mul(a_sub(a_from_base(kb::ONE, v), v), v);
}

// ---- execution ---------------------------------------------------------------------------------------------------------
template <class T>
KB_HD EF eval_execution(const T* flat, const T* shift, const Extra& x) {
    const T one = a_from_base(kb::ONE, flat[0]);
    const T pc = flat[0], fp = flat[1], addr_a = flat[2], addr_b = flat[3], addr_c = flat[4];
    const T value_a = flat[5], value_b = flat[6], value_c = flat[7];
    const T operand_a = flat[8], operand_b = flat[9], operand_c = flat[10];
    const T flag_a = flat[11], flag_b = flat[12], flag_c = flat[13], flag_c_fp = flat[14], flag_ab_fp = flat[15];
    const T mul = flat[16], jump = flat[17], aux = flat[18], precompile_data = flat[19];
    const T pc_shift = shift[0], fp_shift = shift[1];
    const T omfa = a_sub(one, a_add(flag_a, flag_ab_fp));
    const T omfb = a_sub(one, a_add(flag_b, flag_ab_fp));
    const T omfc = a_sub(one, a_add(flag_c, flag_c_fp));
    const T fpa = a_add(fp, operand_a), fpb = a_add(fp, operand_b), fpc = a_add(fp, operand_c);
    const T nu_a = a_add(a_add(a_mul(flag_a, operand_a), a_mul(omfa, value_a)), a_mul(flag_ab_fp, fpa));
    const T nu_b = a_add(a_add(a_mul(flag_b, operand_b), a_mul(omfb, value_b)), a_mul(flag_ab_fp, fpb));
    const T nu_c = a_add(a_add(a_mul(flag_c, operand_c), a_mul(omfc, value_c)), a_mul(flag_c_fp, fpc));
    const T add_ = a_sub(a_add(aux, aux), a_mul(aux, aux));
    const T deref = a_mulc(a_mul(aux, a_sub(aux, one)), mc((kb::P + 1) / 2));
    const T is_precompile = a_sub(one, a_add(a_add(a_add(add_, mul), deref), jump));
    Folder<T> f(x);
    f.assert_zero_ef(bus_column<T>(x, is_precompile, precompile_data, nu_a, nu_b, nu_c));
    f.assert_zero(a_mul(omfa, a_sub(addr_a, fpa)));
    f.assert_zero(a_mul(omfb, a_sub(addr_b, fpb)));
    f.assert_zero(a_mul(omfc, a_sub(addr_c, fpc)));
    f.assert_zero(a_mul(add_, a_sub(nu_b, a_add(nu_a, nu_c))));
    f.assert_zero(a_mul(mul, a_sub(nu_b, a_mul(nu_a, nu_c))));
    // constraints that share a gate: the alpha-weighted sum of the differences is multiplied by the gate once
    // (sum_k alpha^k g d_k = g sum_k alpha^k d_k, exact)
    Folder<T> f_deref(x), f_jc(x), f_njc(x);
    f_deref.k = 6;
    f_deref.assert_zero(a_sub(addr_b, a_add(value_a, operand_b)));
    f_deref.assert_zero(a_sub(value_b, nu_c));
    const T jc = a_mul(jump, nu_a);
    f_jc.k = 8;
    f_jc.assert_zero(a_sub(nu_a, one));
    f_jc.assert_zero(a_sub(pc_shift, nu_b));
    f_jc.assert_zero(a_sub(fp_shift, nu_c));
    const T njc = a_sub(one, jc);
    f_njc.k = 11;
    f_njc.assert_zero(a_sub(pc_shift, a_add(pc, one)));
    f_njc.assert_zero(a_sub(fp_shift, fp));
    return kb::ef_add(kb::ef_add(f.result(), a_scale(f_deref.result(), deref)),
                      kb::ef_add(a_scale(f_jc.result(), jc), a_scale(f_njc.result(), njc)));
}

// quintic product with plain dot products (extension_op/air.rs:37-42)
template <class T>
KB_HD void quintic_mul_air(const T a[5], const T b[5], T out[5]) {
    const T b0m3 = a_sub(b[0], b[3]), b1m4 = a_sub(b[1], b[4]), b4m2 = a_sub(b[4], b[2]), b3m14 = a_sub(b[3], b1m4);
    auto dot = [&](const T& r0, const T& r1, const T& r2, const T& r3, const T& r4) {
        return a_add(a_add(a_add(a_add(a_mul(a[0], r0), a_mul(a[1], r1)), a_mul(a[2], r2)), a_mul(a[3], r3)), a_mul(a[4], r4));
    };
    out[0] = dot(b[0], b[4], b[3], b[2], b1m4);
    out[1] = dot(b[1], b[0], b[4], b[3], b[2]);
    out[2] = dot(b[2], b1m4, b0m3, b4m2, b3m14);
    out[3] = dot(b[3], b[2], b1m4, b0m3, b4m2);
    out[4] = dot(b[4], b[3], b[2], b1m4, b0m3);
}

// ---- extension_op ------------------------------------------------------------------------------------------------------
// PART < 0: all 34 constraints.  PART 0..3: a subset — the constraint sum is additive, so the small (latency-bound) sumcheck
// rounds evaluate the four parts in four workgroups side by side (lm_air.hip), each with only the prerequisites it needs:
//   0: bus, the five boolean checks, the `start` block (alpha^21..25), the seven chaining constraints and the last one  (no product)
//   1: the `add` and `mul` blocks (alpha^6..15): one quintic product
//   2: the `poly_eq` block for coefficients 0..2 (alpha^16..18): two dependent quintic products, three of the five outputs
//   3: the `poly_eq` block for coefficients 3, 4 (alpha^19, 20)
// The longest part is ~56 extension multiplications deep instead of ~132.
static constexpr int EXT_PARTS = 4;
template <class T, int PART = -1>
KB_HD EF eval_extension_op(const T* flat, const T* shift, const Extra& x) {
    constexpr bool ALL = PART < 0;
    const T one = a_from_base(kb::ONE, flat[0]);
    const T is_be = flat[0], start = flat[1], len = flat[2], flag_add = flat[3], flag_mul = flat[4], flag_poly_eq = flat[5];
    const T idx_a = flat[6], idx_b = flat[7], idx_r = flat[13];
    T comp[5], va[5], vb[5], vres[5], comp_shift[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        comp[k] = flat[8 + k];
        va[k] = flat[14 + k];
        vb[k] = flat[19 + k];
        vres[k] = flat[24 + k];
        comp_shift[k] = shift[8 + k];
    }
    const T start_shift = shift[1];
    Folder<T> f(x);
    const T is_ee = a_sub(one, is_be);
    const T nss = a_sub(one, start_shift);
    if constexpr (ALL || PART == 0) {
        const T activation_flag = a_mul(start, a_add(a_add(flag_add, flag_mul), flag_poly_eq));
        const T aux = a_add(a_add(a_add(a_add(a_mulc(is_be, mc(4)), a_mulc(flag_add, mc(8))), a_mulc(flag_mul, mc(16))),
                                  a_mulc(flag_poly_eq, mc(32))),
                            a_mulc(len, mc(64)));
        f.k = 0;
        f.assert_zero_ef(bus_column<T>(x, activation_flag, aux, idx_a, idx_b, idx_r));
        f.assert_zero(bool_check(is_be));
        f.assert_zero(bool_check(start));
        f.assert_zero(bool_check(flag_add));
        f.assert_zero(bool_check(flag_mul));
        f.assert_zero(bool_check(flag_poly_eq));
    }
    T vaf[5], comp_tail[5], vavb[5];
    if constexpr (ALL || PART >= 1) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            vaf[k] = k == 0 ? va[0] : a_mul(va[k], is_ee);
            comp_tail[k] = a_mul(comp_shift[k], nss);
        }
        quintic_mul_air<T>(vaf, vb, vavb);
    }
    if constexpr (ALL || PART == 1) {
        f.k = 6;
#pragma unroll
        for (int k = 0; k < 5; k++) f.assert_zero(a_mul(a_sub(comp[k], a_add(a_add(vaf[k], vb[k]), comp_tail[k])), flag_add));
#pragma unroll
        for (int k = 0; k < 5; k++) f.assert_zero(a_mul(a_sub(comp[k], a_add(vavb[k], comp_tail[k])), flag_mul));
    }
    if constexpr (ALL || PART == 2 || PART == 3) {
        T pev[5], csoo[5], per[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const T base = a_sub(a_sub(a_add(vavb[k], vavb[k]), vaf[k]), vb[k]);
            pev[k] = k == 0 ? a_add(base, one) : base;
            csoo[k] = k == 0 ? a_add(comp_tail[0], start_shift) : comp_tail[k];
        }
        quintic_mul_air<T>(pev, csoo, per);  // (a part uses some of the five outputs: the others are dead code there)
        constexpr int K0 = PART == 3 ? 3 : 0, K1 = PART == 2 ? 3 : 5;
        f.k = 16 + K0;
#pragma unroll
        for (int k = K0; k < K1; k++) f.assert_zero(a_mul(a_sub(comp[k], per[k]), flag_poly_eq));
    }
    if constexpr (ALL || PART == 0) {
        f.k = 21;
#pragma unroll
        for (int k = 0; k < 5; k++) f.assert_zero(a_mul(a_sub(comp[k], vres[k]), start));
        f.assert_zero(a_mul(nss, a_sub(a_sub(len, shift[2]), one)));
        f.assert_zero(a_mul(nss, a_sub(is_be, shift[0])));
        f.assert_zero(a_mul(nss, a_sub(flag_add, shift[3])));
        f.assert_zero(a_mul(nss, a_sub(flag_mul, shift[4])));
        f.assert_zero(a_mul(nss, a_sub(flag_poly_eq, shift[5])));
        const T a_inc = a_add(is_be, a_mulc(is_ee, mc(5)));
        f.assert_zero(a_mul(nss, a_sub(a_sub(shift[6], idx_a), a_inc)));
        f.assert_zero(a_mul(nss, a_sub(a_sub(shift[7], idx_b), a_from_base(mc(5), one))));
        f.assert_zero(a_mul(start_shift, a_sub(len, one)));
    }
    return f.result();
}

// ---- poseidon_16 -------------------------------------------------------------------------------------------------------
KB_HD void mds16(u32 s[16]) { kb::mds_circ16(s); }
KB_HD void mds16(EF s[16]) {  // base-field matrix acts on each coefficient plane
#pragma unroll
    for (int k = 0; k < 5; k++) {
        u32 p[16];
#pragma unroll
        for (int i = 0; i < 16; i++) p[i] = s[i].v[k];
        kb::mds_circ16(p);
#pragma unroll
        for (int i = 0; i < 16; i++) s[i].v[k] = p[i];
    }
}
template <class T>
KB_HD T cube(const T& a) {
    return a_mul(a_mul(a, a), a);
}

// Column access is lazy: `col(c)` returns the value of column c at this evaluation point, so that only the 16-word state
// and the current block of columns are live (109 columns x 5 words would not fit the register file).
//
// The AIR re-bases its state on committed columns after every pair of full rounds, so it splits into four independent
// SEGMENTS, each starting from columns only; a (row pair, evaluation point) is evaluated by 5 lanes, one per segment, and
// the alpha-weighted partial sums are added afterwards (the constraint sum is linear in the segments):
//   segment 0: bus + 7 flag constraints (alpha^0..7), inputs -> beginning_full_rounds[0]        (alpha^8..23)
//   segment 1: beginning_full_rounds[0] -> beginning_full_rounds[1]                               (alpha^24..39)
//   segment 2: the 20 partial-round constraints through the affine forms y_r                       (alpha^40..59)
//   segment 3: affine exit state of the partial block -> ending_full_rounds[0]                     (alpha^60..75)
//   segment 4: ending_full_rounds[0] -> gated outputs                                              (alpha^76..99)
// This cuts the dependent-instruction chain of one evaluation ~5x (the late, small sumcheck rounds are latency bound).
static constexpr int POSEIDON_SEGMENTS = 5;
// Virtual columns of the Poseidon table: the affine forms of the partial block (below) evaluated row by row.  They are
// linear in committed columns, so they fold like columns; the constraint kernels then read 36 values instead of
// recomputing 16 x 36 + 20 x ~26 multiply-adds per plane at every evaluation point of every round.
static constexpr int POS_VIRT_Y = 109;       // 20 columns: value cubed in partial round r
static constexpr int POS_VIRT_E = 109 + 20;  // 16 columns: state entering ending_full_rounds
// Output blocks of segments 0, 1, 3 (s = 0, 1, 2): their 16 constraints  MDS(y)_i - out_i  (y = the S-box layer of the
// segment's second full round, out = 16 committed columns) enter the round polynomial only through
//     sum_i alpha^(k_s+i) (MDS(y)_i - out_i)  =  sum_j beta_sj y_j  -  V_s,      V_s = sum_i alpha^(k_s+i) out_i,
// and V_s is LINEAR in committed columns: it is computed once per table and folded like a column.  Per evaluation this
// replaces one MDS, 16 column reads and 16 challenge-weighted products by 16 products and one column read (exact field
// identity, the same round polynomials).  V_s is extension-field valued; its 5 coefficient planes are 5 virtual BASE
// columns (column POS_VIRT_O + 5 s + k), so the base round reads them as a value and every fold treats them like any
// column: after folding, V_s = sum_k X^k * (folded plane k).
static constexpr int POS_VIRT_O = 109 + 36;
static constexpr int POS_N_VIRT = 36 + 15;
static constexpr int POS_OUT_K0[3] = {8, 24, 60};    // alpha exponent of the first output constraint of segments 0, 1, 3
static constexpr int POS_OUT_COL[3] = {25, 41, 77};  // first of their 16 output columns

// Affine forms of the partial block (gen_poseidon_consts.py::linearise): lanes 1..15 see no S-box inside the block and lane 0
// is re-based on the committed partial_rounds[r] column every round, so over u = (t_0..t_15, q_0..q_19, 1) with
// t = beginning_full_rounds[1] columns and q = partial_rounds columns, the value cubed in round r and the state leaving the
// block are affine.  Same constraint polynomials as the round-by-round substitution of poseidon_16/mod.rs:430-470, but every
// constraint becomes one dot product with delayed reduction and the 20 rounds no longer form a dependent chain.
// The tables are constexpr so that, with every index a compile-time constant, each coefficient becomes an instruction
// literal (s_mov) next to its use.  As a __constant__ table the ~1500 scalar loads are hoisted to the top of the kernel and
// spilled to VGPR lanes (thousands of v_readlane).
struct PoseidonLinear {
    u32 y[20][37];    // y_r = y[r][0..16+r) . u + y[r][36]
    u32 fin[16][37];  // state entering ending_full_rounds
};
static constexpr PoseidonLinear kPoseidonLinear =
#include "poseidon16_linear.inc"
    ;

// (static_for, not "#pragma unroll": the pragma gives up on the EF body size, and a rolled loop indexes s[] dynamically,
// which sends the whole state to scratch)
// full rounds R0 and R0 + 1 of the 8 (0..3 initial, 4..7 terminal); round constants are instruction literals
template <class T, int R>
KB_HD void full_round(T s[16]) {
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr u32 rc = R < 4 ? kb::kPoseidonHost.rc_init[R & 3][i] : kb::kPoseidonHost.rc_term[R & 3][i];
        s[i] = cube(a_addc(s[i], rc));
    });
    mds16(s);
}

// A segment of the Poseidon AIR (see above) in two halves.  (The state behind the first half has degree 3 in the row variable, so it
// could be evaluated at 4 points and extrapolated to the round's 10 — not built: the 16 x 4 extension values to extrapolate from do
// not fit a lane's registers next to the evaluation, and through LDS or across lanes the saving is ~20 % of the large rounds)
//   seg_first  : the 16 input columns of the segment through full round R0 (S-box + MDS)
//   seg_finish : [segment 0: bus + flag constraints] full round R0 + 1, then the segment's constraints
template <int SEG>
struct SegInfo {
    static constexpr int input_col = SEG == 0 ? 9 : SEG == 1 ? 25 : SEG == 3 ? POS_VIRT_E : 77;
    static constexpr int round0 = SEG == 0 ? 0 : SEG == 1 ? 2 : SEG == 3 ? 4 : 6;
};
template <class T, int SEG, class ColFn>
KB_HD void seg_first(ColFn col, T s[16]) {
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = col(SegInfo<SEG>::input_col + i);
    full_round<T, SegInfo<SEG>::round0>(s);
}
// p * X^k in the basis 1, X, .., X^4 (X^5 = 1 - X^2): (a0..a4) * X = (a4, a0, a1 - a4, a2, a3)
template <int K>
KB_HD EF ef_mul_xk(EF a) {
#pragma unroll
    for (int t = 0; t < K; t++) {
        const EF b = a;
        a.v[0] = b.v[4], a.v[1] = b.v[0], a.v[2] = kb::sub(b.v[1], b.v[4]), a.v[3] = b.v[2], a.v[4] = b.v[3];
    }
    return a;
}
// V_s at the evaluation point from its 5 plane columns
template <class T, int S, class ColFn>
KB_HD EF virt_out(ColFn col) {
    if constexpr (sizeof(T) == sizeof(u32)) {
        EF v;
        static_for<0, 5>([&](auto K) { v.v[decltype(K)::value] = (u32)col(POS_VIRT_O + 5 * S + decltype(K)::value); });
        return v;
    } else {
        EF v = col(POS_VIRT_O + 5 * S);
        static_for<1, 5>([&](auto K) { v = kb::ef_add(v, ef_mul_xk<decltype(K)::value>(col(POS_VIRT_O + 5 * S + decltype(K)::value))); });
        return v;
    }
}
// sum_j beta_sj * cube(s_j + rc_j) - V_s: the output block of segment slot S whose second full round is R
template <class T, int S, int R, class ColFn>
KB_HD EF out_block(const T s[16], ColFn col, const Extra& x) {
    const EF v = virt_out<T, S>(col);
    if constexpr (sizeof(T) == sizeof(u32)) {
        // base round: 16 x 5 multiply-adds into 64-bit accumulators, folded every 4 / 3 products (kb::fold32)
        u64 a[5] = {0, 0, 0, 0, 0};
        static_for<0, 16>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr u32 rc = R < 4 ? kb::kPoseidonHost.rc_init[R & 3][j] : kb::kPoseidonHost.rc_term[R & 3][j];
            const u32 y = cube(a_addc(s[j], rc));
            if constexpr (j >= 4 && (j - 4) % 3 == 0) {
#pragma unroll
                for (int k = 0; k < 5; k++) a[k] = kb::fold32(a[k]);
            }
#pragma unroll
            for (int k = 0; k < 5; k++) a[k] += (u64)x.out_beta[S][j].v[k] * y;
        });
        EF r;
#pragma unroll
        for (int k = 0; k < 5; k++) r.v[k] = kb::reduce(kb::fold32(a[k]));
        return kb::ef_sub(r, v);
    } else {
        EF acc = kb::ef_neg(v);
        static_for<0, 16>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr u32 rc = R < 4 ? kb::kPoseidonHost.rc_init[R & 3][j] : kb::kPoseidonHost.rc_term[R & 3][j];
            acc = kb::ef_add(acc, kb::ef_mul(x.out_beta[S][j], cube(a_addc(s[j], rc))));
        });
        return acc;
    }
}
template <class T, int SEG, class ColFn>
KB_HD EF seg_finish(T s[16], ColFn col, const Extra& x) {
    Folder<T> f(x);
    if constexpr (SEG == 0) {
        const T flag_active = col(0), index_b = col(1), index_res = col(2), flag_half = col(3), flag_left = col(4);
        const T offset_left = col(5), eff_first = col(6), eff_second = col(7), flag_permute = col(8);
        const T one = a_from_base(kb::ONE, flag_active);
        // precompile data: 1 + 4 half + 8 left + 16 left*offset + 2 permute (poseidon_16/mod.rs:94-98,336-343)
        const T pdr = a_add(a_add(a_add(a_add(one, a_mulc(flag_half, mc(4))), a_mulc(flag_left, mc(8))),
                                  a_mulc(a_mul(flag_left, offset_left), mc(16))),
                            a_mulc(flag_permute, mc(2)));
        const T omfl = a_sub(one, flag_left);
        const T index_a = a_sub(eff_second, a_mulc(omfl, mc(4)));
        f.assert_zero_ef(bus_column<T>(x, flag_active, pdr, index_a, index_b, index_res));
        f.assert_zero(bool_check(flag_active));
        f.assert_zero(bool_check(flag_half));
        f.assert_zero(bool_check(flag_left));
        f.assert_zero(bool_check(flag_permute));
        f.assert_zero(a_mul(flag_permute, a_add(flag_half, flag_left)));
        f.assert_zero(a_mul(flag_left, a_sub(offset_left, eff_first)));
        f.assert_zero(a_mul(omfl, a_sub(index_a, eff_first)));
        return kb::ef_add(f.result(), out_block<T, 0, 1>(s, col, x));  // alpha^8..23: beginning_full_rounds[0]
    } else if constexpr (SEG == 1) {
        return out_block<T, 1, 3>(s, col, x);  // alpha^24..39: beginning_full_rounds[1]
    } else if constexpr (SEG == 3) {
        return out_block<T, 2, 5>(s, col, x);  // alpha^60..75: ending_full_rounds[0]
    } else {
        static_assert(SEG == 4, "segment 2 has no full rounds");
        f.k = 76;
        full_round<T, 7>(s);
        const T flag_half = col(3), flag_permute = col(8);
        const T one = a_from_base(kb::ONE, flag_half);
        const T not_permute = a_sub(one, flag_permute);
        const T comp_last4 = a_sub(not_permute, flag_half);
        // The 24 output constraints are gate * difference with only three distinct gates: the alpha-weighted sums are taken
        // per gate and multiplied by the gate once (sum_k alpha^k g d_k = g sum_k alpha^k d_k, exact field arithmetic):
        // 24 + 3 multiplications instead of 48.  (Unrolled by hand: a rolled loop would index s[] dynamically.)
        Folder<T> f_np(x), f_c4(x), f_fp(x);
        auto out_row = [&](auto I) {
            constexpr int i = decltype(I)::value;
            Folder<T>& fg = i < 4 ? f_np : f_c4;
            const T ol = col(93 + i);
            fg.k = 76 + 3 * i;
            fg.assert_zero(a_sub(a_add(s[i], col(9 + i)), ol));
            f_fp.k = 77 + 3 * i;
            f_fp.assert_zero(a_sub(s[i], ol));
            f_fp.assert_zero(a_sub(s[i + 8], col(101 + i)));
        };
        out_row(IntC<0>{}), out_row(IntC<1>{}), out_row(IntC<2>{}), out_row(IntC<3>{});
        out_row(IntC<4>{}), out_row(IntC<5>{}), out_row(IntC<6>{}), out_row(IntC<7>{});
        return kb::ef_add(kb::ef_add(a_scale(f_np.result(), not_permute), a_scale(f_c4.result(), comp_last4)),
                          a_scale(f_fp.result(), flag_permute));
    }
    return f.result();
}

// ---- sumcheck round 1 read through the first challenge (lm_air.hip: FoldCols) --------------------------------------------------
// There a column value at the evaluation point is  a + t b  with BASE-FIELD a, b and t = the first challenge (the table after its
// first fold, not materialised).  So the S-box layer of a segment's FIRST full round is a cubic in t with base coefficients,
//     (a + t b)^3 = a^3 + 3 a^2 b t + 3 a b^2 t^2 + b^3 t^3 :
// six base products instead of two extension products (50 multiply-adds), the MDS acts on its FOUR coefficient planes instead of
// five, and the state enters the extension field only in front of the second S-box layer, s = c0 + c1 t + c2 t^2 + c3 t^3 (three
// base-by-extension products per word).  Segment 2's constraints y_r^3 - q_r are cubics in t as well: their alpha-weighted sum is taken
// coefficient by coefficient (base-by-extension multiply-adds, delayed reduction) and meets t, t^2, t^3 once per row pair.
// Exact identities: the same field values as the extension-field evaluation (tests/test_air_gpu.py, proofs word-identical).
struct Ab {
    u32 a, b;
};
struct TPowers {
    EF t1, t2, t3;
};
struct CubeCoeffs {
    u32 c0, c1, c2, c3;
};
KB_HD CubeCoeffs cube_affine(u32 a, u32 b) {
    const u32 a2 = kb::mul(a, a), b2 = kb::mul(b, b);
    const u32 a2b = kb::mul(a2, b), ab2 = kb::mul(a, b2);
    CubeCoeffs c;
    c.c0 = kb::mul(a2, a);
    c.c1 = kb::add(a2b, kb::dbl(a2b));
    c.c2 = kb::add(ab2, kb::dbl(ab2));
    c.c3 = kb::mul(b2, b);
    return c;
}
template <int SEG, class AbFn>
KB_HD void seg_first_affine(AbFn ab, const TPowers& tp, EF s[16]) {
    constexpr int R = SegInfo<SEG>::round0;
    u32 c0[16], c1[16], c2[16], c3[16];
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr u32 rc = R < 4 ? kb::kPoseidonHost.rc_init[R & 3][i] : kb::kPoseidonHost.rc_term[R & 3][i];
        const Ab v = ab(SegInfo<SEG>::input_col + i);
        const CubeCoeffs c = cube_affine(kb::add(v.a, rc), v.b);
        c0[i] = c.c0, c1[i] = c.c1, c2[i] = c.c2, c3[i] = c.c3;
    });
    kb::mds_circ16(c0);
    kb::mds_circ16(c1);
    kb::mds_circ16(c2);
    kb::mds_circ16(c3);
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const u64 acc = (u64)c1[i] * tp.t1.v[k] + (u64)c2[i] * tp.t2.v[k] + (u64)c3[i] * tp.t3.v[k];  // < 3 p^2 < 2^64
            s[i].v[k] = kb::reduce(kb::fold32(acc));
        }
        s[i].v[0] = kb::add(s[i].v[0], c0[i]);
    });
}
// col(c): the extension-field value of column c (FoldCols::at); ab(c): its two base coefficients
template <int SEG, class ColFn, class AbFn>
KB_HD EF eval_poseidon16_segment_affine(ColFn col, AbFn ab, const TPowers& tp, const Extra& x) {
    if constexpr (SEG == 2) {
        u64 acc[4][5];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int k = 0; k < 5; k++) acc[m][k] = 0;
        static_for<0, 20>([&](auto RR) {
            constexpr int r = decltype(RR)::value;
            const Ab y = ab(POS_VIRT_Y + r), q = ab(57 + r);
            const CubeCoeffs c = cube_affine(y.a, y.b);
            const u32 d[4] = {kb::sub(c.c0, q.a), kb::sub(c.c1, q.b), c.c2, c.c3};  // y_r^3 - partial_rounds[r], coefficient by coefficient
            if constexpr (r >= 4 && (r - 4) % 3 == 0) {  // four products fit from zero, three after a fold
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int k = 0; k < 5; k++) acc[m][k] = kb::fold32(acc[m][k]);
            }
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int k = 0; k < 5; k++) acc[m][k] += (u64)x.alpha_powers[40 + r].v[k] * d[m];
        });
        EF a[4];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int k = 0; k < 5; k++) a[m].v[k] = kb::reduce(kb::fold32(acc[m][k]));
        return kb::ef_add(kb::ef_add(a[0], kb::ef_mul(tp.t1, a[1])), kb::ef_add(kb::ef_mul(tp.t2, a[2]), kb::ef_mul(tp.t3, a[3])));
    } else {
        EF s[16];
        seg_first_affine<SEG>(ab, tp, s);
        return seg_finish<EF, SEG>(s, col, x);
    }
}

// ---- the execution table in round 1, on polynomials in the first challenge -----------------------------------------------------
// Same situation as above: every column value is a + t b with base-field a, b.  The 13 constraints have degree <= 5 in the columns,
// so each is a polynomial of degree <= 5 in t with BASE coefficients: the evaluator below works on such polynomials (Pt<D>: D + 1 base
// words; a product of degrees A and B is (A + 1)(B + 1) base multiply-adds with delayed reduction instead of 25 per extension
// product), weights coefficient m of constraint k with alpha^k into one of six extension-field accumulators (5 multiply-adds per
// coefficient, delayed reduction), and meets t, .., t^5 once per row pair:  sum_k alpha^k C_k = sum_m t^m (sum_k alpha^k c_km).
// The bus value enters the same accumulators: (sum_i eq_i d_i + eq_15) beta + flag = sum_m t^m (sum_i (eq_i beta) d_im + flag_m) + eq_15 beta.
// Constraint order and polynomials are eval_execution's (execution/air.rs:56-129); exact field identities throughout.
template <int D>
struct Pt {
    u32 c[D + 1];
};
KB_HD Pt<1> pt_of(const Ab& v) {
    Pt<1> r;
    r.c[0] = v.a, r.c[1] = v.b;
    return r;
}
template <int A, int B>
KB_HD Pt<(A > B ? A : B)> pt_add(const Pt<A>& a, const Pt<B>& b) {
    Pt<(A > B ? A : B)> r;
    static_for<0, (A > B ? A : B) + 1>([&](auto M) {
        constexpr int m = decltype(M)::value;
        if constexpr (m <= A && m <= B)
            r.c[m] = kb::add(a.c[m], b.c[m]);
        else if constexpr (m <= A)
            r.c[m] = a.c[m];
        else
            r.c[m] = b.c[m];
    });
    return r;
}
template <int A, int B>
KB_HD Pt<(A > B ? A : B)> pt_sub(const Pt<A>& a, const Pt<B>& b) {
    Pt<(A > B ? A : B)> r;
    static_for<0, (A > B ? A : B) + 1>([&](auto M) {
        constexpr int m = decltype(M)::value;
        if constexpr (m <= A && m <= B)
            r.c[m] = kb::sub(a.c[m], b.c[m]);
        else if constexpr (m <= A)
            r.c[m] = a.c[m];
        else
            r.c[m] = kb::neg(b.c[m]);
    });
    return r;
}
template <int A>
KB_HD Pt<A> pt_one_minus(const Pt<A>& a) {  // 1 - a
    Pt<A> r;
    r.c[0] = kb::sub(kb::ONE, a.c[0]);
    static_for<1, A + 1>([&](auto M) { r.c[decltype(M)::value] = kb::neg(a.c[decltype(M)::value]); });
    return r;
}
template <int A>
KB_HD Pt<A> pt_sub_one(const Pt<A>& a) {  // a - 1
    Pt<A> r = a;
    r.c[0] = kb::sub(a.c[0], kb::ONE);
    return r;
}
template <int A>
KB_HD Pt<A> pt_scale(const Pt<A>& a, u32 k) {
    Pt<A> r;
    static_for<0, A + 1>([&](auto M) { r.c[decltype(M)::value] = kb::mul(a.c[decltype(M)::value], k); });
    return r;
}
// adds a * b into the 64-bit coefficient accumulators x[0 .. A + B]; n[m] counts the products x[m] holds (four fit from zero, three
// after a fold: kb::fold32)
template <int A, int B, int N>
KB_HD void pt_mul_acc(const Pt<A>& a, const Pt<B>& b, u64 (&x)[N], int (&n)[N]) {
    static_assert(A + B < N, "accumulator count");
    static_for<0, A + 1>([&](auto I) {
        static_for<0, B + 1>([&](auto J) {
            constexpr int m = decltype(I)::value + decltype(J)::value;
            if (n[m] == 4) {
                x[m] = kb::fold32(x[m]);
                n[m] = 1;
            }
            x[m] += (u64)a.c[decltype(I)::value] * b.c[decltype(J)::value];
            n[m]++;
        });
    });
}
template <int D, int N>
KB_HD Pt<D> pt_reduce(u64 (&x)[N], const int (&n)[N]) {
    Pt<D> r;
    static_for<0, D + 1>([&](auto M) {
        constexpr int m = decltype(M)::value;
        r.c[m] = kb::reduce(n[m] <= 2 ? x[m] : kb::fold32(x[m]));  // two products of reduced values stay below 2^32 p
    });
    return r;
}
template <int A, int B>
KB_HD Pt<A + B> pt_mul(const Pt<A>& a, const Pt<B>& b) {
    u64 x[A + B + 1];
    int n[A + B + 1];
#pragma unroll
    for (int m = 0; m <= A + B; m++) x[m] = 0, n[m] = 0;
    pt_mul_acc(a, b, x, n);
    return pt_reduce<A + B>(x, n);
}
// a1 b1 + a2 b2 + a3 b3 (degree 1 each): nu_a, nu_b, nu_c
KB_HD Pt<2> pt_dot3(const Pt<1>& a1, const Pt<1>& b1, const Pt<1>& a2, const Pt<1>& b2, const Pt<1>& a3, const Pt<1>& b3) {
    u64 x[3] = {0, 0, 0};
    int n[3] = {0, 0, 0};
    pt_mul_acc(a1, b1, x, n);
    pt_mul_acc(a2, b2, x, n);
    pt_mul_acc(a3, b3, x, n);
    return pt_reduce<2>(x, n);
}
struct ExecAffine {
    EF tp[5];       // t, t^2, .., t^5
    EF eq_beta[4];  // logup_eq[i] * bus_beta
    EF eq15_beta;   // logup_eq[15] * bus_beta  (LOGUP_PRECOMPILE_DOMAINSEP = 1)
};
struct PtFolder {
    const Extra& x;
    u64 a[6][5];
    int n[6];
    KB_HD explicit PtFolder(const Extra& e) : x(e) {
#pragma unroll
        for (int m = 0; m < 6; m++) {
            n[m] = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) a[m][k] = 0;
        }
    }
    KB_HD void room(int m) {
        if (n[m] == 4) {
#pragma unroll
            for (int k = 0; k < 5; k++) a[m][k] = kb::fold32(a[m][k]);
            n[m] = 1;
        }
    }
    template <int D>
    KB_HD void weigh(const EF& w, const Pt<D>& v) {  // += w * v, coefficient by coefficient
        static_for<0, D + 1>([&](auto M) {
            constexpr int m = decltype(M)::value;
            room(m);
#pragma unroll
            for (int k = 0; k < 5; k++) a[m][k] += (u64)w.v[k] * v.c[m];
            n[m]++;
        });
    }
    template <int D>
    KB_HD void assert_zero(int k, const Pt<D>& v) { weigh(x.alpha_powers[k], v); }
    template <int D>
    KB_HD void add_base(const Pt<D>& v) {  // += v (alpha^0 = 1: the flag of the bus value), plane 0 only
        static_for<0, D + 1>([&](auto M) {
            constexpr int m = decltype(M)::value;
            room(m);
            a[m][0] += (u64)v.c[m] * kb::ONE;  // (= v R: the final reduction divides by R again)
            n[m]++;
        });
    }
    KB_HD EF result(const ExecAffine& ex) const {
        EF r;
#pragma unroll
        for (int k = 0; k < 5; k++) r.v[k] = kb::reduce(kb::fold32(a[0][k]));
        r = kb::ef_add(r, ex.eq15_beta);
        static_for<1, 6>([&](auto M) {
            constexpr int m = decltype(M)::value;
            EF c;
#pragma unroll
            for (int k = 0; k < 5; k++) c.v[k] = kb::reduce(kb::fold32(a[m][k]));
            r = kb::ef_add(r, kb::ef_mul(ex.tp[m - 1], c));
        });
        return r;
    }
};
KB_HD EF eval_execution_affine(const Ab* flat, const Ab* shift, const Extra& x, const ExecAffine& ex) {
    const Pt<1> pc = pt_of(flat[0]), fp = pt_of(flat[1]), addr_a = pt_of(flat[2]), addr_b = pt_of(flat[3]), addr_c = pt_of(flat[4]);
    const Pt<1> value_a = pt_of(flat[5]), value_b = pt_of(flat[6]), value_c = pt_of(flat[7]);
    const Pt<1> operand_a = pt_of(flat[8]), operand_b = pt_of(flat[9]), operand_c = pt_of(flat[10]);
    const Pt<1> flag_a = pt_of(flat[11]), flag_b = pt_of(flat[12]), flag_c = pt_of(flat[13]), flag_c_fp = pt_of(flat[14]), flag_ab_fp = pt_of(flat[15]);
    const Pt<1> mul = pt_of(flat[16]), jump = pt_of(flat[17]), aux = pt_of(flat[18]), precompile_data = pt_of(flat[19]);
    const Pt<1> pc_shift = pt_of(shift[0]), fp_shift = pt_of(shift[1]);
    const Pt<1> omfa = pt_one_minus(pt_add(flag_a, flag_ab_fp)), omfb = pt_one_minus(pt_add(flag_b, flag_ab_fp)), omfc = pt_one_minus(pt_add(flag_c, flag_c_fp));
    const Pt<1> fpa = pt_add(fp, operand_a), fpb = pt_add(fp, operand_b), fpc = pt_add(fp, operand_c);
    const Pt<2> nu_a = pt_dot3(flag_a, operand_a, omfa, value_a, flag_ab_fp, fpa);
    const Pt<2> nu_b = pt_dot3(flag_b, operand_b, omfb, value_b, flag_ab_fp, fpb);
    const Pt<2> nu_c = pt_dot3(flag_c, operand_c, omfc, value_c, flag_c_fp, fpc);
    const Pt<2> aux2 = pt_mul(aux, aux);
    const Pt<2> add_ = pt_sub(pt_add(aux, aux), aux2);                       // 2 aux - aux^2
    const Pt<2> deref = pt_scale(pt_sub(aux2, aux), mc((kb::P + 1) / 2));    // aux (aux - 1) / 2
    const Pt<2> is_precompile = pt_one_minus(pt_add(pt_add(add_, mul), pt_add(deref, jump)));
    PtFolder f(x);
    // alpha^0: the bus value
    f.weigh(ex.eq_beta[0], precompile_data);
    f.weigh(ex.eq_beta[1], nu_a);
    f.weigh(ex.eq_beta[2], nu_b);
    f.weigh(ex.eq_beta[3], nu_c);
    f.add_base(is_precompile);
    f.assert_zero(1, pt_mul(omfa, pt_sub(addr_a, fpa)));
    f.assert_zero(2, pt_mul(omfb, pt_sub(addr_b, fpb)));
    f.assert_zero(3, pt_mul(omfc, pt_sub(addr_c, fpc)));
    f.assert_zero(4, pt_mul(add_, pt_sub(nu_b, pt_add(nu_a, nu_c))));
    f.assert_zero(5, pt_mul(mul, pt_sub(nu_b, pt_mul(nu_a, nu_c))));
    f.assert_zero(6, pt_mul(deref, pt_sub(addr_b, pt_add(value_a, operand_b))));
    f.assert_zero(7, pt_mul(deref, pt_sub(value_b, nu_c)));
    const Pt<3> jc = pt_mul(jump, nu_a);
    f.assert_zero(8, pt_mul(jc, pt_sub_one(nu_a)));
    f.assert_zero(9, pt_mul(jc, pt_sub(pc_shift, nu_b)));
    f.assert_zero(10, pt_mul(jc, pt_sub(fp_shift, nu_c)));
    const Pt<3> njc = pt_one_minus(jc);
    Pt<1> pc_next = pc;
    pc_next.c[0] = kb::add(pc.c[0], kb::ONE);
    f.assert_zero(11, pt_mul(njc, pt_sub(pc_shift, pc_next)));
    f.assert_zero(12, pt_mul(njc, pt_sub(fp_shift, fp)));
    return f.result(ex);
}

// col(c) = column c at the evaluation point
template <class T, int SEG, class ColFn>
KB_HD EF eval_poseidon16_segment(ColFn col, const Extra& x) {
    if constexpr (SEG == 2) {
        Folder<T> f(x);
        f.k = 40;
        // y_r (the value cubed in partial round r) is an affine form of committed columns: it is read as virtual column
        // POS_VIRT_Y + r, computed once per table and folded with the others (lm_air.hip: k_air_virtual_columns)
        static_for<0, 20>([&](auto RR) {
            constexpr int r = decltype(RR)::value;
            f.assert_zero(a_sub(cube(col(POS_VIRT_Y + r)), col(57 + r)));  // assert_eq_low(state[0]^3, partial_rounds[r])
        });
        return f.result();
    } else {
        T s[16];
        seg_first<T, SEG>(col, s);
        return seg_finish<T, SEG>(s, col, x);
    }
}

}  // namespace air
