// ORACLE — TEST INFRASTRUCTURE ONLY (see kb_oracle.hpp header).
//
// AIR constraint polynomials of the three leanVM tables and the batched, back-loaded AIR sumcheck:
//   crates/lean_vm/src/tables/execution/air.rs:56-129          (20 + 2 shift columns, degree 5, 13 constraints)
//   crates/lean_vm/src/tables/extension_op/air.rs:59-163       (29 + 13 shift columns, degree 6, 33 constraints)
//   crates/lean_vm/src/tables/poseidon_16/mod.rs:316-548       (109 columns, degree 10, 100 constraints)
//   crates/lean_vm/src/tables/utils.rs:5-21                    (virtual bus column)
//   crates/backend/air/src/constraint_folder/normal.rs:48-62   (constraint k weighted by alpha_powers[k])
//   crates/sub_protocols/src/air_sumcheck.rs:225-292,560-681   (round polynomials, back-loaded batching)
// Everything is evaluated over EF in NATURAL row order on the full padded domain.  The reference's chunk-bit-reversed
// packed storage, the padding shortcut (air_sumcheck.rs:194-200,236-240) and the Poseidon low-degree split (:403-557)
// are optimisations of exactly these sums.
// The Poseidon AIR is written with the TEXTBOOK partial rounds (add full round constants, substitute lane 0, dense
// MDS): its constraint polynomials equal the reference's sparse-form ones as polynomials in the columns (the sparse
// factorisation only re-associates linear maps around a lane-0-only operation; checked numerically by
// leanmultisig_amd/csrc/gen_poseidon_consts.py and tests/test_air_gpu.py).
#pragma once
#include "gkr_oracle.hpp"

namespace orc {

struct AirExtra {
    std::vector<EF> alpha_powers;          // >= n_constraints
    std::vector<EF> logup_alphas_eq_poly;  // 16 entries (prove_execution.rs:125-126)
    EF bus_beta;
};

static inline EF ef_const(uint32_t canon) { return ef_from_base(to_monty(canon)); }

struct Folder {
    const AirExtra& x;
    EF acc = ef_zero();
    size_t k = 0;
    explicit Folder(const AirExtra& e) : x(e) {}
    void assert_zero(const EF& v) {
        acc = ef_add(acc, ef_mul(x.alpha_powers[k], v));
        k++;
    }
};

// tables/utils.rs:5-21, LOGUP_PRECOMPILE_DOMAINSEP = 1 (core/constants.rs:5)
static inline EF virtual_bus_column(const AirExtra& x, const EF& flag, const EF data[4]) {
    EF s = ef_zero();
    for (int i = 0; i < 4; i++) s = ef_add(s, ef_mul(x.logup_alphas_eq_poly[i], data[i]));
    s = ef_add(s, ef_mul(x.logup_alphas_eq_poly.back(), ef_const(1)));
    return ef_add(ef_mul(s, x.bus_beta), flag);
}

enum AirTable { AIR_EXECUTION = 0, AIR_EXTENSION_OP = 1, AIR_POSEIDON16 = 2 };
static inline size_t air_n_columns(int t) { return t == AIR_EXECUTION ? 20 : t == AIR_EXTENSION_OP ? 29 : 109; }
static inline size_t air_n_shift(int t) { return t == AIR_EXECUTION ? 2 : t == AIR_EXTENSION_OP ? 13 : 0; }
static inline size_t air_degree(int t) { return t == AIR_EXECUTION ? 5 : t == AIR_EXTENSION_OP ? 6 : 10; }
static inline size_t air_n_constraints(int t) { return t == AIR_EXECUTION ? 13 : t == AIR_EXTENSION_OP ? 33 : 100; }

// ---- execution/air.rs:56-129 ---------------------------------------------------------------------------------------
static inline EF air_eval_execution(const EF* flat, const EF* shift, const AirExtra& x) {
    const EF one = ef_one();
    EF pc = flat[0], fp = flat[1], addr_a = flat[2], addr_b = flat[3], addr_c = flat[4];
    EF value_a = flat[5], value_b = flat[6], value_c = flat[7];
    EF operand_a = flat[8], operand_b = flat[9], operand_c = flat[10];
    EF flag_a = flat[11], flag_b = flat[12], flag_c = flat[13], flag_c_fp = flat[14], flag_ab_fp = flat[15];
    EF mul = flat[16], jump = flat[17], aux = flat[18], precompile_data = flat[19];
    EF pc_shift = shift[0], fp_shift = shift[1];
    auto N = [&](const EF& a) { return ef_neg(a); };
    EF omfa = N(ef_sub(ef_add(flag_a, flag_ab_fp), one));
    EF omfb = N(ef_sub(ef_add(flag_b, flag_ab_fp), one));
    EF omfc = N(ef_sub(ef_add(flag_c, flag_c_fp), one));
    EF nu_a = ef_add(ef_add(ef_mul(flag_a, operand_a), ef_mul(omfa, value_a)), ef_mul(flag_ab_fp, ef_add(fp, operand_a)));
    EF nu_b = ef_add(ef_add(ef_mul(flag_b, operand_b), ef_mul(omfb, value_b)), ef_mul(flag_ab_fp, ef_add(fp, operand_b)));
    EF nu_c = ef_add(ef_add(ef_mul(flag_c, operand_c), ef_mul(omfc, value_c)), ef_mul(flag_c_fp, ef_add(fp, operand_c)));
    EF fpa = ef_add(fp, operand_a), fpb = ef_add(fp, operand_b), fpc = ef_add(fp, operand_c);
    EF pc_plus_one = ef_add(pc, one);
    EF nu_a_minus_one = ef_sub(nu_a, one);
    EF add_ = ef_sub(ef_add(aux, aux), ef_mul(aux, aux));
    EF deref = ef_mul_base(ef_mul(aux, ef_sub(aux, one)), inv(add(ONE, ONE)));
    EF is_precompile = N(ef_sub(ef_add(ef_add(ef_add(add_, mul), deref), jump), one));
    Folder f(x);
    EF data[4] = {precompile_data, nu_a, nu_b, nu_c};
    f.assert_zero(virtual_bus_column(x, is_precompile, data));
    f.assert_zero(ef_mul(omfa, ef_sub(addr_a, fpa)));
    f.assert_zero(ef_mul(omfb, ef_sub(addr_b, fpb)));
    f.assert_zero(ef_mul(omfc, ef_sub(addr_c, fpc)));
    f.assert_zero(ef_mul(add_, ef_sub(nu_b, ef_add(nu_a, nu_c))));
    f.assert_zero(ef_mul(mul, ef_sub(nu_b, ef_mul(nu_a, nu_c))));
    f.assert_zero(ef_mul(deref, ef_sub(addr_b, ef_add(value_a, operand_b))));
    f.assert_zero(ef_mul(deref, ef_sub(value_b, nu_c)));
    EF jc = ef_mul(jump, nu_a);
    f.assert_zero(ef_mul(jc, nu_a_minus_one));
    f.assert_zero(ef_mul(jc, ef_sub(pc_shift, nu_b)));
    f.assert_zero(ef_mul(jc, ef_sub(fp_shift, nu_c)));
    EF njc = N(ef_sub(jc, one));
    f.assert_zero(ef_mul(njc, ef_sub(pc_shift, pc_plus_one)));
    f.assert_zero(ef_mul(njc, ef_sub(fp_shift, fp)));
    assert(f.k == 13);
    return f.acc;
}

// quintic_mul with the plain dot product (extension_op/air.rs:37-42; quintic_extension/extension.rs:531-548)
static inline void quintic_mul_air(const EF a[5], const EF b[5], EF out[5]) {
    auto dot = [&](const EF r[5]) {
        EF s = ef_zero();
        for (int i = 0; i < 5; i++) s = ef_add(s, ef_mul(a[i], r[i]));
        return s;
    };
    EF b0m3 = ef_sub(b[0], b[3]), b1m4 = ef_sub(b[1], b[4]), b4m2 = ef_sub(b[4], b[2]);
    EF r0[5] = {b[0], b[4], b[3], b[2], b1m4};
    EF r1[5] = {b[1], b[0], b[4], b[3], b[2]};
    EF r2[5] = {b[2], b1m4, b0m3, b4m2, ef_sub(b[3], b1m4)};
    EF r3[5] = {b[3], b[2], b1m4, b0m3, b4m2};
    EF r4[5] = {b[4], b[3], b[2], b1m4, b0m3};
    out[0] = dot(r0);
    out[1] = dot(r1);
    out[2] = dot(r2);
    out[3] = dot(r3);
    out[4] = dot(r4);
}

// ---- extension_op/air.rs:59-163 --------------------------------------------------------------------------------------
static inline EF air_eval_extension_op(const EF* flat, const EF* shift, const AirExtra& x) {
    const EF one = ef_one();
    EF is_be = flat[0], start = flat[1], len = flat[2], flag_add = flat[3], flag_mul = flat[4], flag_poly_eq = flat[5];
    EF idx_a = flat[6], idx_b = flat[7];
    EF comp[5], va[5], vb[5], vres[5], comp_shift[5];
    for (int k = 0; k < 5; k++) {
        comp[k] = flat[8 + k];
        va[k] = flat[14 + k];
        vb[k] = flat[19 + k];
        vres[k] = flat[24 + k];
        comp_shift[k] = shift[8 + k];
    }
    EF idx_r = flat[13];
    EF is_be_shift = shift[0], start_shift = shift[1], len_shift = shift[2], flag_add_shift = shift[3];
    EF flag_mul_shift = shift[4], flag_poly_eq_shift = shift[5], idx_a_shift = shift[6], idx_b_shift = shift[7];
    EF active = ef_add(ef_add(flag_add, flag_mul), flag_poly_eq);
    EF activation_flag = ef_mul(start, active);
    // EXT_OP_FLAG_IS_BE 4, ADD 8, MUL 16, POLY_EQ 32, LEN_MULTIPLIER 64 (extension_op/mod.rs:10-14)
    EF aux = ef_add(ef_add(ef_add(ef_add(ef_mul(is_be, ef_const(4)), ef_mul(flag_add, ef_const(8))), ef_mul(flag_mul, ef_const(16))),
                           ef_mul(flag_poly_eq, ef_const(32))),
                    ef_mul(len, ef_const(64)));
    Folder f(x);
    EF data[4] = {aux, idx_a, idx_b, idx_r};
    f.assert_zero(virtual_bus_column(x, activation_flag, data));
    EF is_ee = ef_neg(ef_sub(is_be, one));
    EF not_start_shift = ef_neg(ef_sub(start_shift, one));
    EF vaf[5], comp_tail[5];
    for (int k = 0; k < 5; k++) {
        vaf[k] = k == 0 ? va[0] : ef_mul(va[k], is_ee);
        comp_tail[k] = ef_mul(comp_shift[k], not_start_shift);
    }
    auto bool_check = [&](const EF& v) { return ef_mul(ef_sub(one, v), v); };  // (1 - x) * x, field.rs:197-210
    f.assert_zero(bool_check(is_be));
    f.assert_zero(bool_check(start));
    f.assert_zero(bool_check(flag_add));
    f.assert_zero(bool_check(flag_mul));
    f.assert_zero(bool_check(flag_poly_eq));
    for (int k = 0; k < 5; k++) f.assert_zero(ef_mul(ef_sub(comp[k], ef_add(ef_add(vaf[k], vb[k]), comp_tail[k])), flag_add));
    EF vavb[5];
    quintic_mul_air(vaf, vb, vavb);
    for (int k = 0; k < 5; k++) f.assert_zero(ef_mul(ef_sub(comp[k], ef_add(vavb[k], comp_tail[k])), flag_mul));
    EF pev[5], csoo[5], per[5];
    for (int k = 0; k < 5; k++) {
        EF base = ef_sub(ef_sub(ef_add(vavb[k], vavb[k]), vaf[k]), vb[k]);
        pev[k] = k == 0 ? ef_add(base, one) : base;
        csoo[k] = k == 0 ? ef_add(ef_mul(comp_shift[0], not_start_shift), start_shift) : ef_mul(comp_shift[k], not_start_shift);
    }
    quintic_mul_air(pev, csoo, per);
    for (int k = 0; k < 5; k++) f.assert_zero(ef_mul(ef_sub(comp[k], per[k]), flag_poly_eq));
    for (int k = 0; k < 5; k++) f.assert_zero(ef_mul(ef_sub(comp[k], vres[k]), start));
    f.assert_zero(ef_mul(not_start_shift, ef_sub(ef_sub(len, len_shift), one)));
    f.assert_zero(ef_mul(not_start_shift, ef_sub(is_be, is_be_shift)));
    f.assert_zero(ef_mul(not_start_shift, ef_sub(flag_add, flag_add_shift)));
    f.assert_zero(ef_mul(not_start_shift, ef_sub(flag_mul, flag_mul_shift)));
    f.assert_zero(ef_mul(not_start_shift, ef_sub(flag_poly_eq, flag_poly_eq_shift)));
    EF a_inc = ef_add(is_be, ef_mul(is_ee, ef_const(5)));
    f.assert_zero(ef_mul(not_start_shift, ef_sub(ef_sub(idx_a_shift, idx_a), a_inc)));
    f.assert_zero(ef_mul(not_start_shift, ef_sub(ef_sub(idx_b_shift, idx_b), ef_const(5))));
    f.assert_zero(ef_mul(start_shift, ef_sub(len, one)));
    assert(f.k == 34);  // 1 bus + 5 bool + 20 + 8 (the table reports n_constraints() = 33; only max_air_constraints() is used)
    return f.acc;
}

// ---- poseidon_16/mod.rs:316-548 --------------------------------------------------------------------------------------
static inline void ef_mds(EF s[16]) {
    EF o[16];
    for (int i = 0; i < 16; i++) {
        EF acc = ef_zero();
        for (int j = 0; j < 16; j++) acc = ef_add(acc, ef_mul_base(s[j], poseidon_tables().mds[i][j]));
        o[i] = acc;
    }
    for (int i = 0; i < 16; i++) s[i] = o[i];
}
static inline EF ef_cube(const EF& a) { return ef_mul(ef_mul(a, a), a); }
static inline void ef_full_round(EF s[16], int r) {
    for (int i = 0; i < 16; i++) s[i] = ef_cube(ef_add(s[i], ef_from_base(poseidon_tables().rc[r][i])));
    ef_mds(s);
}
static inline EF air_eval_poseidon16(const EF* c, const AirExtra& x) {
    const EF one = ef_one();
    EF flag_active = c[0], index_b = c[1], index_res = c[2], flag_half_output = c[3], flag_hardcoded_left = c[4];
    EF offset_hardcoded_left = c[5], eff_first = c[6], eff_second = c[7], flag_permute = c[8];
    const EF* inputs = c + 9;
    const EF* bfr = c + 25;         // beginning_full_rounds[2][16]
    const EF* partial = c + 57;     // partial_rounds[20]
    const EF* efr = c + 77;         // ending_full_rounds[1][16]
    const EF* out_left = c + 93;    // outputs_left[8]
    const EF* out_right = c + 101;  // outputs_right[8]
    // POSEIDON_*_SHIFT: permute 2, half 4, hardcoded-left flag 8, offset 16 (poseidon_16/mod.rs:94-98)
    EF pdr = ef_add(ef_add(ef_add(ef_add(one, ef_mul(flag_half_output, ef_const(4))), ef_mul(flag_hardcoded_left, ef_const(8))),
                           ef_mul(ef_mul(flag_hardcoded_left, offset_hardcoded_left), ef_const(16))),
                    ef_mul(flag_permute, ef_const(2)));
    EF omfhl = ef_sub(one, flag_hardcoded_left);
    EF index_a = ef_sub(eff_second, ef_mul(omfhl, ef_const(4)));  // HALF_DIGEST_LEN = 4
    Folder f(x);
    EF data[4] = {pdr, index_a, index_b, index_res};
    f.assert_zero(virtual_bus_column(x, flag_active, data));
    auto bool_check = [&](const EF& v) { return ef_mul(ef_sub(one, v), v); };  // (1 - x) * x
    f.assert_zero(bool_check(flag_active));
    f.assert_zero(bool_check(flag_half_output));
    f.assert_zero(bool_check(flag_hardcoded_left));
    f.assert_zero(bool_check(flag_permute));
    f.assert_zero(ef_mul(flag_permute, ef_add(flag_half_output, flag_hardcoded_left)));
    f.assert_zero(ef_mul(flag_hardcoded_left, ef_sub(offset_hardcoded_left, eff_first)));
    f.assert_zero(ef_mul(omfhl, ef_sub(index_a, eff_first)));
    // eval_poseidon1_16
    EF s[16];
    for (int i = 0; i < 16; i++) s[i] = inputs[i];
    int r = 0;
    for (int blk = 0; blk < 2; blk++) {
        ef_full_round(s, r++);
        ef_full_round(s, r++);
        for (int i = 0; i < 16; i++) {
            f.assert_zero(ef_sub(s[i], bfr[blk * 16 + i]));
            s[i] = bfr[blk * 16 + i];
        }
    }
    for (int pr = 0; pr < 20; pr++, r++) {
        for (int i = 0; i < 16; i++) s[i] = ef_add(s[i], ef_from_base(poseidon_tables().rc[r][i]));
        f.assert_zero(ef_sub(ef_cube(s[0]), partial[pr]));
        s[0] = partial[pr];
        ef_mds(s);
    }
    ef_full_round(s, r++);
    ef_full_round(s, r++);
    for (int i = 0; i < 16; i++) {
        f.assert_zero(ef_sub(s[i], efr[i]));
        s[i] = efr[i];
    }
    ef_full_round(s, r++);
    ef_full_round(s, r++);
    assert(r == 28);
    EF not_permute = ef_sub(one, flag_permute);
    EF comp_last4 = ef_sub(not_permute, flag_half_output);
    for (int i = 0; i < 8; i++) {
        EF gate = i < 4 ? not_permute : comp_last4;
        f.assert_zero(ef_mul(gate, ef_sub(ef_add(s[i], inputs[i]), out_left[i])));
        f.assert_zero(ef_mul(flag_permute, ef_sub(s[i], out_left[i])));
        f.assert_zero(ef_mul(flag_permute, ef_sub(s[i + 8], out_right[i])));
    }
    assert(f.k == 100);
    return f.acc;
}

static inline EF air_eval(int table, const EF* flat_and_shift, const AirExtra& x) {
    const size_t nf = air_n_columns(table);
    if (table == AIR_EXECUTION) return air_eval_execution(flat_and_shift, flat_and_shift + nf, x);
    if (table == AIR_EXTENSION_OP) return air_eval_extension_op(flat_and_shift, flat_and_shift + nf, x);
    return air_eval_poseidon16(flat_and_shift, x);
}

// ---- trace helpers ---------------------------------------------------------------------------------------------------
// generate_trace_rows_for_perm (poseidon_16/trace_gen.rs:44-112), textbook form, base field.  row = 109 words, the
// first 9 (flags / indices) and the 16 inputs are given; the rest is filled.
static inline void poseidon16_fill_row(uint32_t* row) {
    const PoseidonTables& t = poseidon_tables();
    uint32_t s[16];
    std::memcpy(s, row + 9, 64);
    int r = 0;
    auto full = [&](int rr) {
        for (int i = 0; i < 16; i++) s[i] = cube(add(s[i], t.rc[rr][i]));
        mds_dense(s);
    };
    for (int blk = 0; blk < 2; blk++) {
        full(r++);
        full(r++);
        std::memcpy(row + 25 + 16 * blk, s, 64);
    }
    for (int pr = 0; pr < 20; pr++, r++) {
        for (int i = 0; i < 16; i++) s[i] = add(s[i], t.rc[r][i]);
        s[0] = cube(s[0]);
        row[57 + pr] = s[0];
        mds_dense(s);
    }
    full(r++);
    full(r++);
    std::memcpy(row + 77, s, 64);
    full(r++);
    full(r++);
    uint32_t fp = row[8];
    for (int i = 0; i < 8; i++) {
        uint32_t cv = add(s[i], row[9 + i]);
        row[93 + i] = add(mul(sub(ONE, fp), cv), mul(fp, s[i]));
        row[101 + i] = mul(fp, s[i + 8]);
    }
}
// compute_shifted_columns (air_sumcheck.rs:683-694)
static inline std::vector<uint32_t> shifted_column(const uint32_t* col, size_t n) {
    std::vector<uint32_t> s(n);
    for (size_t i = 0; i + 1 < n; i++) s[i] = col[i + 1];
    s[n - 1] = col[n - 1];
    return s;
}

// ---- sumcheck session (air_sumcheck.rs:45-292), natural order, full domain -------------------------------------------
struct AirSession {
    int table;
    size_t n_vars;
    std::vector<std::vector<EF>> cols;  // n_columns + n_shift columns, each 2^(n_vars - rounds_done)
    std::vector<EF> eq_factor;          // last element removed each round
    EF sum, mmf;
    AirExtra extra;
    size_t degree() const { return air_degree(table); }
    EF eq_alpha() const { return eq_factor.back(); }
    // compute_bare_round_poly (:225-266)
    std::vector<EF> compute_bare_round_poly() const {
        const size_t deg = degree();
        const size_t pairs = cols[0].size() / 2;
        const size_t nc = cols.size();
        std::vector<EF> eqt = eq_table(eq_factor.data(), eq_factor.size() - 1, ef_one());
        std::vector<EF> acc(deg + 1, ef_zero());  // index = z (z = 1 unused)
        // (OpenMP: per-thread partial sums, merged at the end — exact arithmetic, the order of the sum does not matter)
#pragma omp parallel if (pairs >= 64)
        {
            std::vector<EF> local(deg + 1, ef_zero()), point(nc), diff(nc);
#pragma omp for schedule(static)
            for (size_t j = 0; j < pairs; j++) {
                for (size_t c = 0; c < nc; c++) {
                    point[c] = cols[c][2 * j];
                    diff[c] = ef_sub(cols[c][2 * j + 1], cols[c][2 * j]);
                }
                for (size_t z = 0; z <= deg; z++) {
                    if (z != 1) local[z] = ef_add(local[z], ef_mul(air_eval(table, point.data(), extra), eqt[j]));
                    for (size_t c = 0; c < nc; c++) point[c] = ef_add(point[c], diff[c]);
                }
            }
#pragma omp critical
            for (size_t z = 0; z <= deg; z++) acc[z] = ef_add(acc[z], local[z]);
        }
        std::vector<EF> ev(deg + 1);
        for (size_t z = 0; z <= deg; z++) ev[z] = ef_mul(acc[z], mmf);
        ev[1] = ef_mul(ef_sub(sum, ef_mul(ef_sub(ef_one(), eq_alpha()), ev[0])), ef_inv(eq_alpha()));
        // Lagrange interpolation on the points 0..deg -> coefficients (DensePolynomial::lagrange_interpolation)
        std::vector<EF> coeffs(deg + 1, ef_zero());
        for (size_t i = 0; i <= deg; i++) {
            // numerator polynomial prod_{j != i} (X - j), denominator prod_{j != i} (i - j)
            std::vector<EF> num{ef_one()};
            uint32_t den = ONE;
            for (size_t j = 0; j <= deg; j++) {
                if (j == i) continue;
                std::vector<EF> nn(num.size() + 1, ef_zero());
                EF mj = ef_neg(ef_const((uint32_t)j));
                for (size_t k = 0; k < num.size(); k++) {
                    nn[k + 1] = ef_add(nn[k + 1], num[k]);
                    nn[k] = ef_add(nn[k], ef_mul(num[k], mj));
                }
                num.swap(nn);
                den = mul(den, sub(to_monty((uint32_t)i), to_monty((uint32_t)j)));
            }
            EF scale = ef_mul_base(ev[i], inv(den));
            for (size_t k = 0; k < num.size(); k++) coeffs[k] = ef_add(coeffs[k], ef_mul(num[k], scale));
        }
        return coeffs;
    }
    // process_challenge (:268-292)
    void process_challenge(EF ch, const std::vector<EF>& bare) {
        EF a = eq_alpha();
        EF eq_eval = ef_add(ef_mul(ef_sub(ef_one(), a), ef_sub(ef_one(), ch)), ef_mul(a, ch));
        sum = ef_mul(poly_eval(bare, ch), eq_eval);
        mmf = ef_mul(mmf, eq_eval);
#pragma omp parallel for schedule(dynamic) if (cols.size() >= 4 && cols[0].size() >= 256)
        for (size_t ci = 0; ci < cols.size(); ci++) {  // columns are independent (the fold of one column is in place)
            auto& col = cols[ci];
            size_t h = col.size() / 2;
            for (size_t j = 0; j < h; j++) col[j] = ef_add(col[2 * j], ef_mul(ch, ef_sub(col[2 * j + 1], col[2 * j])));
            col.resize(h);
        }
        eq_factor.pop_back();
    }
    std::vector<EF> final_column_evals() const {
        std::vector<EF> r;
        for (auto& c : cols) r.push_back(c[0]);
        return r;
    }
};

// prove_batched_air_sumcheck (:636-681)
static inline std::vector<EF> prove_batched_air_sumcheck(ProverState& ps, std::vector<AirSession>& sessions, EF eta) {
    size_t n_rounds = 0, max_full_degree = 1;
    for (auto& s : sessions) {
        n_rounds = std::max(n_rounds, s.n_vars);
        max_full_degree = std::max(max_full_degree, s.degree() + 1);
    }
    std::vector<EF> eta_p(sessions.size(), ef_one()), k(sessions.size(), ef_one());
    for (size_t i = 1; i < sessions.size(); i++) eta_p[i] = ef_mul(eta_p[i - 1], eta);
    std::vector<EF> challenges;
    for (size_t round = 0; round < n_rounds; round++) {
        std::vector<EF> combined(max_full_degree + 1, ef_zero());
        std::vector<std::vector<EF>> bare(sessions.size());
        for (size_t i = 0; i < sessions.size(); i++) {
            AirSession& s = sessions[i];
            size_t join = n_rounds - s.n_vars;
            EF w = ef_mul(eta_p[i], k[i]);
            if (round < join) {
                combined[1] = ef_add(combined[1], ef_mul(w, s.sum));
            } else {
                bare[i] = s.compute_bare_round_poly();
                std::vector<EF> full = expand_bare_to_full(bare[i], s.eq_alpha());
                for (size_t c = 0; c < full.size(); c++) combined[c] = ef_add(combined[c], ef_mul(w, full[c]));
            }
        }
        ps.add_sumcheck_polynomial(combined, nullptr);
        EF ch = ps.sample();
        challenges.push_back(ch);
        for (size_t i = 0; i < sessions.size(); i++) {
            size_t join = n_rounds - sessions[i].n_vars;
            if (round < join)
                k[i] = ef_mul(k[i], ch);
            else
                sessions[i].process_challenge(ch, bare[i]);
        }
    }
    return challenges;
}

}  // namespace orc
