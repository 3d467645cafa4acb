// ORACLE — TEST INFRASTRUCTURE ONLY (see kb_oracle.hpp header).
//
// GKR for a sum of fractions sum_i n_i / d_i (crates/sub_protocols/src/quotient_gkr/{mod,layers,sumcheck_utils}.rs),
// restated in NATURAL index order on fully padded (power-of-two) vectors.  The reference stores chunk-bit-reversed,
// SIMD-packed active prefixes and accounts for the implicit (0, 1) padding symbolically (sumcheck_utils.rs:136,225,331);
// both are layout optimisations — every transcript value is a field element defined by the natural-order sums below.
#pragma once
#include "whir_oracle.hpp"

namespace orc {

static const size_t N_VARS_TO_SEND_GKR_COEFFS = 5;  // sub_protocols/src/lib.rs:14

struct GkrLayer {
    std::vector<EF> nums, dens;
};
// layers.rs:124-153 (natural variant): (n0 d1 + n1 d0, d0 d1) over adjacent pairs
static inline GkrLayer gkr_layer_up(const GkrLayer& in) {
    GkrLayer out;
    size_t m = in.nums.size() / 2;
    out.nums.resize(m);
    out.dens.resize(m);
#pragma omp parallel for schedule(static) if (m >= 4096)
    for (size_t i = 0; i < m; i++) {
        out.nums[i] = ef_add(ef_mul(in.dens[2 * i + 1], in.nums[2 * i]), ef_mul(in.dens[2 * i], in.nums[2 * i + 1]));
        out.dens[i] = ef_mul(in.dens[2 * i], in.dens[2 * i + 1]);
    }
    return out;
}
// sumcheck_utils.rs:491-503
static inline std::vector<EF> build_bare_from_coeffs(EF c0_raw, EF c2_raw, EF eq_alpha, EF sum, EF mmf) {
    EF c0 = ef_mul(c0_raw, mmf), c2 = ef_mul(c2_raw, mmf);
    EF h1 = ef_mul(ef_sub(sum, ef_mul(ef_sub(ef_one(), eq_alpha), c0)), ef_inv(eq_alpha));
    EF c1 = ef_sub(ef_sub(h1, c0), c2);
    return {c0, c1, c2};
}

// prove_gkr_layer + run_phase2_sumcheck (mod.rs:80-141, sumcheck_utils.rs:278-357): K = claim_point.size() rounds,
// LSB first.  Returns the new point (K+1 coords) and the two claims.
static inline void gkr_prove_layer(ProverState& ps, const GkrLayer& layer, std::vector<EF>& point, EF& claim_num, EF& claim_den) {
    ps.duplex();
    EF alpha = ps.sample();
    EF sum = ef_add(claim_num, ef_mul(alpha, claim_den));
    EF mmf = ef_one();
    size_t m = layer.nums.size() / 2;
    std::vector<EF> nl(m), nr(m), dl(m), dr(m);
    for (size_t i = 0; i < m; i++) {
        nl[i] = layer.nums[2 * i];
        nr[i] = layer.nums[2 * i + 1];
        dl[i] = layer.dens[2 * i];
        dr[i] = layer.dens[2 * i + 1];
    }
    std::vector<EF> remaining = point, q;
    size_t K = point.size();
    for (size_t round = 0; round < K; round++) {
        EF eq_alpha = remaining.back();
        std::vector<EF> eqt = eq_table(remaining.data(), remaining.size() - 1, ef_one());
        EF c0n = ef_zero(), c2n = ef_zero(), c0d = ef_zero(), c2d = ef_zero();
        const size_t n_pairs = nl.size() / 2;
#pragma omp parallel for schedule(static) reduction(efsum : c0n, c2n, c0d, c2d) if (n_pairs >= 2048)
        for (size_t j = 0; j < n_pairs; j++) {
            // pair_coeffs, sumcheck_utils.rs:65-79
            EF dl0 = dl[2 * j], dl1 = dl[2 * j + 1], dr0 = dr[2 * j], dr1 = dr[2 * j + 1];
            EF nl0 = nl[2 * j], nl1 = nl[2 * j + 1], nr0 = nr[2 * j], nr1 = nr[2 * j + 1];
            EF a0 = ef_mul(dl0, dr0), a2 = ef_mul(ef_sub(dl1, dl0), ef_sub(dr1, dr0));
            EF b0 = ef_add(ef_mul(nl0, dr0), ef_mul(nr0, dl0));
            EF b2 = ef_add(ef_mul(ef_sub(nl1, nl0), ef_sub(dr1, dr0)), ef_mul(ef_sub(nr1, nr0), ef_sub(dl1, dl0)));
            c0d = ef_add(c0d, ef_mul(a0, eqt[j]));
            c2d = ef_add(c2d, ef_mul(a2, eqt[j]));
            c0n = ef_add(c0n, ef_mul(b0, eqt[j]));
            c2n = ef_add(c2n, ef_mul(b2, eqt[j]));
        }
        std::vector<EF> bare = build_bare_from_coeffs(ef_add(c0n, ef_mul(alpha, c0d)), ef_add(c2n, ef_mul(alpha, c2d)), eq_alpha, sum, mmf);
        ps.add_sumcheck_polynomial(bare, &eq_alpha);
        EF r = ps.sample();
        EF eq_eval = ef_add(ef_mul(ef_sub(ef_one(), eq_alpha), ef_sub(ef_one(), r)), ef_mul(eq_alpha, r));
        sum = ef_mul(eq_eval, poly_eval(bare, r));
        mmf = ef_mul(mmf, eq_eval);
        size_t h = nl.size() / 2;
        {  // fold into fresh vectors (an in-place fold cannot be split across threads)
            std::vector<EF> tnl(h), tnr(h), tdl(h), tdr(h);
#pragma omp parallel for schedule(static) if (h >= 2048)
            for (size_t j = 0; j < h; j++) {
                tnl[j] = ef_add(nl[2 * j], ef_mul(r, ef_sub(nl[2 * j + 1], nl[2 * j])));
                tnr[j] = ef_add(nr[2 * j], ef_mul(r, ef_sub(nr[2 * j + 1], nr[2 * j])));
                tdl[j] = ef_add(dl[2 * j], ef_mul(r, ef_sub(dl[2 * j + 1], dl[2 * j])));
                tdr[j] = ef_add(dr[2 * j], ef_mul(r, ef_sub(dr[2 * j + 1], dr[2 * j])));
            }
            nl.swap(tnl);
            nr.swap(tnr);
            dl.swap(tdl);
            dr.swap(tdr);
        }
        q.push_back(r);
        remaining.pop_back();
    }
    std::vector<EF> qn(q.rbegin(), q.rend());
    std::vector<EF> inner{nl[0], nr[0], dl[0], dr[0]};
    ps.add_extension_scalars(inner);
    EF beta = ps.sample();
    EF omb = ef_sub(ef_one(), beta);
    claim_num = ef_add(ef_mul(omb, inner[0]), ef_mul(beta, inner[1]));
    claim_den = ef_add(ef_mul(omb, inner[2]), ef_mul(beta, inner[3]));
    qn.push_back(beta);
    point = qn;
}

// prove_gkr_quotient (mod.rs:31-78).  nums: base words, dens: EF, both of length 2^n_vars (already padded with (0,1)).
static inline void gkr_prove(ProverState& ps, const uint32_t* nums, const EF* dens, size_t n_vars, EF& quotient,
                             std::vector<EF>& point, EF& claim_num, EF& claim_den) {
    assert(n_vars > N_VARS_TO_SEND_GKR_COEFFS);
    std::vector<GkrLayer> layers(1);
    size_t len = (size_t)1 << n_vars;
    layers[0].nums.resize(len);
    layers[0].dens.assign(dens, dens + len);
    for (size_t i = 0; i < len; i++) layers[0].nums[i] = ef_from_base(nums[i]);
    size_t cur = n_vars;
    while (cur > N_VARS_TO_SEND_GKR_COEFFS) {
        layers.push_back(gkr_layer_up(layers.back()));
        cur--;
    }
    GkrLayer top = std::move(layers.back());
    layers.pop_back();
    ps.add_extension_scalars(top.nums);
    ps.add_extension_scalars(top.dens);
    quotient = ef_zero();
    for (size_t i = 0; i < top.nums.size(); i++) quotient = ef_add(quotient, ef_mul(top.nums[i], ef_inv(top.dens[i])));
    point = ps.sample_vec(N_VARS_TO_SEND_GKR_COEFFS);
    claim_num = mle_eval_ext(top.nums.data(), N_VARS_TO_SEND_GKR_COEFFS, point.data());
    claim_den = mle_eval_ext(top.dens.data(), N_VARS_TO_SEND_GKR_COEFFS, point.data());
    for (size_t i = layers.size(); i-- > 0;) gkr_prove_layer(ps, layers[i], point, claim_num, claim_den);
}

// verify_gkr_quotient (mod.rs:147-190) + sumcheck_verify (sumcheck/src/verify.rs:5-27)
static inline void gkr_verify(VerifierState& vs, size_t n_vars, EF& quotient, std::vector<EF>& point, EF& claim_num, EF& claim_den) {
    size_t send = (size_t)1 << N_VARS_TO_SEND_GKR_COEFFS;
    std::vector<EF> ln = vs.next_extension_scalars_vec(send), ld = vs.next_extension_scalars_vec(send);
    quotient = ef_zero();
    for (size_t i = 0; i < send; i++) quotient = ef_add(quotient, ef_mul(ln[i], ef_inv(ld[i])));
    point = vs.sample_vec(N_VARS_TO_SEND_GKR_COEFFS);
    claim_num = mle_eval_ext(ln.data(), N_VARS_TO_SEND_GKR_COEFFS, point.data());
    claim_den = mle_eval_ext(ld.data(), N_VARS_TO_SEND_GKR_COEFFS, point.data());
    for (size_t nv = N_VARS_TO_SEND_GKR_COEFFS; nv < n_vars; nv++) {
        vs.duplex();
        EF alpha = vs.sample();
        EF target = ef_add(claim_num, ef_mul(alpha, claim_den));
        std::vector<EF> challenges;
        for (size_t round = 0; round < nv; round++) {
            EF eq_alpha = point[nv - 1 - round];  // eq_alphas_rev
            std::vector<EF> coeffs = vs.next_sumcheck_polynomial(4, target, &eq_alpha);
            EF ch = vs.sample();
            challenges.push_back(ch);
            target = poly_eval(coeffs, ch);
        }
        std::vector<EF> pp(challenges.rbegin(), challenges.rend());
        std::vector<EF> ie = vs.next_extension_scalars_vec(4);
        EF ce = ef_add(ef_mul(ef_mul(alpha, ie[2]), ie[3]), ef_add(ef_mul(ie[0], ie[3]), ef_mul(ie[1], ie[2])));
        EF eqv = ef_one();
        for (size_t j = 0; j < nv; j++)
            eqv = ef_mul(eqv, ef_add(ef_mul(point[j], pp[j]), ef_mul(ef_sub(ef_one(), point[j]), ef_sub(ef_one(), pp[j]))));
        if (!ef_eq(target, ef_mul(eqv, ce))) throw std::runtime_error("InvalidProof (gkr layer)");
        EF beta = vs.sample();
        EF omb = ef_sub(ef_one(), beta);
        claim_num = ef_add(ef_mul(omb, ie[0]), ef_mul(beta, ie[1]));
        claim_den = ef_add(ef_mul(omb, ie[2]), ef_mul(beta, ie[3]));
        pp.push_back(beta);
        point = pp;
    }
}

}  // namespace orc
