// verify_execution (crates/lean_prover/src/verify_execution.rs:14-251) and everything below it, on the host: VerifierState
// (fiat-shamir/src/verifier.rs), PrunedMerklePaths::restore (merkle_pruning.rs:89-170), verify_gkr_quotient
// (sub_protocols/src/quotient_gkr/mod.rs:147-190), verify_generic_logup (logup.rs:326-493), sumcheck_verify
// (sumcheck/src/verify.rs), the AIR check at the sumcheck point, stacked_pcs_global_statements and WhirConfig::verify
// (whir/src/verify.rs:83-435).  SURVEY.md §8(f) rank 3: a shippable checker of the proofs this library produces — the
// reference's verifier runs on the CPU (it is milliseconds of work plus one pass over the bytecode table), so does this one;
// it touches no GPU state and is independent of the test oracle.  The AIR constraint polynomials are the ones the device
// kernels evaluate (csrc/air_tables.h, instantiated for the extension field on the host).
#include <stdio.h>
#include <string>
#include "../air_tables.h"
#include "lm_host_internal.h"

namespace lmh {
const std::vector<u32>& proof_transcript(const lmh_proof* p);
const std::vector<PrunedBatch>& proof_batches(const lmh_proof* p);
}  // namespace lmh

namespace {
using namespace lmh;
using kb::ef_add;
using kb::ef_from_base;
using kb::ef_mul;
using kb::ef_mul_base;
using kb::ef_one;
using kb::ef_sub;
using kb::ef_zero;
using kb::to_monty;

}  // namespace
struct lmh_raw_proof {  // RawProof::transcript (fiat-shamir/src/transcript.rs:20-31) + what the PCS opening was asked to prove
    std::vector<u32> transcript;
    lm_whir_opening_claim claim;
    lm_pcs_statement_claim stmt;
};
namespace {
struct Fail {  // thrown inside this file only; every entry point catches it (nothing unwinds across the ABI)
    std::string why;
};
[[noreturn]] void fail(const char* why) { throw Fail{why}; }
void require(bool ok, const char* why) {
    if (!ok) fail(why);
}

EF ef_load(const u32* p) {
    EF r;
    memcpy(r.v, p, 20);
    return r;
}
u32 ilog2(u64 x) {
    u32 l = 0;
    while ((1ull << (l + 1)) <= x) l++;
    return l;
}
EF one_minus(const EF& a) { return ef_sub(ef_one(), a); }
EF eq_term(const EF& l, const EF& r) {  // l r + (1 - l)(1 - r)
    const EF lr = ef_mul(l, r);
    return ef_sub(ef_sub(ef_add(ef_one(), kb::ef_dbl(lr)), l), r);
}
EF eq_poly_outside(const EF* a, const EF* b, size_t n) {  // poly/src/point.rs:77-89
    EF acc = ef_one();
    for (size_t i = 0; i < n; i++) acc = ef_mul(acc, eq_term(a[i], b[i]));
    return acc;
}
EF poly_eval(const std::vector<EF>& coeffs, const EF& x) {  // DensePolynomial::evaluate (Horner)
    EF acc = ef_zero();
    for (size_t i = coeffs.size(); i-- > 0;) acc = ef_add(ef_mul(acc, x), coeffs[i]);
    return acc;
}
// EvaluationsList::evaluate: evals (base words or AoS EF) at an EF point, point[0] <-> most significant index bit
EF mle_eval(const u32* evals, bool is_ext, const EF* point, u32 n) {
    const u64 len = 1ull << n;
    std::vector<EF> cur(len);
    for (u64 i = 0; i < len; i++) cur[i] = is_ext ? ef_load(evals + 5 * i) : ef_from_base(evals[i]);
    u64 m = len;
    for (u32 j = 0; j < n; j++) {
        const u64 half = m >> 1;
        for (u64 i = 0; i < half; i++) cur[i] = ef_add(cur[i], ef_mul(point[j], ef_sub(cur[i + half], cur[i])));
        m = half;
    }
    return cur[0];
}
// same for a large base table: the first fold is EF x base, done blockwise to keep the working set at half the table
EF mle_eval_base_big(const u32* evals, const EF* point, u32 n) {
    if (n == 0) return ef_from_base(evals[0]);
    const u64 half = 1ull << (n - 1);
    std::vector<EF> cur(half);
    for (u64 i = 0; i < half; i++) {
        EF t = ef_mul_base(point[0], kb::sub(evals[i + half], evals[i]));
        t.v[0] = kb::add(t.v[0], evals[i]);
        cur[i] = t;
    }
    u64 m = half;
    for (u32 j = 1; j < n; j++) {
        const u64 h = m >> 1;
        for (u64 i = 0; i < h; i++) cur[i] = ef_add(cur[i], ef_mul(point[j], ef_sub(cur[i + h], cur[i])));
        m = h;
    }
    return cur[0];
}
std::vector<EF> expand_from_univariate(EF a, u32 n) {  // poly/src/point.rs:51-61
    std::vector<EF> r(n);
    for (u32 i = 0; i < n; i++) {
        r[i] = a;
        a = kb::ef_sqr(a);
    }
    return r;
}
EF mle_of_zeros_then_ones(u64 n_zeros, const EF* point, u32 n) {  // poly/src/mle/mle_custom.rs:4-19
    const u64 n_values = 1ull << n;
    require(n_zeros <= n_values, "mle_of_zeros_then_ones: too many zeros");
    if (n_zeros == 0) return ef_one();
    if (n_zeros == n_values) return ef_zero();
    const u64 half = n_values / 2;
    if (n_zeros < half) return ef_add(ef_mul(one_minus(point[0]), mle_of_zeros_then_ones(n_zeros, point + 1, n - 1)), point[0]);
    return ef_mul(point[0], mle_of_zeros_then_ones(n_zeros - half, point + 1, n - 1));
}
EF mle_of_01234567_etc(const EF* point, u32 n) {  // utils/src/multilinear.rs:67-74
    if (n == 0) return ef_zero();
    const EF e = mle_of_01234567_etc(point + 1, n - 1);
    return ef_add(ef_mul(one_minus(point[0]), e), ef_mul(point[0], kb::ef_add_base(e, to_monty((u32)(1u << (n - 1))))));
}
EF next_mle(const EF* x, const EF* y, u32 n) {  // poly/src/next_mle.rs:9-29
    std::vector<EF> eq_prefix(n + 1), low_suffix(n + 1, ef_one());
    eq_prefix[0] = ef_one();
    for (u32 i = 0; i < n; i++)
        eq_prefix[i + 1] = ef_mul(eq_prefix[i], ef_add(ef_mul(x[i], y[i]), ef_mul(one_minus(x[i]), one_minus(y[i]))));
    for (u32 i = n; i-- > 0;) low_suffix[i] = ef_mul(ef_mul(low_suffix[i + 1], x[i]), one_minus(y[i]));
    EF sum = ef_zero(), all = ef_one();
    for (u32 a = 0; a < n; a++) {
        const EF carry = ef_mul(one_minus(x[a]), y[a]);
        sum = ef_add(sum, ef_mul(ef_mul(eq_prefix[a], carry), low_suffix[a + 1]));
    }
    for (u32 i = 0; i < n; i++) all = ef_mul(all, x[i]);
    for (u32 i = 0; i < n; i++) all = ef_mul(all, y[i]);
    return ef_add(sum, all);
}
EF eval_multilinear_coeffs(const EF* coeffs, u64 len, const EF* point) {  // poly/src/evals.rs:69-82
    if (len == 1) return coeffs[0];
    return ef_add(eval_multilinear_coeffs(coeffs, len / 2, point + 1), ef_mul(eval_multilinear_coeffs(coeffs + len / 2, len / 2, point + 1), point[0]));
}

// ---- hashing (symetric/src/sponge.rs:7-24, compression.rs:5, merkle.rs:92-121) -------------------------------------------
void hash_slice(const u32* data, size_t n, u32 out[8]) {
    require(n % 8 == 0 && n >= 16, "hash_slice: leaf length must be a multiple of 8, at least 16");
    u32 st[16];
    memcpy(st, data + n - 16, 64);
    lmh::host_compress(st);
    for (size_t chunk = n / 8 - 2; chunk-- > 0;) {
        memcpy(st + 8, data + chunk * 8, 32);
        lmh::host_compress(st);
    }
    memcpy(out, st, 32);
}
void compress_pair(const u32 l[8], const u32 r[8], u32 out[8]) {
    u32 st[16];
    memcpy(st, l, 32);
    memcpy(st + 8, r, 32);
    lmh::host_compress(st);
    memcpy(out, st, 32);
}
bool merkle_verify(const u32 root[8], u32 log_height, u64 index, const std::vector<u32>& leaf, const std::vector<u32>& path) {
    if (path.size() != (size_t)log_height * 8) return false;
    u32 h[8];
    hash_slice(leaf.data(), leaf.size(), h);
    for (u32 l = 0; l < log_height; l++) {
        u32 nx[8];
        if ((index & 1) == 0)
            compress_pair(h, &path[8 * l], nx);
        else
            compress_pair(&path[8 * l], h, nx);
        memcpy(h, nx, 32);
        index >>= 1;
    }
    return memcmp(h, root, 32) == 0;
}
unsigned lca_level(u64 a, u64 b) {
    unsigned l = 0;
    for (u64 x = a ^ b; x; x >>= 1) l++;
    return l;
}
// PrunedMerklePaths::restore (merkle_pruning.rs:89-170): openings in the prover's original order
std::vector<Opening> restore(const PrunedBatch& b) {
    const size_t n = b.paths.size();
    const u32 h = b.merkle_height;
    require(h < 32, "restore: tree height");
    require(b.n_trailing_zeros <= 1024, "restore: trailing zeros");
    require(n > 0, "restore: empty batch");
    // every entry of original_order copies a whole opening: bound it by what a query set can be (the largest schedules of
    // WhirConfig::new stay below 2^10 queries) and by the paths it can refer to — no quadratic amplification from a forged proof
    require(b.original_order.size() <= (1u << 12) && n <= b.original_order.size(), "restore: original_order length");
    std::vector<std::vector<u32>> leaves(n);
    for (size_t i = 0; i < n; i++) {
        leaves[i] = b.paths[i].leaf;
        leaves[i].resize(leaves[i].size() + b.n_trailing_zeros, 0);
    }
    auto levels = [&](size_t i) { return i == 0 ? h : lca_level(b.paths[i - 1].leaf_index, b.paths[i].leaf_index); };
    auto skip = [&](size_t i) { return i + 1 < n ? (int)lca_level(b.paths[i].leaf_index, b.paths[i + 1].leaf_index) - 1 : -1; };
    std::vector<std::vector<u32>> subtree(n);  // subtree[i]: 8 words per level 0..levels(i)
    for (size_t i = n; i-- > 0;) {
        const u64 idx = b.paths[i].leaf_index;
        require(idx < (1ull << h), "restore: leaf index outside the tree");
        require(levels(i) <= h, "restore: leaf indices must be strictly increasing");
        size_t stored = 0;
        u32 cur[8];
        hash_slice(leaves[i].data(), leaves[i].size(), cur);
        subtree[i].insert(subtree[i].end(), cur, cur + 8);
        for (unsigned lvl = 0; lvl < levels(i); lvl++) {
            const u32* sib;
            if (skip(i) == (int)lvl) {
                require(subtree[i + 1].size() >= 8 * (size_t)(lvl + 1), "restore: missing subtree hash");
                sib = &subtree[i + 1][8 * lvl];
            } else {
                require(stored + 8 <= b.paths[i].siblings.size(), "restore: missing sibling");
                sib = &b.paths[i].siblings[stored];
                stored += 8;
            }
            u32 nx[8];
            if (((idx >> lvl) & 1) == 0)
                compress_pair(cur, sib, nx);
            else
                compress_pair(sib, cur, nx);
            memcpy(cur, nx, 32);
            subtree[i].insert(subtree[i].end(), cur, cur + 8);
        }
    }
    std::vector<Opening> restored(n);
    for (size_t i = 0; i < n; i++) {
        size_t stored = 0;
        Opening& o = restored[i];
        o.index = b.paths[i].leaf_index;
        o.leaf = leaves[i];
        for (unsigned lvl = 0; lvl < levels(i); lvl++) {
            const u32* sib;
            if (skip(i) == (int)lvl) {
                sib = &subtree[i + 1][8 * lvl];
            } else {
                require(stored + 8 <= b.paths[i].siblings.size(), "restore: missing sibling");
                sib = &b.paths[i].siblings[stored];
                stored += 8;
            }
            o.path.insert(o.path.end(), sib, sib + 8);
        }
        if (i > 0) {
            require(restored[i - 1].path.size() >= 8 * (size_t)levels(i), "restore: previous path too short");
            o.path.insert(o.path.end(), restored[i - 1].path.begin() + 8 * levels(i), restored[i - 1].path.end());
        }
    }
    std::vector<Opening> out;
    for (u32 k : b.original_order) {
        require(k < n, "restore: original_order entry");
        out.push_back(restored[k]);
    }
    return out;
}

// ---- VerifierState (fiat-shamir/src/verifier.rs:15-197) ------------------------------------------------------------------
struct Verifier {
    Challenger ch;
    const std::vector<u32>& transcript;
    size_t off = 0;
    std::vector<Opening> openings;
    size_t opening_idx = 0;
    // VerifierState::raw_transcript (verifier.rs:21,54-60): what was absorbed, in the format the recursion program reads — every
    // absorbed slice zero-padded to the rate, sumcheck polynomials with all their coefficients, a grinding witness as a block of its own
    std::vector<u32> raw;
    void record(const u32* s, size_t n) {
        raw.insert(raw.end(), s, s + n);
        raw.resize((raw.size() + 7) / 8 * 8, 0);
    }
    explicit Verifier(const std::vector<u32>& t) : transcript(t) {}
    const u32* read(size_t n) {
        require(off + n <= transcript.size(), "transcript exhausted");
        const u32* p = transcript.data() + off;
        off += n;
        return p;
    }
    void observe(const u32* s, size_t n) { ch.observe_many(s, n); }
    std::vector<u32> next_base(size_t n) {
        const u32* p = read(n);
        ch.observe_many(p, n);
        record(p, n);
        return std::vector<u32>(p, p + n);
    }
    std::vector<EF> next_ext(size_t n) {
        const u32* p = read(5 * n);
        ch.observe_many(p, 5 * n);
        record(p, 5 * n);
        std::vector<EF> r(n);
        for (size_t i = 0; i < n; i++) r[i] = ef_load(p + 5 * i);
        return r;
    }
    EF next_ext1() { return next_ext(1)[0]; }
    std::vector<EF> sample_vec(size_t n) {  // fiat-shamir/src/utils.rs:43-58
        std::vector<u32> fe;
        require(ch.sample_many((n * 5 + 7) / 8, fe), "stale challenger rate");
        std::vector<EF> r(n);
        for (size_t i = 0; i < n; i++) r[i] = ef_load(&fe[5 * i]);
        return r;
    }
    EF sample() { return sample_vec(1)[0]; }
    std::vector<u64> sample_in_range(u32 bits, size_t n) {  // challenger.rs:66-75
        std::vector<u32> fe;
        require(ch.sample_many((n + 7) / 8, fe), "stale challenger rate");
        std::vector<u64> r(n);
        for (size_t i = 0; i < n; i++) r[i] = (u64)kb::from_monty(fe[i]) & ((1ull << bits) - 1);
        return r;
    }
    void duplex() { ch.duplex(); }
    void check_pow(u32 bits) {  // verifier.rs:145-157
        if (bits == 0) return;
        const u32 w = read(1)[0];
        ch.observe_many(&w, 1);
        require((kb::from_monty(ch.state[8]) & ((1u << bits) - 1)) == 0, "invalid grinding witness");
        record(&w, 1);
    }
    // next_sumcheck_polynomial (verifier.rs:159-195): the proof carries coefficients 1.. ; the constant one follows from the sum
    std::vector<EF> next_sumcheck_poly(size_t n_coeffs, const EF& claimed, const EF* eq_alpha) {
        std::vector<EF> full;
        if (!eq_alpha) {
            const u32* p = read((n_coeffs - 1) * 5);
            EF s = ef_zero();
            std::vector<EF> rest(n_coeffs - 1);
            for (size_t i = 0; i + 1 < n_coeffs; i++) {
                rest[i] = ef_load(p + 5 * i);
                s = ef_add(s, rest[i]);
            }
            const EF c0 = ef_mul_base(ef_sub(claimed, s), to_monty((kb::P + 1) / 2));  // halve
            full.push_back(c0);
            full.insert(full.end(), rest.begin(), rest.end());
        } else {
            const u32* p = read((n_coeffs - 2) * 5);
            EF s = ef_zero();
            std::vector<EF> bare(n_coeffs - 1);
            for (size_t i = 0; i + 2 < n_coeffs; i++) {
                bare[i + 1] = ef_load(p + 5 * i);
                s = ef_add(s, bare[i + 1]);
            }
            bare[0] = ef_sub(claimed, ef_mul(*eq_alpha, s));
            // expand_bare_to_full (fiat-shamir/src/utils.rs:30-41): eq(alpha, X) * bare(X)
            const EF oma = one_minus(*eq_alpha), tam = ef_sub(kb::ef_dbl(*eq_alpha), ef_one());
            const size_t d = bare.size() - 1;
            full.push_back(ef_mul(oma, bare[0]));
            for (size_t k = 1; k <= d; k++) full.push_back(ef_add(ef_mul(oma, bare[k]), ef_mul(tam, bare[k - 1])));
            full.push_back(ef_mul(tam, bare[d]));
        }
        std::vector<u32> flat(full.size() * 5);
        for (size_t i = 0; i < full.size(); i++) memcpy(&flat[5 * i], full[i].v, 20);
        ch.observe_many(flat.data(), flat.size());
        record(flat.data(), flat.size());
        return full;
    }
    const Opening& next_opening() {
        require(opening_idx < openings.size(), "merkle openings exhausted");
        return openings[opening_idx++];
    }
};

// sumcheck_verify (sumcheck/src/verify.rs:5-27): returns the challenges, *target = the claimed final value
std::vector<EF> sumcheck_verify(Verifier& vs, u32 n_vars, u32 degree, EF& target, const EF* eq_alphas) {
    std::vector<EF> challenges;
    for (u32 r = 0; r < n_vars; r++) {
        const std::vector<EF> coeffs = vs.next_sumcheck_poly(degree + 1, target, eq_alphas ? &eq_alphas[r] : nullptr);
        const EF c = vs.sample();
        challenges.push_back(c);
        target = poly_eval(coeffs, c);
    }
    return challenges;
}

// verify_gkr_quotient (quotient_gkr/mod.rs:147-190)
void verify_gkr_quotient(Verifier& vs, u32 n_vars, EF& quotient, std::vector<EF>& point, EF& claim_num, EF& claim_den) {
    require(n_vars > 5, "gkr: too few variables");
    const std::vector<EF> nums = vs.next_ext(32), dens = vs.next_ext(32);
    quotient = ef_zero();
    for (int i = 0; i < 32; i++) {
        require(!kb::ef_is_zero(dens[i]), "gkr: zero denominator");
        quotient = ef_add(quotient, ef_mul(nums[i], kb::ef_inv(dens[i])));
    }
    point = vs.sample_vec(5);
    claim_num = mle_eval(nums[0].v, true, point.data(), 5);
    claim_den = mle_eval(dens[0].v, true, point.data(), 5);
    for (u32 nv = 5; nv < n_vars; nv++) {
        vs.duplex();
        const EF alpha = vs.sample();
        EF target = ef_add(claim_num, ef_mul(alpha, claim_den));
        std::vector<EF> eq_rev(point.rbegin(), point.rend());
        std::vector<EF> ch = sumcheck_verify(vs, nv, 3, target, eq_rev.data());
        std::reverse(ch.begin(), ch.end());
        const std::vector<EF> ie = vs.next_ext(4);
        const EF constraints = ef_add(ef_mul(alpha, ef_mul(ie[2], ie[3])), ef_add(ef_mul(ie[0], ie[3]), ef_mul(ie[1], ie[2])));
        require(kb::ef_eq(target, ef_mul(eq_poly_outside(point.data(), ch.data(), nv), constraints)), "gkr: layer sumcheck does not match the inner evaluations");
        const EF beta = vs.sample(), omb = one_minus(beta);
        claim_num = ef_add(ef_mul(omb, ie[0]), ef_mul(beta, ie[1]));
        claim_den = ef_add(ef_mul(omb, ie[2]), ef_mul(beta, ie[3]));
        ch.push_back(beta);
        point = ch;
    }
}

EF finger_print(u32 domsep, const std::vector<EF>& data, const EF* aeq) {  // utils/src/multilinear.rs:76-84
    EF r = ef_mul_base(aeq[15], to_monty(domsep));
    for (size_t i = 0; i < data.size(); i++) r = ef_add(r, ef_mul(aeq[i], data[i]));
    return r;
}

struct WhirCommitment {  // ParsedCommitment (whir/src/verify.rs:12-60)
    u32 num_variables;
    u32 root[8];
    std::vector<EF> ood_points, ood_answers;
};
WhirCommitment parse_commitment(Verifier& vs, u32 num_variables, u32 ood_samples) {
    WhirCommitment c;
    c.num_variables = num_variables;
    const std::vector<u32> r = vs.next_base(8);
    memcpy(c.root, r.data(), 32);
    if (ood_samples) {
        c.ood_points = vs.sample_vec(ood_samples);
        c.ood_answers = vs.next_ext(ood_samples);
    }
    return c;
}
struct Constraint {  // SparseStatement (whir/src/lib.rs:31-108)
    std::vector<EF> point;
    bool is_next = false;
    std::vector<std::pair<u64, EF>> values;  // (selector, value)
};
u32 fold_at(const lm_whir_config* c, u32 round) { return round == 0 ? c->folding_factor_first : c->folding_factor_subsequent; }
u32 total_fold(const lm_whir_config* c, u32 n_rounds) { return c->folding_factor_first + c->folding_factor_subsequent * n_rounds; }
u32 two_adic_generator(u32 bits) {
    u32 g = to_monty(0x6ac49f88u);  // generator of the 2^24-th roots (koala_bear.rs:50-54)
    for (u32 i = bits; i < 24; i++) g = kb::sqr(g);
    return g;
}

// WhirConfig::verify (whir/src/verify.rs:83-232).  Returns the folding randomness.
std::vector<EF> whir_verify(const lm_whir_config* c, Verifier& vs, const WhirCommitment& commitment, std::vector<Constraint> statement,
                            lm_whir_opening_claim* cap = nullptr) {
    const u32 n = c->num_variables;
    require(c->n_rounds <= LM_MAX_WHIR_ROUNDS && n == total_fold(c, c->n_rounds) + c->final_sumcheck_rounds, "whir: inconsistent configuration");
    require(commitment.num_variables == n, "whir: commitment size");
    std::vector<std::pair<std::vector<EF>, std::vector<Constraint>>> round_constraints;
    std::vector<std::vector<EF>> round_randomness;
    EF claimed = ef_zero();
    auto oods = [](const WhirCommitment& cm) {
        std::vector<Constraint> r;
        for (size_t i = 0; i < cm.ood_points.size(); i++) {
            Constraint k;
            k.point = expand_from_univariate(cm.ood_points[i], cm.num_variables);
            k.values.push_back({0, cm.ood_answers[i]});
            r.push_back(k);
        }
        return r;
    };
    auto combine = [&](const std::vector<Constraint>& cs) {  // combine_constraints :234-252
        const EF gen = vs.sample();
        std::vector<EF> pw;
        EF cur = ef_one();
        for (const Constraint& k : cs)
            for (const auto& v : k.values) {
                claimed = ef_add(claimed, ef_mul(cur, v.second));
                pw.push_back(cur);
                cur = ef_mul(cur, gen);
            }
        return pw;
    };
    auto sumcheck_rounds = [&](u32 rounds, u32 pow_bits) {  // verify_sumcheck_rounds :405-435
        std::vector<EF> rnd;
        for (u32 i = 0; i < rounds; i++) {
            const std::vector<EF> coeffs = vs.next_sumcheck_poly(3, claimed, nullptr);
            vs.check_pow(pow_bits);
            const EF r = vs.sample();
            claimed = poly_eval(coeffs, r);
            rnd.push_back(r);
        }
        return rnd;
    };
    u64 domain_size = 1ull << (n + c->starting_log_inv_rate);
    // verify_stir_challenges :254-317: round_index selects base / EF leaves
    auto stir = [&](u32 query_pow_bits, u32 num_queries, u32 fold, u32 num_variables_after, const WhirCommitment& prev, const std::vector<EF>& fold_rnd,
                    bool base_leaves, u64 dom) {
        vs.check_pow(query_pow_bits);
        const u64 folded = dom >> fold;
        const std::vector<u64> idx = vs.sample_in_range(ilog2(folded), num_queries);
        const u32 gen = two_adic_generator(ilog2(dom) - fold);
        std::vector<Constraint> out;
        for (u64 q : idx) {
            const Opening& o = vs.next_opening();
            const size_t want = (size_t)(base_leaves ? 1 : 5) << fold;
            require(o.leaf.size() == want, "whir: leaf width");
            require(merkle_verify(prev.root, ilog2(folded), q, o.leaf, o.path), "whir: Merkle path does not authenticate");
            Constraint k;
            k.point = expand_from_univariate(ef_from_base(kb::pow(gen, q)), num_variables_after);
            k.values.push_back({0, mle_eval(o.leaf.data(), !base_leaves, fold_rnd.data(), fold)});
            out.push_back(k);
        }
        return out;
    };
    WhirCommitment prev = commitment;
    vs.duplex();
    if (cap) {  // where whir_open of the recursion program starts (recursion.py:470-532)
        memset(cap, 0, sizeof *cap);
        require(commitment.ood_points.size() <= 4 && n <= 32, "whir claim: more OOD samples / variables than lm_whir_opening_claim holds");
        cap->transcript_offset = vs.raw.size();
        memcpy(cap->challenger_state, vs.ch.state, sizeof cap->challenger_state);
        cap->num_variables = n, cap->log_inv_rate = c->starting_log_inv_rate, cap->n_ood = (u32)commitment.ood_points.size();
        memcpy(cap->root, commitment.root, 32);
        for (size_t i = 0; i < commitment.ood_points.size(); i++)
            memcpy(cap->ood_points + 5 * i, commitment.ood_points[i].v, 20), memcpy(cap->ood_answers + 5 * i, commitment.ood_answers[i].v, 20);
    }
    {
        std::vector<Constraint> cs = oods(prev);
        cs.insert(cs.end(), statement.begin(), statement.end());
        std::vector<EF> pw = combine(cs);
        if (cap) {
            const size_t n_ood = commitment.ood_points.size();
            memcpy(cap->combination_gen, (pw.size() > 1 ? pw[1] : ef_zero()).v, 20);
            EF ssum = ef_zero();
            size_t i = n_ood;
            for (const Constraint& k : statement)
                for (const auto& v : k.values) ssum = ef_add(ssum, ef_mul(pw[i++], v.second));
            memcpy(cap->statement_sum, ssum.v, 20);
            cap->n_statement_values = (u32)(pw.size() - n_ood);
        }
        round_constraints.push_back({pw, cs});
    }
    round_randomness.push_back(sumcheck_rounds(fold_at(c, 0), c->starting_folding_pow_bits));
    for (u32 round = 0; round < c->n_rounds; round++) {
        const u32 nv_round = n - total_fold(c, round);
        const WhirCommitment next = parse_commitment(vs, nv_round, c->rounds[round].ood_samples);
        std::vector<Constraint> st = stir(c->rounds[round].query_pow_bits, c->rounds[round].num_queries, fold_at(c, round), nv_round, prev,
                                          round_randomness.back(), round == 0, domain_size);
        std::vector<Constraint> cs = oods(next);
        cs.insert(cs.end(), st.begin(), st.end());
        vs.duplex();
        std::vector<EF> pw = combine(cs);
        round_constraints.push_back({pw, cs});
        round_randomness.push_back(sumcheck_rounds(fold_at(c, round + 1), c->rounds[round].folding_pow_bits));
        prev = next;
        domain_size >>= round == 0 ? c->rs_domain_initial_reduction_factor : 1;
    }
    const u32 nv_final = n - total_fold(c, c->n_rounds);
    const std::vector<EF> final_coeffs = vs.next_ext(1ull << nv_final);
    const std::vector<Constraint> st = stir(c->final_query_pow_bits, c->final_queries, fold_at(c, c->n_rounds), nv_final, prev,
                                            round_randomness.back(), c->n_rounds == 0, domain_size);
    for (const Constraint& k : st) {  // verify_constraint_coeffs :382-397: the final polynomial as a univariate at alpha
        const EF alpha = k.point.empty() ? ef_zero() : k.point[0];
        EF ev = ef_zero();
        for (size_t i = final_coeffs.size(); i-- > 0;) ev = ef_add(ef_mul(ev, alpha), final_coeffs[i]);
        require(kb::ef_eq(ev, k.values[0].second), "whir: final polynomial disagrees with a queried fold");
    }
    const std::vector<EF> final_rnd = sumcheck_rounds(c->final_sumcheck_rounds, 0);
    round_randomness.push_back(final_rnd);
    std::vector<EF> folding;
    for (const auto& r : round_randomness) folding.insert(folding.end(), r.begin(), r.end());
    // eval_constraints_poly :339-379
    EF weights = ef_zero(), stmt_weights = ef_zero();
    {
        std::vector<EF> point = folding;
        for (size_t round = 0; round < round_constraints.size(); round++) {
            if (round > 0) point.erase(point.begin(), point.begin() + fold_at(c, (u32)round - 1));
            const std::vector<EF>& rnd = round_constraints[round].first;
            size_t i = 0;
            for (const Constraint& k : round_constraints[round].second) {
                const size_t inner = k.point.size(), sel_vars = point.size() - inner;
                require(inner <= point.size(), "whir: statement larger than the polynomial");
                const EF* ip = point.data() + sel_vars;
                const EF common = k.is_next ? next_mle(k.point.data(), ip, (u32)inner) : eq_poly_outside(k.point.data(), ip, inner);
                for (const auto& v : k.values) {
                    EF e = common;
                    for (size_t j = 0; j < sel_vars; j++)
                        e = ef_mul(e, (v.first & (1ull << (sel_vars - 1 - j))) ? point[j] : one_minus(point[j]));
                    if (round == 0 && i >= commitment.ood_points.size()) stmt_weights = ef_add(stmt_weights, ef_mul(e, rnd[i]));
                    weights = ef_add(weights, ef_mul(e, rnd[i++]));
                }
            }
            require(i == rnd.size(), "whir: combination randomness count");
        }
    }
    std::vector<EF> rev(final_rnd.rbegin(), final_rnd.rend());
    const EF final_value = eval_multilinear_coeffs(final_coeffs.data(), final_coeffs.size(), rev.data());
    require(kb::ef_eq(claimed, ef_mul(weights, final_value)), "whir: final sumcheck value does not match the constraints");
    if (cap) {
        memcpy(cap->statement_weights, stmt_weights.v, 20);
        for (size_t i = 0; i < folding.size(); i++) memcpy(cap->folding_randomness + 5 * i, folding[i].v, 20);
    }
    return folding;
}

// AIR constraint polynomial of table t at the column evaluations (SumcheckComputation::eval_extension)
EF air_eval(int t, const std::vector<EF>& ce, const air::Extra& x) {
    if (t == air::T_EXECUTION) return air::eval_execution<EF>(ce.data(), ce.data() + 20, x);
    if (t == air::T_EXTENSION_OP) return air::eval_extension_op<EF>(ce.data(), ce.data() + 29, x);
    // Poseidon16: the device kernels read the affine forms of the partial block and the challenge-weighted output blocks as
    // virtual columns (air_tables.h: POS_VIRT_*); at a point they are the same linear forms of the column evaluations
    std::vector<EF> col(ce.begin(), ce.begin() + 109);
    col.resize(109 + air::POS_N_VIRT, ef_zero());
    const EF* u = &ce[41];  // beginning_full_rounds[1] (16) then partial_rounds (20)
    for (int r = 0; r < 20; r++) {
        EF s = ef_from_base(air::kPoseidonLinear.y[r][36]);
        for (int j = 0; j < 16 + r; j++) s = ef_add(s, ef_mul_base(u[j], air::kPoseidonLinear.y[r][j]));
        col[air::POS_VIRT_Y + r] = s;
    }
    for (int i = 0; i < 16; i++) {
        EF s = ef_from_base(air::kPoseidonLinear.fin[i][36]);
        for (int j = 0; j < 36; j++) s = ef_add(s, ef_mul_base(u[j], air::kPoseidonLinear.fin[i][j]));
        col[air::POS_VIRT_E + i] = s;
    }
    for (int sgm = 0; sgm < 3; sgm++)
        for (int k = 0; k < 5; k++) {  // plane k of V_s = sum_i alpha^(k0 + i) out_i
            EF s = ef_zero();
            for (int i = 0; i < 16; i++) {
                const EF& o = sgm == 1 ? u[i] : ce[air::POS_OUT_COL[sgm] + i];
                s = ef_add(s, ef_mul_base(o, x.alpha_powers[air::POS_OUT_K0[sgm] + i].v[k]));
            }
            col[air::POS_VIRT_O + 5 * sgm + k] = s;
        }
    auto cf = [&](int c) { return col[c]; };
    EF r = air::eval_poseidon16_segment<EF, 0>(cf, x);
    r = ef_add(r, air::eval_poseidon16_segment<EF, 1>(cf, x));
    r = ef_add(r, air::eval_poseidon16_segment<EF, 2>(cf, x));
    r = ef_add(r, air::eval_poseidon16_segment<EF, 3>(cf, x));
    r = ef_add(r, air::eval_poseidon16_segment<EF, 4>(cf, x));
    return r;
}

void verify_execution(const lm_verify_instance* in, const std::vector<u32>& transcript, const std::vector<PrunedBatch>& batches,
                      const lm_whir_builder* builder_override, lmh_raw_proof* raw = nullptr) {
    Verifier vs(transcript);
    for (const PrunedBatch& b : batches) {  // VerifierState::new (verifier.rs:28-44)
        std::vector<Opening> r = restore(b);
        vs.openings.insert(vs.openings.end(), r.begin(), r.end());
    }
    vs.observe(in->public_input, in->n_public_input);
    {
        u32 st[16];
        memcpy(st, in->bytecode_hash, 32);
        for (int i = 0; i < 8; i++) st[8 + i] = to_monty(kSnarkDomainSep[i]);
        lmh::host_compress(st);
        vs.observe(st, 8);
        if (raw) memcpy(raw->stmt.bytecode_hash_domsep, st, 32);
    }
    const std::vector<u32> dims_m = vs.next_base(6);
    u32 dims[6];
    for (int i = 0; i < 6; i++) dims[i] = kb::from_monty(dims_m[i]);
    const u32 log_inv_rate = dims[0], log_mem = dims[1], log_bc = in->log_bytecode;
    require(dims[2] == in->n_public_input, "public input length differs from the proof's");
    const u32 log_rows[3] = {dims[3], dims[4], dims[5]};
    require(rate_ok(log_inv_rate), "invalid rate");
    for (int t = 0; t < 3; t++)
        require(log_rows[t] >= MIN_LOG_N_ROWS_PER_TABLE && log_rows[t] <= max_log_n_rows_per_table(t), "table height out of range");
    const u32 max_rows = std::max(log_rows[0], std::max(log_rows[1], log_rows[2]));
    require(log_mem >= std::max(max_rows, log_bc), "memory smaller than a table or the bytecode");
    require(log_mem >= MIN_LOG_MEMORY_SIZE && log_mem <= MAX_LOG_MEMORY_SIZE, "memory size out of range");
    require(log_bc >= MIN_BYTECODE_LOG_SIZE, "bytecode too small");
    require(log_rows[0] >= max_rows, "the execution table must be the tallest (stacked_pcs.rs:111)");
    int order[3];
    sorted_tables(log_rows, order);
    // stacked_pcs_parse_commitment (stacked_pcs.rs:159-181)
    u64 total = (2ull << log_mem) + (1ull << std::max(log_bc, max_rows));
    for (int t = 0; t < 3; t++) total += (u64)kVmTables[t].n_columns << log_rows[t];
    const u32 stacked_n_vars = log2_ceil_u64(total);
    lm_whir_builder builder;
    if (builder_override)
        builder = *builder_override;
    else
        lmh_default_whir_builder(log_inv_rate, 0, &builder);
    require(builder.starting_log_inv_rate == log_inv_rate, "builder rate differs from the proof's");
    lm_whir_config cfg;
    require(lmh_whir_config_new(&builder, stacked_n_vars, &cfg) == LM_OK, "WhirConfig::new rejected the parameters");
    const WhirCommitment commitment = parse_commitment(vs, stacked_n_vars, cfg.commitment_ood_samples);

    const EF logup_c = vs.sample();
    vs.duplex();
    const std::vector<EF> alphas = vs.sample_vec(4);
    EF aeq[16];  // eval_eq(&logup_alphas)
    for (u32 i = 0; i < 16; i++) {
        EF a = ef_one();
        for (u32 j = 0; j < 4; j++) a = ef_mul(a, ((i >> (3 - j)) & 1) ? alphas[j] : one_minus(alphas[j]));
        aeq[i] = a;
    }
    // ---- verify_generic_logup (logup.rs:326-493) ----
    u64 active = (1ull << log_mem) + std::max(1ull << log_bc, 1ull << max_rows) + (1ull << log_rows[0]);
    for (int t = 0; t < 3; t++) {
        u32 cols = 1;
        for (u32 l = 0; l < kVmTables[t].n_lookups; l++) cols += kVmTables[t].lookups[l].n_values;
        active += (u64)cols << log_rows[t];
    }
    const u32 gkr_n_vars = log2_ceil_u64(active);
    EF quotient, num_value, den_value;
    std::vector<EF> gp;
    verify_gkr_quotient(vs, gkr_n_vars, quotient, gp, num_value, den_value);
    require(kb::ef_is_zero(quotient), "logup sum != 0");
    EF r_num = ef_zero(), r_den = ef_zero();
    auto from_end = [&](u32 k) { return gp.data() + (gkr_n_vars - k); };
    auto pref_at = [&](u64 offset, u32 log_height) {
        const u32 miss = gkr_n_vars - log_height;
        EF acc = ef_one();
        for (u32 j = 0; j < miss; j++) acc = ef_mul(acc, (((offset >> log_height) >> (miss - 1 - j)) & 1) ? gp[j] : one_minus(gp[j]));
        return acc;
    };
    lm_pcs_statement_claim scl;
    memset(&scl, 0, sizeof scl);
    EF pref = pref_at(0, log_mem);
    scl.off_value_memory_acc = vs.raw.size();
    const EF value_memory_acc = vs.next_ext1();
    r_num = ef_sub(r_num, ef_mul(pref, value_memory_acc));
    scl.off_value_memory = vs.raw.size();
    const EF value_memory = vs.next_ext1();
    r_den = ef_add(r_den, ef_mul(pref, ef_sub(logup_c, finger_print(0, {value_memory, mle_of_01234567_etc(from_end(log_mem), log_mem)}, aeq))));
    u64 offset = 1ull << log_mem;
    const u32 log_bc_padded = std::max(log_bc, max_rows);
    pref = pref_at(offset, log_bc);
    const EF pref_padded = pref_at(offset, log_bc_padded);
    scl.off_value_bytecode_acc = vs.raw.size();
    const EF value_bytecode_acc = vs.next_ext1();
    r_num = ef_sub(r_num, ef_mul(pref, value_bytecode_acc));
    {
        const EF index_value = mle_of_01234567_etc(from_end(log_bc), log_bc);
        std::vector<EF> bp(from_end(log_bc), from_end(log_bc) + log_bc);
        bp.insert(bp.end(), alphas.begin(), alphas.end());  // from_end(alphas, log2_ceil(12) = 4): all four
        const EF bytecode_value = mle_eval_base_big(in->bytecode, bp.data(), log_bc + 4);
        memcpy(scl.bytecode_value, bytecode_value.v, 20);
        // (alphas[..len - 4] is empty: the corrective product is 1)
        const EF d = ef_add(ef_add(bytecode_value, ef_mul(index_value, aeq[12])), ef_mul_base(aeq[15], to_monty(2)));
        r_den = ef_add(r_den, ef_mul(pref, ef_sub(logup_c, d)));
        r_den = ef_add(r_den, ef_mul(pref_padded, mle_of_zeros_then_ones(1ull << log_bc, from_end(log_bc_padded), log_bc_padded)));
    }
    offset += 1ull << log_bc_padded;
    std::vector<ColVal> columns_values[3];
    EF bus_num[3], bus_den[3];
    for (int k = 0; k < 3; k++) {
        const int t = order[k];
        const VmTableDef& def = kVmTables[t];
        const u32 lr = log_rows[t];
        auto note = [&](u32 col, u64 off) {  // where the statement's value of column `col` of table t lies in the raw transcript
            require(scl.n_logup_values[t] < 40, "statement claim: more logup values than lm_pcs_statement_claim holds");
            scl.logup_col[t][scl.n_logup_values[t]] = col, scl.logup_off[t][scl.n_logup_values[t]++] = off;
        };
        if (t == 0) {
            note(0, vs.raw.size());
            const EF on_pc = vs.next_ext1();
            columns_values[t].push_back({0, on_pc});
            const u64 instr_off = vs.raw.size();
            std::vector<EF> instr = vs.next_ext(12);
            for (u32 i = 0; i < 12; i++) columns_values[t].push_back({8 + i, instr[i]}), note(8 + i, instr_off + 5 * i);
            pref = pref_at(offset, lr);
            r_num = ef_add(r_num, pref);
            instr.push_back(on_pc);
            r_den = ef_add(r_den, ef_mul(pref, ef_sub(logup_c, finger_print(2, instr, aeq))));
            offset += 1ull << lr;
        }
        scl.off_bus_selector[t] = vs.raw.size();
        const EF on_selector = vs.next_ext1();
        pref = pref_at(offset, lr);
        r_num = ef_add(r_num, ef_mul(pref, on_selector));
        scl.off_bus_data[t] = vs.raw.size();
        const EF on_data = vs.next_ext1();
        r_den = ef_add(r_den, ef_mul(pref, on_data));
        bus_num[t] = on_selector;
        bus_den[t] = on_data;
        offset += 1ull << lr;
        for (u32 l = 0; l < def.n_lookups; l++) {
            note(def.lookups[l].index, vs.raw.size());
            const EF index_eval = vs.next_ext1();
            columns_values[t].push_back({def.lookups[l].index, index_eval});
            for (u32 i = 0; i < def.lookups[l].n_values; i++) {
                note(def.lookups[l].first_value + i, vs.raw.size());
                const EF value_eval = vs.next_ext1();
                columns_values[t].push_back({def.lookups[l].first_value + i, value_eval});
                pref = pref_at(offset, lr);
                r_num = ef_add(r_num, pref);
                r_den = ef_add(r_den, ef_mul(pref, ef_sub(logup_c, finger_print(0, {value_eval, kb::ef_add_base(index_eval, to_monty(i))}, aeq))));
                offset += 1ull << lr;
            }
        }
    }
    r_den = ef_add(r_den, mle_of_zeros_then_ones(offset, gp.data(), gkr_n_vars));
    require(kb::ef_eq(r_num, num_value), "logup: numerator claim does not match the column evaluations");
    require(kb::ef_eq(r_den, den_value), "logup: denominator claim does not match the column evaluations");

    // ---- AIR (verify_execution.rs:100-186) ----
    scl.air_offset = vs.raw.size();
    memcpy(scl.air_challenger_state, vs.ch.state, sizeof scl.air_challenger_state);
    memcpy(scl.logup_c, logup_c.v, 20);
    const EF bus_beta = vs.sample();
    vs.duplex();
    const EF air_alpha = vs.sample();
    vs.duplex();
    const EF eta = vs.sample();
    air::Extra x;
    {
        EF p = ef_one();
        for (int i = 0; i < air::MAX_ALPHA; i++) {
            x.alpha_powers[i] = p;
            p = ef_mul(p, air_alpha);
        }
        static const u32 MDS_COL[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
        for (int sgm = 0; sgm < 3; sgm++)
            for (int j = 0; j < 16; j++) {
                EF b = ef_zero();
                for (int i = 0; i < 16; i++)
                    b = ef_add(b, ef_mul_base(x.alpha_powers[air::POS_OUT_K0[sgm] + i], to_monty(MDS_COL[(16 + i - j) & 15])));
                x.out_beta[sgm][j] = b;
            }
        memcpy(x.logup_eq, aeq, sizeof aeq);
        x.bus_beta = bus_beta;
    }
    EF initial_sum = ef_zero(), eta_power = ef_one(), eta_powers[3];
    u32 max_full_degree = 0;
    for (int k = 0; k < 3; k++) {
        const int t = order[k];
        const EF dir = kVmTables[t].pull ? kb::ef_neg(ef_one()) : ef_one();
        const EF bfv = ef_add(ef_mul(bus_num[t], dir), ef_mul(bus_beta, ef_sub(bus_den[t], logup_c)));
        initial_sum = ef_add(initial_sum, ef_mul(eta_power, bfv));
        eta_powers[k] = eta_power;
        eta_power = ef_mul(eta_power, eta);
        max_full_degree = std::max(max_full_degree, (u32)air::degree(t) + 1);
    }
    const u32 n_max = log_rows[order[0]];
    EF claimed_air = initial_sum;
    const std::vector<EF> air_point = sumcheck_verify(vs, n_max, max_full_degree, claimed_air, nullptr);
    EF my_air = ef_zero();
    std::vector<u32> col_evals_flat;
    for (int k = 0; k < 3; k++) {
        const int t = order[k];
        const u32 nt = log_rows[t], n_tot = kVmTables[t].n_columns + kVmTables[t].n_shift;
        scl.off_inner_evals[t] = vs.raw.size();
        const std::vector<EF> ce = vs.next_ext(n_tot);
        for (const EF& e : ce) col_evals_flat.insert(col_evals_flat.end(), e.v, e.v + 5);
        const EF constraint_eval = air_eval(t, ce, x);
        memcpy(scl.air_constraint_evals[t], constraint_eval.v, 20);
        // back_loaded_table_contribution (verify_execution.rs:233-251)
        std::vector<EF> nat(nt);
        for (u32 j = 0; j < nt; j++) nat[j] = air_point[n_max - 1 - j];
        EF kt = ef_one();
        for (u32 j = 0; j < n_max - nt; j++) kt = ef_mul(kt, air_point[j]);
        const EF eqv = eq_poly_outside(from_end(nt), nat.data(), nt);
        my_air = ef_add(my_air, ef_mul(ef_mul(eta_powers[k], kt), ef_mul(eqv, constraint_eval)));
    }
    require(kb::ef_eq(my_air, claimed_air), "AIR: the constraint polynomials do not vanish on the claimed column evaluations");

    // ---- public memory + statements (:188-223) ----
    u32 lpm = 0;
    while ((1ull << lpm) < in->n_public_input) lpm++;
    const std::vector<EF> pm_point = vs.sample_vec(lpm);
    std::vector<u32> public_memory((size_t)1 << lpm, 0);
    memcpy(public_memory.data(), in->public_input, (size_t)in->n_public_input * 4);
    const EF pm_eval = mle_eval(public_memory.data(), false, pm_point.data(), lpm);
    std::vector<u32> gp_flat, ap_flat, pm_flat;
    for (const EF& e : gp) gp_flat.insert(gp_flat.end(), e.v, e.v + 5);
    for (const EF& e : air_point) ap_flat.insert(ap_flat.end(), e.v, e.v + 5);
    for (const EF& e : pm_point) pm_flat.insert(pm_flat.end(), e.v, e.v + 5);
    Statements S;
    assemble_statements(S, log_rows, log_mem, log_bc, in->ending_pc, gp_flat.data(), gkr_n_vars, value_memory, value_memory_acc,
                        value_bytecode_acc, pm_flat.data(), lpm, pm_eval, columns_values, ap_flat.data(), col_evals_flat.data());
    std::vector<Constraint> statement;
    for (const lm_sparse_statement& s : S.sts) {
        Constraint k;
        k.is_next = s.is_next != 0;
        for (u32 j = 0; j < s.point_len; j++) k.point.push_back(ef_load(&S.pts[(s.point_offset + j) * 5]));
        for (u32 v = 0; v < s.n_values; v++) k.values.push_back({S.sels[s.values_offset + v], ef_load(&S.vals[(s.values_offset + v) * 5])});
        statement.push_back(k);
    }
    if (raw) {
        require(gkr_n_vars <= 32 && n_max <= 32 && lpm <= 8, "statement claim: a point longer than lm_pcs_statement_claim holds");
        for (int t = 0; t < 3; t++) scl.log_rows[t] = log_rows[t], scl.table_order[t] = (u32)order[t];
        scl.log_memory = log_mem, scl.log_bytecode = log_bc, scl.gkr_n_vars = gkr_n_vars, scl.n_max = n_max, scl.ending_pc = in->ending_pc, scl.log_public_memory = lpm, scl.air_degree = max_full_degree;
        memcpy(scl.gkr_point, gp_flat.data(), gp_flat.size() * 4);
        memcpy(scl.air_point, ap_flat.data(), ap_flat.size() * 4);
        memcpy(scl.pm_point, pm_flat.data(), pm_flat.size() * 4);
        memcpy(scl.bytecode_hash_domsep, raw->stmt.bytecode_hash_domsep, 32);
        raw->stmt = scl;
    }
    whir_verify(&cfg, vs, commitment, statement, raw ? &raw->claim : nullptr);
    require(vs.off == transcript.size(), "trailing transcript words");
    require(vs.opening_idx == vs.openings.size(), "unused Merkle openings");
    if (raw) raw->transcript = std::move(vs.raw);
}

int run(const lm_verify_instance* in, const std::vector<u32>& transcript, const std::vector<PrunedBatch>& batches, const lm_whir_builder* b,
        lmh_raw_proof* raw = nullptr) {
    if (!in || !in->bytecode || !in->bytecode_hash || (in->n_public_input && !in->public_input)) {
        lm_set_error("lmh_verify_execution: missing instance data");
        return LM_E_INVALID;
    }
    try {
        verify_execution(in, transcript, batches, b, raw);
    } catch (const Fail& f) {
        lm_set_error("verify_execution: %s", f.why.c_str());
        return LM_E_INVALID;
    } catch (const std::exception& e) {
        lm_set_error("verify_execution: %s", e.what());
        return LM_E_INVALID;
    }
    return LM_OK;
}
}  // namespace

extern "C" {

int lmh_verify_execution(const lm_verify_instance* instance, const lmh_proof* proof, const lm_whir_builder* builder) {
    if (!proof) return LM_E_INVALID;
    return run(instance, lmh::proof_transcript(proof), lmh::proof_batches(proof), builder);
}
int lmh_verify_execution_bytes(const lm_verify_instance* instance, const uint8_t* bytes, uint64_t n, int compressed, const lm_whir_builder* builder) {
    lmh_proof* p = compressed ? lmh_proof_decompress(bytes, n) : lmh_proof_from_postcard(bytes, n);
    if (!p) return LM_E_INVALID;
    const int rc = lmh_verify_execution(instance, p, builder);
    lmh_proof_free(p);
    return rc;
}
int lmh_verify_execution_prover(const lm_verify_instance* instance, const lmh_prover* p, const lm_whir_builder* builder) {
    if (!p) return LM_E_INVALID;
    return run(instance, p->transcript, lmh::prune(p), builder);
}
int lmh_verify_execution_raw(const lm_verify_instance* instance, const lmh_prover* p, const lm_whir_builder* builder, lmh_raw_proof** out) {
    if (!p || !out) return LM_E_INVALID;
    lmh_raw_proof* r = new lmh_raw_proof();
    const int rc = run(instance, p->transcript, lmh::prune(p), builder, r);
    if (rc != LM_OK) {
        delete r;
        r = nullptr;
    }
    *out = r;
    return rc;
}
const uint32_t* lmh_raw_proof_transcript(const lmh_raw_proof* r, uint64_t* n_words) {
    if (n_words) *n_words = r ? r->transcript.size() : 0;
    return r ? r->transcript.data() : nullptr;
}
const lm_whir_opening_claim* lmh_raw_proof_whir_claim(const lmh_raw_proof* r) { return r ? &r->claim : nullptr; }
const lm_pcs_statement_claim* lmh_raw_proof_statement_claim(const lmh_raw_proof* r) { return r ? &r->stmt : nullptr; }
void lmh_raw_proof_free(lmh_raw_proof* r) { delete r; }

}  // extern "C"
