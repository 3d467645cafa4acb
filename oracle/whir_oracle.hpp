// ORACLE — TEST INFRASTRUCTURE ONLY (see kb_oracle.hpp header).
//
// Scalar restatement of the WHIR PCS as used by leanMultisig: parameter derivation (crates/whir/src/config.rs),
// Fiat–Shamir prover/verifier state (crates/backend/fiat-shamir/src/{prover,verifier,utils}.rs), commit
// (crates/whir/src/commit.rs), open (crates/whir/src/open.rs), product sumcheck
// (crates/backend/sumcheck/src/product_computation.rs) and verify (crates/whir/src/verify.rs).
//
// Determinism: the reference picks the PoW witness with rayon `find_any`; here — and in the device
// path — the witness is the SMALLEST valid one, which every verifier accepts.
// Wire format: Merkle hints are kept un-pruned (RawProof-like, fiat-shamir/src/transcript.rs:20-31); pruning
// (merkle_pruning.rs) is wire-format work ranked "next" in SURVEY.md §8(f).
#pragma once
#include <cmath>
#include <stdexcept>
#include "kb_oracle.hpp"

namespace orc {

// ---------------------------------------------------------------------------------------------
// config.rs:7-80
// ---------------------------------------------------------------------------------------------
struct FoldingFactor {
    size_t first_round, subsequent_round;
    size_t at_round(size_t r) const { return r == 0 ? first_round : subsequent_round; }
    size_t total_number(size_t n_rounds) const { return first_round + subsequent_round * n_rounds; }
    // config.rs:53-73
    void compute_number_of_rounds(size_t num_variables, size_t max_send, size_t& n_rounds, size_t& final_sc) const {
        size_t nv = num_variables - first_round;
        if (nv < max_send) {
            n_rounds = 0;
            final_sc = nv;
            return;
        }
        n_rounds = (nv - max_send + subsequent_round - 1) / subsequent_round;
        final_sc = nv - n_rounds * subsequent_round;
    }
};

enum SecurityAssumption { UniqueDecoding = 0, JohnsonBound = 1, CapacityBound = 2 };

// config.rs:444-617 (f64 maths, restated verbatim; SURVEY.md F11: the integers it yields are inputs of the device path)
struct Soundness {
    SecurityAssumption t;
    double log_eta(size_t log_inv_rate, double log_c) const {
        if (t == JohnsonBound) return -(0.5 * (double)log_inv_rate + log_c);
        if (t == CapacityBound) return -((double)log_inv_rate + log_c);
        throw std::runtime_error("log_eta: UD");
    }
    double list_size_bits(size_t log_degree, size_t log_inv_rate, double log_c) const {
        if (t == UniqueDecoding) return 0.;
        double le = log_eta(log_inv_rate, log_c);
        if (t == JohnsonBound) return (double)log_inv_rate / 2. - (1. + le);
        return (double)(log_degree + log_inv_rate) - le;
    }
    double prox_gaps_error(size_t log_degree, size_t log_inv_rate, size_t field_size_bits, size_t num_functions,
                           double log_c) const {
        double error;
        if (t == UniqueDecoding) {
            error = (double)(log_degree + log_inv_rate);
        } else {
            double le = log_eta(log_inv_rate, log_c);
            if (t == JohnsonBound) {
                double eta = std::pow(2.0, le);
                double rho = 1. / (double)(1u << log_inv_rate);
                double rho_sqrt = std::sqrt(rho);
                double gamma = 1. - rho_sqrt - eta;
                double n = (double)((size_t)1 << (log_degree + log_inv_rate));
                double m = std::fmax(std::ceil(rho_sqrt / (2. * eta)), 3.);
                double mh = m + 0.5;
                double num_1 = (2. * (mh * mh * mh * mh * mh) + 3. * mh * gamma * rho) * n;
                double den_1 = 3. * rho * rho_sqrt;
                double num_2 = mh;
                double den_2 = rho_sqrt;
                error = std::log2((num_1 / den_1) + (num_2 / den_2));
            } else {
                error = (double)(log_degree + 2 * log_inv_rate) - le;
            }
        }
        double nf = std::log2((double)num_functions - 1.);
        return (double)field_size_bits - (error + nf);
    }
    double log_1_delta(size_t log_inv_rate, double log_c) const {
        double eta = t == UniqueDecoding ? 0. : std::pow(2.0, log_eta(log_inv_rate, log_c));
        double rate = 1. / (double)(1u << log_inv_rate);
        double delta = t == UniqueDecoding ? 0.5 * (1. - rate)
                       : t == JohnsonBound ? 1. - std::sqrt(rate) - eta
                                           : 1. - rate - eta;
        return std::log2(1. - delta);
    }
    size_t queries(size_t level, size_t log_inv_rate, double log_c) const {
        return (size_t)std::ceil(-(double)level / log_1_delta(log_inv_rate, log_c));
    }
    double queries_error(size_t log_inv_rate, size_t num_queries, double log_c) const {
        return -(double)num_queries * log_1_delta(log_inv_rate, log_c);
    }
    double ood_error(size_t log_degree, size_t log_inv_rate, size_t field_size_bits, size_t ood_samples,
                     double log_c) const {
        if (t == UniqueDecoding) return 0.;
        double ls = list_size_bits(log_degree, log_inv_rate, log_c);
        double error = 2. * ls + (double)(log_degree * ood_samples);
        return (double)(ood_samples * field_size_bits) + 1. - error;
    }
    size_t determine_ood_samples(size_t security_level, size_t log_degree, size_t log_inv_rate, size_t field_size_bits,
                                 double log_c) const {
        if (t == UniqueDecoding) return 0;
        for (size_t s = 1; s < 64; s++)
            if (ood_error(log_degree, log_inv_rate, field_size_bits, s, log_c) >= (double)security_level) return s;
        throw std::runtime_error("no ood samples");
    }
};

struct WhirConfigBuilder {
    size_t starting_log_inv_rate = 1;
    size_t max_num_variables_to_send_coeffs = 8;
    size_t rs_domain_initial_reduction_factor = 5;
    FoldingFactor folding_factor{7, 5};
    SecurityAssumption soundness_type = JohnsonBound;
    size_t security_level = 124;
    size_t pow_bits = 16;
};

struct RoundConfig {
    size_t query_pow_bits, folding_pow_bits, num_queries, ood_samples, log_inv_rate, num_variables, folding_factor;
    size_t domain_size;
    uint32_t folded_domain_gen;
};

struct WhirConfig {
    size_t num_variables = 0;
    size_t commitment_ood_samples = 0, starting_log_inv_rate = 0, starting_folding_pow_bits = 0;
    FoldingFactor folding_factor{0, 0};
    size_t rs_domain_initial_reduction_factor = 0;
    std::vector<RoundConfig> round_parameters;
    size_t final_queries = 0, final_query_pow_bits = 0, final_log_inv_rate = 0, final_sumcheck_rounds = 0;

    size_t n_rounds() const { return round_parameters.size(); }
    size_t rs_reduction_factor(size_t round) const { return round == 0 ? rs_domain_initial_reduction_factor : 1; }
    size_t starting_domain_size() const { return (size_t)1 << (num_variables + starting_log_inv_rate); }
    size_t n_vars_of_final_polynomial() const { return num_variables - folding_factor.total_number(n_rounds()); }

    static double fold_sumcheck(Soundness s, size_t fsb, size_t nv, size_t lir, double log_c) {
        return (double)fsb - (s.list_size_bits(nv, lir, log_c) + 1.);
    }
    static double folding_pow_bits(size_t level, Soundness s, size_t fsb, size_t nv, size_t lir, double log_c) {
        double e = std::fmin(s.prox_gaps_error(nv, lir, fsb, 2, log_c), fold_sumcheck(s, fsb, nv, lir, log_c));
        return std::fmax(0., (double)level - e);
    }
    static double queries_combination(Soundness s, size_t fsb, size_t nv, size_t lir, size_t ood, size_t nq,
                                      double log_c) {
        double ls = s.list_size_bits(nv, lir, log_c);
        return (double)fsb - (std::log2((double)(ood + nq)) + ls + 1.);
    }
    // config.rs:146-183
    static double optimal_log_c(const WhirConfigBuilder& b, size_t fsb, size_t nv, size_t lir) {
        if (b.soundness_type == UniqueDecoding) return 0.0;
        Soundness s{b.soundness_type};
        size_t qsl = b.security_level > b.pow_bits ? b.security_level - b.pow_bits : 0;
        size_t best_m = 3, best_q = (size_t)-1;
        for (size_t m = 3; m <= 100; m++) {
            double log_c = std::log2(2.0 * (double)m);
            double fp = folding_pow_bits(b.security_level, s, fsb, nv, lir, log_c);
            if ((size_t)std::ceil(fp) > b.pow_bits) break;
            size_t q = s.queries(qsl, lir, log_c);
            if (q < best_q) {
                best_q = q;
                best_m = m;
            }
        }
        return std::log2(2.0 * (double)best_m);
    }
    // config.rs:186-334
    static WhirConfig make(const WhirConfigBuilder& b, size_t num_variables) {
        Soundness s{b.soundness_type};
        const size_t fsb = 155;  // EF::bits()
        size_t qsl = b.security_level > b.pow_bits ? b.security_level - b.pow_bits : 0;
        size_t log_inv_rate = b.starting_log_inv_rate;
        size_t domain_size = (size_t)1 << (num_variables + log_inv_rate);
        size_t num_rounds, final_sc;
        b.folding_factor.compute_number_of_rounds(num_variables, b.max_num_variables_to_send_coeffs, num_rounds,
                                                  final_sc);
        double log_c_old = optimal_log_c(b, fsb, num_variables, log_inv_rate);
        WhirConfig c;
        c.num_variables = num_variables;
        c.commitment_ood_samples = s.determine_ood_samples(b.security_level, num_variables, log_inv_rate, fsb, log_c_old);
        c.starting_log_inv_rate = b.starting_log_inv_rate;
        c.starting_folding_pow_bits =
            (size_t)std::ceil(folding_pow_bits(b.security_level, s, fsb, num_variables, log_inv_rate, log_c_old));
        c.folding_factor = b.folding_factor;
        c.rs_domain_initial_reduction_factor = b.rs_domain_initial_reduction_factor;
        size_t nvm = num_variables - b.folding_factor.at_round(0);
        for (size_t round = 0; round < num_rounds; round++) {
            size_t rsr = round == 0 ? b.rs_domain_initial_reduction_factor : 1;
            size_t next_rate = log_inv_rate + (b.folding_factor.at_round(round) - rsr);
            double log_c_new = optimal_log_c(b, fsb, nvm, next_rate);
            size_t num_queries = s.queries(qsl, log_inv_rate, log_c_old);
            size_t ood = s.determine_ood_samples(b.security_level, nvm, next_rate, fsb, log_c_new);
            double query_error = s.queries_error(log_inv_rate, num_queries, log_c_old);
            double comb = queries_combination(s, fsb, nvm, next_rate, ood, num_queries, log_c_new);
            double qpb = std::fmax(0., (double)b.security_level - std::fmin(query_error, comb));
            double fpb = folding_pow_bits(b.security_level, s, fsb, nvm, next_rate, log_c_new);
            size_t ff = b.folding_factor.at_round(round);
            size_t nff = b.folding_factor.at_round(round + 1);
            unsigned log_dom = 0;
            while (((size_t)1 << (log_dom + 1)) <= domain_size) log_dom++;
            RoundConfig rc{(size_t)std::ceil(qpb), (size_t)std::ceil(fpb), num_queries, ood, log_inv_rate, nvm, ff,
                           domain_size, two_adic_generator(log_dom - (unsigned)ff)};
            c.round_parameters.push_back(rc);
            nvm -= nff;
            log_inv_rate = next_rate;
            domain_size >>= rsr;
            log_c_old = log_c_new;
        }
        c.final_queries = s.queries(qsl, log_inv_rate, log_c_old);
        c.final_query_pow_bits = (size_t)std::ceil(
            std::fmax(0., (double)b.security_level - s.queries_error(log_inv_rate, c.final_queries, log_c_old)));
        c.final_sumcheck_rounds = final_sc;
        c.final_log_inv_rate = log_inv_rate;
        return c;
    }
    // config.rs:423-443
    RoundConfig final_round_config() const {
        if (round_parameters.empty()) throw std::runtime_error("final_round_config: no rounds (config.rs:424 asserts)");
        const RoundConfig& last = round_parameters.back();
        size_t rsr = rs_reduction_factor(n_rounds() - 1);
        size_t ff = folding_factor.at_round(n_rounds());
        size_t domain_size = last.domain_size >> rsr;
        unsigned log_dom = 0;
        while (((size_t)1 << (log_dom + 1)) <= domain_size) log_dom++;
        return RoundConfig{final_query_pow_bits, 0, final_queries, last.ood_samples, last.log_inv_rate,
                           last.num_variables - ff, ff, domain_size, two_adic_generator(log_dom - (unsigned)ff)};
    }
};

// ---------------------------------------------------------------------------------------------
// Transcript (fiat-shamir/src/prover.rs, verifier.rs, utils.rs)
// ---------------------------------------------------------------------------------------------
struct MerkleOpening {
    size_t leaf_index;
    std::vector<uint32_t> leaf_data;
    std::vector<uint32_t> path;  // log_height x 8
};

// utils.rs:30-41
static inline std::vector<EF> expand_bare_to_full(const std::vector<EF>& bare, EF alpha) {
    EF oma = ef_sub(ef_one(), alpha);
    EF tam = ef_sub(ef_add(alpha, alpha), ef_one());
    size_t d = bare.size() - 1;
    std::vector<EF> full;
    full.push_back(ef_mul(oma, bare[0]));
    for (size_t k = 1; k <= d; k++) full.push_back(ef_add(ef_mul(oma, bare[k]), ef_mul(tam, bare[k - 1])));
    full.push_back(ef_mul(tam, bare[d]));
    return full;
}
static inline std::vector<uint32_t> flatten(const std::vector<EF>& v) {
    std::vector<uint32_t> r(v.size() * 5);
    for (size_t i = 0; i < v.size(); i++) std::memcpy(&r[5 * i], v[i].v, 20);
    return r;
}

struct ProverState {
    Challenger ch;
    std::vector<uint32_t> transcript;
    std::vector<MerkleOpening> merkle_openings;
    uint64_t pow_permutations = 0;  // statistics

    void add_base_scalars(const uint32_t* s, size_t n) {
        ch.observe_many(s, n);
        transcript.insert(transcript.end(), s, s + n);
    }
    void observe_scalars(const uint32_t* s, size_t n) { ch.observe_many(s, n); }
    void add_extension_scalars(const std::vector<EF>& v) {
        std::vector<uint32_t> f = flatten(v);
        add_base_scalars(f.data(), f.size());
    }
    void duplex() { ch.duplex(); }
    EF sample() { return ch.sample_ef(); }
    std::vector<EF> sample_vec(size_t n) { return ch.sample_vec(n); }
    std::vector<size_t> sample_in_range(unsigned bits, size_t n) { return ch.sample_in_range(bits, n); }
    // prover.rs:100-114
    void add_sumcheck_polynomial(const std::vector<EF>& coeffs, const EF* eq_alpha) {
        std::vector<uint32_t> bare = flatten(coeffs);
        if (!eq_alpha) {
            ch.observe_many(bare.data(), bare.size());
        } else {
            std::vector<uint32_t> full = flatten(expand_bare_to_full(coeffs, *eq_alpha));
            ch.observe_many(full.data(), full.size());
        }
        transcript.insert(transcript.end(), bare.begin() + 5, bare.end());
    }
    // prover.rs:120-177 with the canonical (smallest) witness.  Candidates are scanned in ascending blocks; inside a
    // block the host threads (OpenMP, like the reference's rayon batches) test candidates independently and the
    // minimum hit wins, so the result does not depend on scheduling.
    void pow_grinding(size_t bits) {
        if (bits == 0) return;
        const uint32_t mask = ((uint32_t)1 << bits) - 1;
        const uint32_t BLOCK = 1u << 13;
        for (uint64_t base = 0; base < P; base += BLOCK) {
            uint32_t best = 0xffffffffu;
            const uint32_t n = (uint32_t)((uint64_t)P - base < BLOCK ? (uint64_t)P - base : BLOCK);
#pragma omp parallel for reduction(min : best) schedule(static)
            for (uint32_t i = 0; i < n; i++) {
                uint32_t w = (uint32_t)base + i;
                uint32_t s[16];
                std::memcpy(s, ch.state, 32);
                std::memset(s + 8, 0, 32);
                s[8] = to_monty(w);
                poseidon16_permute(s);
                if ((from_monty(s[8]) & mask) == 0 && w < best) best = w;
            }
            pow_permutations += n;
            if (best != 0xffffffffu) {
                uint32_t wm = to_monty(best);
                ch.observe_many(&wm, 1);
                assert((from_monty(ch.state[8]) & mask) == 0);
                transcript.push_back(wm);
                return;
            }
        }
        throw std::runtime_error("failed to find witness");
    }
    void hint_merkle_path(MerkleOpening o) { merkle_openings.push_back(std::move(o)); }
};

struct VerifierState {
    Challenger ch;
    std::vector<uint32_t> transcript;
    size_t off = 0;
    std::vector<MerkleOpening> merkle_openings;
    size_t merkle_idx = 0;

    std::vector<uint32_t> read(size_t n) {
        if (off + n > transcript.size()) throw std::runtime_error("ExceededTranscript");
        std::vector<uint32_t> r(transcript.begin() + off, transcript.begin() + off + n);
        off += n;
        return r;
    }
    std::vector<uint32_t> next_base_scalars_vec(size_t n) {
        std::vector<uint32_t> s = read(n);
        ch.observe_many(s.data(), s.size());
        return s;
    }
    std::vector<EF> next_extension_scalars_vec(size_t n) {
        std::vector<uint32_t> s = next_base_scalars_vec(n * 5);
        std::vector<EF> r(n);
        for (size_t i = 0; i < n; i++) std::memcpy(r[i].v, &s[5 * i], 20);
        return r;
    }
    void observe_scalars(const uint32_t* s, size_t n) { ch.observe_many(s, n); }
    void duplex() { ch.duplex(); }
    EF sample() { return ch.sample_ef(); }
    std::vector<EF> sample_vec(size_t n) { return ch.sample_vec(n); }
    std::vector<size_t> sample_in_range(unsigned bits, size_t n) { return ch.sample_in_range(bits, n); }
    MerkleOpening next_merkle_opening() {
        if (merkle_idx >= merkle_openings.size()) throw std::runtime_error("ExceededTranscript (merkle)");
        return merkle_openings[merkle_idx++];
    }
    // verifier.rs:146-158
    void check_pow_grinding(size_t bits) {
        if (bits == 0) return;
        uint32_t w = read(1)[0];
        ch.observe_many(&w, 1);
        if ((from_monty(ch.state[8]) & (((uint32_t)1 << bits) - 1)) != 0) throw std::runtime_error("InvalidGrindingWitness");
    }
    // verifier.rs:160-196
    std::vector<EF> next_sumcheck_polynomial(size_t n_coeffs, EF claimed_sum, const EF* eq_alpha) {
        if (!eq_alpha) {
            std::vector<uint32_t> rest = read((n_coeffs - 1) * 5);
            std::vector<EF> full(n_coeffs);
            EF s = ef_zero();
            for (size_t i = 1; i < n_coeffs; i++) {
                std::memcpy(full[i].v, &rest[5 * (i - 1)], 20);
                s = ef_add(s, full[i]);
            }
            // c0 = (claimed_sum - sum(rest)) / 2
            full[0] = ef_mul_base(ef_sub(claimed_sum, s), inv(add(ONE, ONE)));
            std::vector<uint32_t> all = flatten(full);
            ch.observe_many(all.data(), all.size());
            return full;
        }
        std::vector<uint32_t> rest = read((n_coeffs - 2) * 5);
        std::vector<EF> bare(n_coeffs - 1);
        EF s = ef_zero();
        for (size_t i = 1; i + 1 < n_coeffs; i++) {
            std::memcpy(bare[i].v, &rest[5 * (i - 1)], 20);
            s = ef_add(s, bare[i]);
        }
        bare[0] = ef_sub(claimed_sum, ef_mul(*eq_alpha, s));
        std::vector<EF> full = expand_bare_to_full(bare, *eq_alpha);
        std::vector<uint32_t> all = flatten(full);
        ch.observe_many(all.data(), all.size());
        return full;
    }
};

static inline EF poly_eval(const std::vector<EF>& coeffs, EF x) {
    EF acc = ef_zero();
    for (size_t i = coeffs.size(); i-- > 0;) acc = ef_add(ef_mul(acc, x), coeffs[i]);
    return acc;
}

// ---------------------------------------------------------------------------------------------
// Statements (whir/src/lib.rs:31-108)
// ---------------------------------------------------------------------------------------------
struct SparseValue {
    size_t selector;
    EF value;
};
struct SparseStatement {
    size_t total_num_variables;
    std::vector<EF> point;
    std::vector<SparseValue> values;
    bool is_next = false;
    size_t inner_num_variables() const { return point.size(); }
    size_t selector_num_variables() const { return total_num_variables - point.size(); }
    static SparseStatement dense(std::vector<EF> point, EF value) {
        SparseStatement s;
        s.total_num_variables = point.size();
        s.point = std::move(point);
        s.values.push_back({0, value});
        return s;
    }
};

// poly/src/next_mle.rs:35-53
static inline std::vector<EF> matrix_next_mle_folded(const std::vector<EF>& oc) {
    size_t n = oc.size();
    std::vector<EF> res((size_t)1 << n, ef_zero());
    for (size_t k = 0; k < n; k++) {
        EF prod = ef_sub(ef_one(), oc[n - k - 1]);
        for (size_t j = n - k; j < n; j++) prod = ef_mul(prod, oc[j]);
        std::vector<EF> eq = eq_table(oc.data(), n - k - 1, prod);
        for (size_t i = 0; i < eq.size(); i++) {
            size_t idx = (i << (k + 1)) + ((size_t)1 << k);
            res[idx] = ef_add(res[idx], eq[i]);
        }
    }
    EF all = ef_one();
    for (size_t j = 0; j < n; j++) all = ef_mul(all, oc[j]);
    res[((size_t)1 << n) - 1] = ef_add(res[((size_t)1 << n) - 1], all);
    return res;
}
// poly/src/next_mle.rs:9-29
static inline EF next_mle(const std::vector<EF>& x, const EF* y) {
    size_t n = x.size();
    std::vector<EF> eq_prefix(n + 1, ef_one());
    for (size_t i = 0; i < n; i++) {
        EF e = ef_add(ef_mul(x[i], y[i]), ef_mul(ef_sub(ef_one(), x[i]), ef_sub(ef_one(), y[i])));
        eq_prefix[i + 1] = ef_mul(eq_prefix[i], e);
    }
    std::vector<EF> low_suffix(n + 1, ef_one());
    for (size_t i = n; i-- > 0;) low_suffix[i] = ef_mul(ef_mul(low_suffix[i + 1], x[i]), ef_sub(ef_one(), y[i]));
    EF sum = ef_zero();
    for (size_t arr = 0; arr < n; arr++) {
        EF carry = ef_mul(ef_sub(ef_one(), x[arr]), y[arr]);
        sum = ef_add(sum, ef_mul(ef_mul(eq_prefix[arr], carry), low_suffix[arr + 1]));
    }
    EF all = ef_one();
    for (size_t i = 0; i < n; i++) all = ef_mul(ef_mul(all, x[i]), y[i]);
    return ef_add(sum, all);
}

// whir/src/open.rs:518-584: W = sum_k gamma^k * place(selector_k, eq(point) or next(point)); sum = sum_k gamma^k value_k
static inline void combine_statement(const std::vector<SparseStatement>& st, EF gamma, std::vector<EF>& W, EF& sum) {
    size_t n = st[0].total_num_variables;
    W.assign((size_t)1 << n, ef_zero());
    sum = ef_zero();
    EF gp = ef_one();
    for (const SparseStatement& s : st) {
        assert(s.total_num_variables == n);
        std::vector<EF> inner = s.is_next ? matrix_next_mle_folded(s.point) : eq_table(s.point.data(), s.point.size(), ef_one());
        for (const SparseValue& e : s.values) {
            size_t base = e.selector << s.point.size();
#pragma omp parallel for schedule(static) if (inner.size() >= 4096)
            for (size_t i = 0; i < inner.size(); i++) W[base + i] = ef_add(W[base + i], ef_mul(inner[i], gp));
            sum = ef_add(sum, ef_mul(e.value, gp));
            gp = ef_mul(gp, gamma);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Prover (commit.rs:64-99, open.rs)
// ---------------------------------------------------------------------------------------------
struct Witness {
    MerkleTree tree;
    bool tree_is_ext = false;
    std::vector<EF> ood_points, ood_answers;
};

static inline void sample_ood(ProverState& ps, size_t n_samples, size_t num_variables, const std::vector<EF>* ext,
                              const uint32_t* base, std::vector<EF>& pts, std::vector<EF>& ans) {
    pts.clear();
    ans.clear();
    if (n_samples == 0) return;
    pts = ps.sample_vec(n_samples);
    for (EF z : pts) {
        std::vector<EF> pt = expand_from_univariate(z, num_variables);
        ans.push_back(ext ? mle_eval_ext(ext->data(), num_variables, pt.data()) : mle_eval_base(base, num_variables, pt.data()));
    }
    ps.add_extension_scalars(ans);
}

static inline Witness whir_commit(const WhirConfig& c, ProverState& ps, const uint32_t* poly, size_t actual_len) {
    size_t fold0 = c.folding_factor.at_round(0);
    size_t n_blocks = (size_t)1 << fold0;
    size_t len = (size_t)1 << c.num_variables;
    size_t eff = (actual_len + len / n_blocks - 1) / (len / n_blocks);
    (void)eff;  // zero columns hash like explicit zeros; the oracle transforms all columns
    std::vector<uint32_t> m = prepare_evals_for_fft<uint32_t>(poly, len, (unsigned)fold0, (unsigned)c.starting_log_inv_rate, n_blocks);
    size_t h = (len << c.starting_log_inv_rate) >> fold0;
    dft_batch_by_evals(m.data(), h, n_blocks);
    Witness w;
    w.tree = merkle_build(m.data(), h, n_blocks, n_blocks);
    ps.add_base_scalars(w.tree.root(), 8);
    sample_ood(ps, c.commitment_ood_samples, c.num_variables, nullptr, poly, w.ood_points, w.ood_answers);
    return w;
}

struct SumcheckSingle {
    std::vector<EF> evals, weights;
    EF sum;
};

// product_computation.rs:37-125 + prove.rs:86-151 (all variants compute the same polynomial): MSB-first.
static inline std::vector<EF> run_sumcheck_rounds(SumcheckSingle& sc, ProverState& ps, size_t n_rounds, size_t pow_bits) {
    std::vector<EF> challenges;
    for (size_t r = 0; r < n_rounds; r++) {
        size_t half = sc.evals.size() / 2;
        EF c0 = ef_zero(), c2 = ef_zero();
#pragma omp parallel for schedule(static) reduction(efsum : c0, c2) if (half >= 4096)
        for (size_t i = 0; i < half; i++) {
            c0 = ef_add(c0, ef_mul(sc.evals[i], sc.weights[i]));
            c2 = ef_add(c2, ef_mul(ef_sub(sc.evals[i + half], sc.evals[i]), ef_sub(sc.weights[i + half], sc.weights[i])));
        }
        EF c1 = ef_sub(ef_sub(sc.sum, ef_add(c0, c0)), c2);
        std::vector<EF> poly{c0, c1, c2};
        ps.add_sumcheck_polynomial(poly, nullptr);
        ps.pow_grinding(pow_bits);
        EF ch = ps.sample();
        challenges.push_back(ch);
        sc.sum = poly_eval(poly, ch);
#pragma omp parallel for schedule(static) if (half >= 4096)
        for (size_t i = 0; i < half; i++) {
            sc.evals[i] = ef_add(sc.evals[i], ef_mul(ch, ef_sub(sc.evals[i + half], sc.evals[i])));
            sc.weights[i] = ef_add(sc.weights[i], ef_mul(ch, ef_sub(sc.weights[i + half], sc.weights[i])));
        }
        sc.evals.resize(half);
        sc.weights.resize(half);
    }
    return challenges;
}

static inline std::vector<MerkleOpening> open_at(const MerkleTree& t, const std::vector<size_t>& idx) {
    std::vector<MerkleOpening> r;
    size_t log_h = t.layers.size() - 1;
    for (size_t i : idx) {
        MerkleOpening o;
        o.leaf_index = i;
        o.leaf_data.resize(t.full_width);
        o.path.resize(log_h * 8);
        merkle_open(t, i, o.leaf_data.data(), o.path.data());
        r.push_back(std::move(o));
    }
    return r;
}
static inline unsigned ilog2(size_t x) {
    unsigned l = 0;
    while (((size_t)1 << (l + 1)) <= x) l++;
    return l;
}
// poly/src/evals.rs:44-56
static inline void evals_to_coeffs(std::vector<EF>& d) {
    size_t n = d.size();
    for (size_t half = 1; half < n; half <<= 1)
        for (size_t i = 0; i < n; i += 2 * half)
            for (size_t j = 0; j < half; j++) d[i + j + half] = ef_sub(d[i + j + half], d[i + j]);
    unsigned log_n = ilog2(n);
    for (size_t i = 0; i < n; i++) {
        size_t j = 0;
        for (unsigned b = 0; b < log_n; b++)
            if (i >> b & 1) j |= (size_t)1 << (log_n - 1 - b);
        if (i < j) std::swap(d[i], d[j]);
    }
}

// open.rs:37-248, 467-510.  Returns the full folding randomness.
static inline std::vector<EF> whir_prove(const WhirConfig& c, ProverState& ps, std::vector<SparseStatement> statement,
                                         Witness witness, const uint32_t* poly) {
    const size_t n = c.num_variables;
    // initialize_first_round_state
    {
        std::vector<SparseStatement> ood;
        for (size_t i = 0; i < witness.ood_points.size(); i++)
            ood.push_back(SparseStatement::dense(expand_from_univariate(witness.ood_points[i], n), witness.ood_answers[i]));
        statement.insert(statement.begin(), ood.begin(), ood.end());
    }
    ps.duplex();
    EF gamma = ps.sample();
    SumcheckSingle sc;
    combine_statement(statement, gamma, sc.weights, sc.sum);
    sc.evals.resize((size_t)1 << n);
    for (size_t i = 0; i < sc.evals.size(); i++) sc.evals[i] = ef_from_base(poly[i]);
    std::vector<EF> randomness = run_sumcheck_rounds(sc, ps, c.folding_factor.at_round(0), c.starting_folding_pow_bits);
    size_t domain_size = c.starting_domain_size();
    uint32_t next_domain_gen = two_adic_generator(ilog2(domain_size) - (unsigned)c.folding_factor.at_round(0));
    MerkleTree tree = std::move(witness.tree);
    bool tree_is_ext = false;

    for (size_t round = 0; round <= c.n_rounds(); round++) {
        size_t num_variables = n - c.folding_factor.total_number(round);
        if (round == c.n_rounds()) {
            // final_round, open.rs:182-248
            std::vector<EF> coeffs = sc.evals;
            evals_to_coeffs(coeffs);
            ps.add_extension_scalars(coeffs);
            ps.pow_grinding(c.final_query_pow_bits);
            std::vector<size_t> idx =
                ps.sample_in_range(ilog2(domain_size >> c.folding_factor.at_round(round)), c.final_queries);
            for (MerkleOpening& o : open_at(tree, idx)) ps.hint_merkle_path(std::move(o));
            if (c.final_sumcheck_rounds > 0) {
                std::vector<EF> fr = run_sumcheck_rounds(sc, ps, c.final_sumcheck_rounds, 0);
                randomness.insert(randomness.end(), fr.begin(), fr.end());
            }
            break;
        }
        const RoundConfig& rp = c.round_parameters[round];
        size_t fnext = c.folding_factor.at_round(round + 1);
        size_t new_domain_size = domain_size >> c.rs_reduction_factor(round);
        size_t inv_rate = new_domain_size >> num_variables;
        // reorder_and_dft on the folded EF evals + MerkleData::build
        size_t n_cols = (size_t)1 << fnext;
        std::vector<EF> m = prepare_evals_for_fft<EF>(sc.evals.data(), sc.evals.size(), (unsigned)fnext, ilog2(inv_rate), n_cols);
        size_t h = (sc.evals.size() * inv_rate) >> fnext;
        dft_batch_by_evals((uint32_t*)m.data(), h, n_cols * 5);
        MerkleTree new_tree = merkle_build((const uint32_t*)m.data(), h, n_cols * 5, n_cols * 5);
        ps.add_base_scalars(new_tree.root(), 8);
        std::vector<EF> ood_points, ood_answers;
        sample_ood(ps, rp.ood_samples, num_variables, &sc.evals, nullptr, ood_points, ood_answers);
        ps.pow_grinding(rp.query_pow_bits);
        // compute_stir_queries, open.rs:250-277
        std::vector<size_t> idx = ps.sample_in_range(ilog2(domain_size >> c.folding_factor.at_round(round)), rp.num_queries);
        size_t ff = c.folding_factor.at_round(round);
        std::vector<EF> folding_randomness(randomness.end() - ff, randomness.end());
        std::vector<MerkleOpening> answers = open_at(tree, idx);
        std::vector<EF> stir_evals;
        for (const MerkleOpening& o : answers) {
            if (!tree_is_ext)
                stir_evals.push_back(mle_eval_base(o.leaf_data.data(), ff, folding_randomness.data()));
            else
                stir_evals.push_back(mle_eval_ext((const EF*)o.leaf_data.data(), ff, folding_randomness.data()));
        }
        for (MerkleOpening& o : answers) ps.hint_merkle_path(std::move(o));
        ps.duplex();
        EF g = ps.sample();
        // add_new_equality / add_new_base_equality, open.rs:337-382
        EF gp = ef_one();
        for (size_t i = 0; i < ood_points.size(); i++) {
            std::vector<EF> pt = expand_from_univariate(ood_points[i], num_variables);
            std::vector<EF> eq = eq_table(pt.data(), num_variables, gp);
            for (size_t k = 0; k < eq.size(); k++) sc.weights[k] = ef_add(sc.weights[k], eq[k]);
            sc.sum = ef_add(sc.sum, ef_mul(gp, ood_answers[i]));
            gp = ef_mul(gp, g);
        }
        for (size_t i = 0; i < idx.size(); i++) {
            uint32_t z = pow_u64(next_domain_gen, idx[i]);
            std::vector<EF> pt = expand_from_univariate(ef_from_base(z), num_variables);
            std::vector<EF> eq = eq_table(pt.data(), num_variables, gp);
            for (size_t k = 0; k < eq.size(); k++) sc.weights[k] = ef_add(sc.weights[k], eq[k]);
            sc.sum = ef_add(sc.sum, ef_mul(gp, stir_evals[i]));
            gp = ef_mul(gp, g);
        }
        std::vector<EF> nr = run_sumcheck_rounds(sc, ps, fnext, rp.folding_pow_bits);
        randomness.insert(randomness.end(), nr.begin(), nr.end());
        domain_size = new_domain_size;
        next_domain_gen = two_adic_generator(ilog2(new_domain_size) - (unsigned)fnext);
        tree = std::move(new_tree);
        tree_is_ext = true;
    }
    return randomness;
}

// ---------------------------------------------------------------------------------------------
// Verifier (verify.rs)
// ---------------------------------------------------------------------------------------------
struct ParsedCommitment {
    size_t num_variables;
    uint32_t root[8];
    std::vector<EF> ood_points, ood_answers;
    std::vector<SparseStatement> oods_constraints() const {
        std::vector<SparseStatement> r;
        for (size_t i = 0; i < ood_points.size(); i++)
            r.push_back(SparseStatement::dense(expand_from_univariate(ood_points[i], num_variables), ood_answers[i]));
        return r;
    }
};
static inline ParsedCommitment parse_commitment(VerifierState& vs, size_t num_variables, size_t ood_samples) {
    ParsedCommitment pc;
    pc.num_variables = num_variables;
    std::vector<uint32_t> root = vs.next_base_scalars_vec(8);
    std::memcpy(pc.root, root.data(), 32);
    if (ood_samples > 0) {
        pc.ood_points = vs.sample_vec(ood_samples);
        pc.ood_answers = vs.next_extension_scalars_vec(ood_samples);
    }
    return pc;
}
static inline std::vector<EF> verify_sumcheck_rounds(VerifierState& vs, EF& claimed_sum, size_t rounds, size_t pow_bits) {
    std::vector<EF> r;
    for (size_t i = 0; i < rounds; i++) {
        std::vector<EF> coeffs = vs.next_sumcheck_polynomial(3, claimed_sum, nullptr);
        vs.check_pow_grinding(pow_bits);
        EF x = vs.sample();
        claimed_sum = poly_eval(coeffs, x);
        r.push_back(x);
    }
    return r;
}
static inline std::vector<EF> combine_constraints(VerifierState& vs, EF& claimed_sum, const std::vector<SparseStatement>& cs) {
    EF g = vs.sample();
    std::vector<EF> cr{ef_one()};
    for (const SparseStatement& s : cs)
        for (const SparseValue& e : s.values) {
            EF p = cr.back();
            claimed_sum = ef_add(claimed_sum, ef_mul(p, e.value));
            cr.push_back(ef_mul(p, g));
        }
    cr.pop_back();
    return cr;
}
static inline std::vector<SparseStatement> verify_stir_challenges(const WhirConfig& c, VerifierState& vs, const RoundConfig& params,
                                                                  const ParsedCommitment& commitment,
                                                                  const std::vector<EF>& folding_randomness, size_t round_index) {
    bool leafs_base = round_index == 0;
    vs.check_pow_grinding(params.query_pow_bits);
    size_t folded = params.domain_size >> params.folding_factor;
    std::vector<size_t> idx = vs.sample_in_range(ilog2(folded), params.num_queries);
    size_t width = (size_t)1 << params.folding_factor;
    std::vector<SparseStatement> out;
    for (size_t q = 0; q < idx.size(); q++) {
        MerkleOpening o = vs.next_merkle_opening();
        size_t leaf_len = leafs_base ? width : width * 5;
        if (o.leaf_data.size() != leaf_len || o.path.size() != ilog2(folded) * 8) throw std::runtime_error("InvalidProof (opening shape)");
        if (!merkle_verify(commitment.root, ilog2(folded), idx[q], o.leaf_data.data(), leaf_len, o.path.data()))
            throw std::runtime_error("InvalidProof (merkle)");
        EF fold = leafs_base ? mle_eval_base(o.leaf_data.data(), params.folding_factor, folding_randomness.data())
                             : mle_eval_ext((const EF*)o.leaf_data.data(), params.folding_factor, folding_randomness.data());
        uint32_t z = pow_u64(params.folded_domain_gen, idx[q]);
        out.push_back(SparseStatement::dense(expand_from_univariate(ef_from_base(z), params.num_variables), fold));
    }
    (void)c;
    return out;
}
// verify.rs:331-374
static inline EF eval_constraints_poly(const WhirConfig& c,
                                       const std::vector<std::pair<std::vector<EF>, std::vector<SparseStatement>>>& rcs,
                                       std::vector<EF> point) {
    EF value = ef_zero();
    for (size_t round = 0; round < rcs.size(); round++) {
        if (round > 0) {
            size_t k = c.folding_factor.at_round(round - 1);
            point.erase(point.begin(), point.begin() + k);
        }
        size_t i = 0;
        for (const SparseStatement& s : rcs[round].second) {
            const EF* inner = point.data() + (point.size() - s.inner_num_variables());
            EF common;
            if (s.is_next) {
                common = next_mle(s.point, inner);
            } else {
                common = ef_one();
                for (size_t j = 0; j < s.point.size(); j++)
                    common = ef_mul(common, ef_add(ef_mul(s.point[j], inner[j]),
                                                   ef_mul(ef_sub(ef_one(), s.point[j]), ef_sub(ef_one(), inner[j]))));
            }
            for (const SparseValue& e : s.values) {
                EF ev = common;
                size_t sv = s.selector_num_variables();
                for (size_t j = 0; j < sv; j++)
                    ev = ef_mul(ev, (e.selector >> (sv - 1 - j)) & 1 ? point[j] : ef_sub(ef_one(), point[j]));
                value = ef_add(value, ef_mul(ev, rcs[round].first[i]));
                i++;
            }
        }
        assert(i == rcs[round].first.size());
    }
    return value;
}
// poly/src/evals.rs:69-82
static inline EF eval_multilinear_coeffs(const EF* coeffs, size_t len, const EF* point) {
    if (len == 1) return coeffs[0];
    return ef_add(eval_multilinear_coeffs(coeffs, len / 2, point + 1),
                  ef_mul(eval_multilinear_coeffs(coeffs + len / 2, len / 2, point + 1), point[0]));
}
// verify.rs:83-204; throws on failure, returns the folding randomness
static inline std::vector<EF> whir_verify(const WhirConfig& c, VerifierState& vs, const ParsedCommitment& pc,
                                          const std::vector<SparseStatement>& statement) {
    std::vector<std::pair<std::vector<EF>, std::vector<SparseStatement>>> round_constraints;
    std::vector<std::vector<EF>> round_fr;
    EF claimed_sum = ef_zero();
    ParsedCommitment prev = pc;
    vs.duplex();
    std::vector<SparseStatement> constraints = prev.oods_constraints();
    constraints.insert(constraints.end(), statement.begin(), statement.end());
    std::vector<EF> cr = combine_constraints(vs, claimed_sum, constraints);
    round_constraints.push_back({cr, constraints});
    round_fr.push_back(verify_sumcheck_rounds(vs, claimed_sum, c.folding_factor.at_round(0), c.starting_folding_pow_bits));
    for (size_t round = 0; round < c.n_rounds(); round++) {
        const RoundConfig& rp = c.round_parameters[round];
        ParsedCommitment nc = parse_commitment(vs, rp.num_variables, rp.ood_samples);
        std::vector<SparseStatement> stir = verify_stir_challenges(c, vs, rp, prev, round_fr.back(), round);
        std::vector<SparseStatement> cs = nc.oods_constraints();
        cs.insert(cs.end(), stir.begin(), stir.end());
        vs.duplex();
        std::vector<EF> cr2 = combine_constraints(vs, claimed_sum, cs);
        round_constraints.push_back({cr2, cs});
        round_fr.push_back(verify_sumcheck_rounds(vs, claimed_sum, c.folding_factor.at_round(round + 1), rp.folding_pow_bits));
        prev = nc;
    }
    size_t n_final = (size_t)1 << c.n_vars_of_final_polynomial();
    std::vector<EF> final_coeffs = vs.next_extension_scalars_vec(n_final);
    std::vector<SparseStatement> stir =
        verify_stir_challenges(c, vs, c.final_round_config(), prev, round_fr.back(), c.n_rounds());
    for (const SparseStatement& s : stir) {
        EF alpha = s.point[0];
        EF u = ef_zero();
        for (size_t i = final_coeffs.size(); i-- > 0;) u = ef_add(ef_mul(u, alpha), final_coeffs[i]);
        if (!ef_eq(u, s.values[0].value)) throw std::runtime_error("InvalidProof (final stir)");
    }
    std::vector<EF> fsr = verify_sumcheck_rounds(vs, claimed_sum, c.final_sumcheck_rounds, 0);
    round_fr.push_back(fsr);
    std::vector<EF> all;
    for (auto& v : round_fr) all.insert(all.end(), v.begin(), v.end());
    EF ew = eval_constraints_poly(c, round_constraints, all);
    std::vector<EF> rev(fsr.rbegin(), fsr.rend());
    EF fv = eval_multilinear_coeffs(final_coeffs.data(), final_coeffs.size(), rev.data());
    if (!ef_eq(claimed_sum, ef_mul(ew, fv))) throw std::runtime_error("InvalidProof (final check)");
    return all;
}

}  // namespace orc
