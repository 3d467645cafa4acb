// WhirConfig::new (crates/whir/src/config.rs:186-334) and the SecurityAssumption error formulas (config.rs:445-614): the
// integers of a WHIR schedule (query counts, PoW bits, OOD samples, round structure) from a WhirConfigBuilder.
// f64 arithmetic in the reference's operation order — the results are ceil()ed, so the order matters only on exact ties,
// but it costs nothing to keep it (powi(5) follows compiler-rt's __powidf2 square-and-multiply, which is what Rust's
// f64::powi lowers to).
#include <math.h>
#include <string.h>
#include "../../../include/leanmultisig_host.h"

void lm_set_error(const char* fmt, ...);

namespace {

constexpr uint32_t FIELD_SIZE_BITS = 155;  // EF::bits() of the quintic extension of KoalaBear (5 x 31)
constexpr uint32_t TWO_ADICITY = 24;       // koala_bear.rs:48

double powi(double a, int b) {  // compiler-rt __powidf2, b >= 0
    double r = 1;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}

struct Soundness {
    uint32_t kind;  // LM_SOUNDNESS_*
    // config.rs:467-476
    double log_eta(uint32_t log_inv_rate, double log_c) const {
        if (kind == LM_SOUNDNESS_JOHNSON_BOUND) return -(0.5 * (double)log_inv_rate + log_c);
        return -((double)log_inv_rate + log_c);  // CapacityBound (UniqueDecoding never asks)
    }
    // config.rs:479-495
    double list_size_bits(uint32_t log_degree, uint32_t log_inv_rate, double log_c) const {
        if (kind == LM_SOUNDNESS_UNIQUE_DECODING) return 0.;
        const double le = log_eta(log_inv_rate, log_c);
        if (kind == LM_SOUNDNESS_JOHNSON_BOUND) {
            const double log_inv_sqrt_rate = (double)log_inv_rate / 2.;
            return log_inv_sqrt_rate - (1. + le);
        }
        return (double)(log_degree + log_inv_rate) - le;
    }
    // config.rs:498-536
    double prox_gaps_error(uint32_t log_degree, uint32_t log_inv_rate, uint32_t field_size_bits, uint32_t num_functions,
                           double log_c) const {
        double error;
        if (kind == LM_SOUNDNESS_UNIQUE_DECODING) {
            error = (double)(log_degree + log_inv_rate);
        } else {
            const double le = log_eta(log_inv_rate, log_c);
            if (kind == LM_SOUNDNESS_JOHNSON_BOUND) {  // Theorem 1.5 of BCSS25
                const double eta = pow(2.0, le);
                const double rho = 1. / (double)(1u << log_inv_rate);
                const double rho_sqrt = sqrt(rho);
                const double gamma = 1. - rho_sqrt - eta;
                const double n = (double)((uint64_t)1 << (log_degree + log_inv_rate));
                const double m = fmax(ceil(rho_sqrt / (2. * eta)), 3.);
                const double num_1 = (2. * powi(m + 0.5, 5) + 3. * (m + 0.5) * gamma * rho) * n;
                const double den_1 = 3. * rho * rho_sqrt;
                const double num_2 = m + 0.5;
                const double den_2 = rho_sqrt;
                error = log2((num_1 / den_1) + (num_2 / den_2));
            } else {
                error = (double)(log_degree + 2 * log_inv_rate) - le;
            }
        }
        const double num_functions_1_log = log2((double)num_functions - 1.);
        return (double)field_size_bits - (error + num_functions_1_log);
    }
    // config.rs:544-559
    double log_1_delta(uint32_t log_inv_rate, double log_c) const {
        const double eta = kind == LM_SOUNDNESS_UNIQUE_DECODING ? 0. : pow(2.0, log_eta(log_inv_rate, log_c));
        const double rate = 1. / (double)(1u << log_inv_rate);
        double delta;
        if (kind == LM_SOUNDNESS_UNIQUE_DECODING)
            delta = 0.5 * (1. - rate);
        else if (kind == LM_SOUNDNESS_JOHNSON_BOUND)
            delta = 1. - sqrt(rate) - eta;
        else
            delta = 1. - rate - eta;
        return log2(1. - delta);
    }
    // config.rs:563-567
    uint32_t queries(uint32_t protocol_security_level, uint32_t log_inv_rate, double log_c) const {
        return (uint32_t)ceil(-(double)protocol_security_level / log_1_delta(log_inv_rate, log_c));
    }
    // config.rs:571-575
    double queries_error(uint32_t log_inv_rate, uint32_t num_queries, double log_c) const {
        return -(double)num_queries * log_1_delta(log_inv_rate, log_c);
    }
    // config.rs:579-594
    double ood_error(uint32_t log_degree, uint32_t log_inv_rate, uint32_t field_size_bits, uint32_t ood_samples, double log_c) const {
        if (kind == LM_SOUNDNESS_UNIQUE_DECODING) return 0.;
        const double ls = list_size_bits(log_degree, log_inv_rate, log_c);
        const double error = 2. * ls + (double)(log_degree * ood_samples);
        return (double)(ood_samples * field_size_bits) + 1. - error;
    }
    // config.rs:598-613; 0 = "Could not find an appropriate number of OOD samples" for anything but UniqueDecoding
    uint32_t determine_ood_samples(uint32_t security_level, uint32_t log_degree, uint32_t log_inv_rate, uint32_t field_size_bits,
                                   double log_c) const {
        if (kind == LM_SOUNDNESS_UNIQUE_DECODING) return 0;
        for (uint32_t s = 1; s < 64; s++)
            if (ood_error(log_degree, log_inv_rate, field_size_bits, s, log_c) >= (double)security_level) return s;
        return 0;
    }
};

// config.rs:373-401
double folding_pow_bits(uint32_t security_level, const Soundness& s, uint32_t fsb, uint32_t nv, uint32_t lir, double log_c) {
    const double prox = s.prox_gaps_error(nv, lir, fsb, 2, log_c);
    const double sumcheck = (double)fsb - (s.list_size_bits(nv, lir, log_c) + 1.);
    return fmax(0., (double)security_level - fmin(prox, sumcheck));
}
// config.rs:403-418
double queries_combination(const Soundness& s, uint32_t fsb, uint32_t nv, uint32_t lir, uint32_t ood, uint32_t nq, double log_c) {
    const double ls = s.list_size_bits(nv, lir, log_c);
    return (double)fsb - (log2((double)(ood + nq)) + ls + 1.);
}
// compute_optimal_log_c_for_rate, config.rs:146-183
double optimal_log_c(const lm_whir_builder* b, const Soundness& s, uint32_t nv, uint32_t lir) {
    if (s.kind == LM_SOUNDNESS_UNIQUE_DECODING) return 0.0;
    const uint32_t qsl = b->security_level > b->pow_bits ? b->security_level - b->pow_bits : 0;
    uint32_t best_m = 3;
    uint64_t best_queries = ~0ull;
    for (uint32_t m = 3; m <= 100; m++) {
        const double log_c = log2(2.0 * (double)m);
        const double fp = folding_pow_bits(b->security_level, s, FIELD_SIZE_BITS, nv, lir, log_c);
        if ((uint64_t)ceil(fp) > b->pow_bits) break;
        const uint64_t q = s.queries(qsl, lir, log_c);
        if (q < best_queries) {
            best_queries = q;
            best_m = m;
        }
    }
    return log2(2.0 * (double)best_m);
}

}  // namespace

extern "C" {

void lmh_default_whir_builder(uint32_t starting_log_inv_rate, int prox_gaps_conjecture, lm_whir_builder* out) {
    memset(out, 0, sizeof *out);
    out->starting_log_inv_rate = starting_log_inv_rate;
    out->max_num_variables_to_send_coeffs = 8;
    out->rs_domain_initial_reduction_factor = 5;
    out->folding_factor_first = 7;
    out->folding_factor_subsequent = 5;
    out->soundness_type = prox_gaps_conjecture ? LM_SOUNDNESS_CAPACITY_BOUND : LM_SOUNDNESS_JOHNSON_BOUND;
    out->security_level = 124;
    out->pow_bits = 16;
}

int lmh_whir_config_new(const lm_whir_builder* b, uint32_t num_variables, lm_whir_config* out) {
    if (!b || !out) return LM_E_INVALID;
    const uint32_t f0 = b->folding_factor_first, f1 = b->folding_factor_subsequent;
    // FoldingFactor::check_validity, config.rs:38-49; "Increasing the code rate is not a good idea" :190-193
    if (f0 == 0 || f1 == 0 || f0 > num_variables || f1 > num_variables || b->rs_domain_initial_reduction_factor > f0 ||
        b->soundness_type > LM_SOUNDNESS_CAPACITY_BOUND || num_variables + b->starting_log_inv_rate > 40) {
        lm_set_error("lmh_whir_config_new: invalid folding factors / reduction factor for %u variables", num_variables);
        return LM_E_INVALID;
    }
    if (num_variables + b->starting_log_inv_rate - f0 > TWO_ADICITY) {  // "Increase folding_factor_0", :202-206
        lm_set_error("lmh_whir_config_new: folded domain 2^%u exceeds the two-adicity", num_variables + b->starting_log_inv_rate - f0);
        return LM_E_INVALID;
    }
    if (FIELD_SIZE_BITS <= b->security_level) {  // :311-314
        lm_set_error("lmh_whir_config_new: field size must be greater than the security level");
        return LM_E_INVALID;
    }
    const Soundness s{b->soundness_type};
    const uint32_t qsl = b->security_level > b->pow_bits ? b->security_level - b->pow_bits : 0;
    uint32_t log_inv_rate = b->starting_log_inv_rate;
    // compute_number_of_rounds, config.rs:53-72
    uint32_t num_rounds, final_sumcheck_rounds;
    const uint32_t nv_except_first = num_variables - f0;
    if (nv_except_first < b->max_num_variables_to_send_coeffs) {
        num_rounds = 0;
        final_sumcheck_rounds = nv_except_first;
    } else {
        num_rounds = (nv_except_first - b->max_num_variables_to_send_coeffs + f1 - 1) / f1;
        if (num_rounds * f1 > nv_except_first) {
            lm_set_error("lmh_whir_config_new: the subsequent folding factor overshoots the polynomial");
            return LM_E_INVALID;
        }
        final_sumcheck_rounds = nv_except_first - num_rounds * f1;
    }
    if (num_rounds > LM_MAX_WHIR_ROUNDS) {
        lm_set_error("lmh_whir_config_new: %u rounds > LM_MAX_WHIR_ROUNDS", num_rounds);
        return LM_E_INVALID;
    }
    memset(out, 0, sizeof *out);
    double log_c_old = optimal_log_c(b, s, num_variables, log_inv_rate);
    out->num_variables = num_variables;
    out->starting_log_inv_rate = b->starting_log_inv_rate;
    out->folding_factor_first = f0;
    out->folding_factor_subsequent = f1;
    out->rs_domain_initial_reduction_factor = b->rs_domain_initial_reduction_factor;
    out->commitment_ood_samples = s.determine_ood_samples(b->security_level, num_variables, log_inv_rate, FIELD_SIZE_BITS, log_c_old);
    out->starting_folding_pow_bits = (uint32_t)ceil(folding_pow_bits(b->security_level, s, FIELD_SIZE_BITS, num_variables, log_inv_rate, log_c_old));
    out->n_rounds = num_rounds;
    uint32_t nv_moving = num_variables - f0;
    for (uint32_t round = 0; round < num_rounds; round++) {
        // queries are set w.r.t. the old rate, the rest w.r.t. the new one (config.rs:236-298)
        const uint32_t rs_red = round == 0 ? b->rs_domain_initial_reduction_factor : 1;
        const uint32_t fold = round == 0 ? f0 : f1;
        if (fold < rs_red) {
            lm_set_error("lmh_whir_config_new: reduction factor above the folding factor");
            return LM_E_INVALID;
        }
        const uint32_t next_rate = log_inv_rate + (fold - rs_red);
        const double log_c_new = optimal_log_c(b, s, nv_moving, next_rate);
        const uint32_t num_queries = s.queries(qsl, log_inv_rate, log_c_old);
        const uint32_t ood = s.determine_ood_samples(b->security_level, nv_moving, next_rate, FIELD_SIZE_BITS, log_c_new);
        const double query_error = s.queries_error(log_inv_rate, num_queries, log_c_old);
        const double comb_error = queries_combination(s, FIELD_SIZE_BITS, nv_moving, next_rate, ood, num_queries, log_c_new);
        const double query_pow = fmax(0., (double)b->security_level - fmin(query_error, comb_error));
        const double fold_pow = folding_pow_bits(b->security_level, s, FIELD_SIZE_BITS, nv_moving, next_rate, log_c_new);
        out->rounds[round].query_pow_bits = (uint32_t)ceil(query_pow);
        out->rounds[round].folding_pow_bits = (uint32_t)ceil(fold_pow);
        out->rounds[round].num_queries = num_queries;
        out->rounds[round].ood_samples = ood;
        nv_moving -= f1;  // at_round(round + 1)
        log_inv_rate = next_rate;
        log_c_old = log_c_new;
    }
    out->final_queries = s.queries(qsl, log_inv_rate, log_c_old);
    out->final_query_pow_bits =
        (uint32_t)ceil(fmax(0., (double)b->security_level - s.queries_error(log_inv_rate, out->final_queries, log_c_old)));
    out->final_sumcheck_rounds = final_sumcheck_rounds;
    return LM_OK;
}

}  // extern "C"
