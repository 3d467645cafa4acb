// ORACLE — TEST INFRASTRUCTURE ONLY (see kb_oracle.hpp header).  C entry points for ctypes.
#include "kb_oracle.hpp"
#ifdef _OPENMP
#include <omp.h>
#endif
using namespace orc;

extern "C" {

uint32_t orc_to_monty(uint32_t x) { return to_monty(x); }
uint32_t orc_from_monty(uint32_t x) { return from_monty(x); }
uint32_t orc_add(uint32_t a, uint32_t b) { return add(a, b); }
uint32_t orc_sub(uint32_t a, uint32_t b) { return sub(a, b); }
uint32_t orc_mul(uint32_t a, uint32_t b) { return mul(a, b); }
uint32_t orc_inv(uint32_t a) { return inv(a); }
uint32_t orc_two_adic_generator(uint32_t bits) { return two_adic_generator(bits); }

void orc_ef_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    EF x, y;
    std::memcpy(x.v, a, 20);
    std::memcpy(y.v, b, 20);
    EF r = ef_mul(x, y);
    std::memcpy(out, r.v, 20);
}
void orc_ef_inv(const uint32_t* a, uint32_t* out) {
    EF x;
    std::memcpy(x.v, a, 20);
    EF r = ef_inv(x);
    std::memcpy(out, r.v, 20);
}

void orc_poseidon16_permute(uint32_t* state, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) poseidon16_permute(state + 16 * i);
}
void orc_poseidon16_compress(uint32_t* state, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) poseidon16_compress(state + 16 * i);
}
void orc_hash_slice(const uint32_t* data, uint64_t len, uint32_t* out8) { hash_slice(data, len, out8); }

// LDE of `evals` (base words, len = 2^n) -> row-major h x dft_n_cols matrix (out), h = len*2^rate / 2^fold
void orc_lde_base(const uint32_t* evals, uint64_t len, uint32_t fold, uint32_t log_inv_rate, uint64_t dft_n_cols,
                  uint32_t* out) {
    std::vector<uint32_t> m = prepare_evals_for_fft<uint32_t>(evals, len, fold, log_inv_rate, dft_n_cols);
    size_t h = (len << log_inv_rate) >> fold;
    dft_batch_by_evals(m.data(), h, dft_n_cols);
    std::memcpy(out, m.data(), m.size() * 4);
}
// EF version: evals = len x 5 words (AoS); out = row-major h x (dft_n_cols*5) base words
// (whir/src/dft.rs:147-155: EF matrices are flattened to 5x wider base matrices).
void orc_lde_ext(const uint32_t* evals, uint64_t len, uint32_t fold, uint32_t log_inv_rate, uint64_t dft_n_cols,
                 uint32_t* out) {
    std::vector<EF> m = prepare_evals_for_fft<EF>((const EF*)evals, len, fold, log_inv_rate, dft_n_cols);
    size_t h = (len << log_inv_rate) >> fold;
    dft_batch_by_evals((uint32_t*)m.data(), h, dft_n_cols * 5);
    std::memcpy(out, m.data(), m.size() * 20);
}

// Merkle: rows = row-major height x width; leaves zero-padded to full_width.
// out_layers: concatenation of all digest layers bottom-up ((2*height - 1) * 8 words).
void orc_merkle_build(const uint32_t* rows, uint64_t height, uint64_t width, uint64_t full_width,
                      uint32_t* out_layers) {
    MerkleTree t = merkle_build(rows, height, width, full_width);
    size_t off = 0;
    for (auto& l : t.layers) {
        std::memcpy(out_layers + off, l.data(), l.size() * 4);
        off += l.size();
    }
}
int orc_merkle_verify(const uint32_t* root, uint64_t log_height, uint64_t index, const uint32_t* leaf,
                      uint64_t leaf_len, const uint32_t* siblings) {
    return merkle_verify(root, log_height, index, leaf, leaf_len, siblings) ? 1 : 0;
}

// MLE evaluation; point = n_vars x 5 words, point[0] <-> MSB.
void orc_mle_eval_base(const uint32_t* v, uint32_t n_vars, const uint32_t* point, uint32_t* out5) {
    EF r = mle_eval_base(v, n_vars, (const EF*)point);
    std::memcpy(out5, r.v, 20);
}
void orc_mle_eval_ext(const uint32_t* v, uint32_t n_vars, const uint32_t* point, uint32_t* out5) {
    EF r = mle_eval_ext((const EF*)v, n_vars, (const EF*)point);
    std::memcpy(out5, r.v, 20);
}
void orc_expand_from_univariate(const uint32_t* a5, uint32_t n, uint32_t* out) {
    EF a;
    std::memcpy(a.v, a5, 20);
    std::vector<EF> p = expand_from_univariate(a, n);
    std::memcpy(out, p.data(), n * 20);
}
void orc_eq_table(const uint32_t* point, uint32_t n, const uint32_t* scalar5, uint32_t* out) {
    EF s;
    std::memcpy(s.v, scalar5, 20);
    std::vector<EF> t = eq_table((const EF*)point, n, s);
    std::memcpy(out, t.data(), t.size() * 20);
}

}  // extern "C"

// ================================================================================================
// WHIR (whir_oracle.hpp)
// ================================================================================================
#include "whir_oracle.hpp"

namespace {
WhirConfigBuilder make_builder(const uint32_t* b) {
    // b = [starting_log_inv_rate, max_num_variables_to_send_coeffs, rs_domain_initial_reduction_factor,
    //      fold_first, fold_subsequent, soundness_type, security_level, pow_bits]
    WhirConfigBuilder w;
    w.starting_log_inv_rate = b[0];
    w.max_num_variables_to_send_coeffs = b[1];
    w.rs_domain_initial_reduction_factor = b[2];
    w.folding_factor = FoldingFactor{b[3], b[4]};
    w.soundness_type = (SecurityAssumption)b[5];
    w.security_level = b[6];
    w.pow_bits = b[7];
    return w;
}
// statements blob (u32 words): [n_statements] then per statement:
//   [point_len, is_next, n_values] [point: 5*point_len] then n_values x [selector_lo, selector_hi, value x5]
std::vector<SparseStatement> parse_statements(const uint32_t* blob, size_t total_vars) {
    std::vector<SparseStatement> out;
    size_t k = 0;
    uint32_t n = blob[k++];
    for (uint32_t s = 0; s < n; s++) {
        SparseStatement st;
        st.total_num_variables = total_vars;
        uint32_t pl = blob[k++];
        st.is_next = blob[k++] != 0;
        uint32_t nv = blob[k++];
        st.point.resize(pl);
        for (uint32_t i = 0; i < pl; i++) {
            std::memcpy(st.point[i].v, blob + k, 20);
            k += 5;
        }
        for (uint32_t i = 0; i < nv; i++) {
            SparseValue v;
            v.selector = (size_t)blob[k] | ((size_t)blob[k + 1] << 32);
            k += 2;
            std::memcpy(v.value.v, blob + k, 20);
            k += 5;
            st.values.push_back(v);
        }
        out.push_back(std::move(st));
    }
    return out;
}
std::vector<uint32_t> serialize_proof(const ProverState& ps) {
    std::vector<uint32_t> o;
    o.push_back((uint32_t)ps.transcript.size());
    o.insert(o.end(), ps.transcript.begin(), ps.transcript.end());
    o.push_back((uint32_t)ps.merkle_openings.size());
    for (const MerkleOpening& m : ps.merkle_openings) {
        o.push_back((uint32_t)m.leaf_index);
        o.push_back((uint32_t)((uint64_t)m.leaf_index >> 32));
        o.push_back((uint32_t)m.leaf_data.size());
        o.push_back((uint32_t)m.path.size());
        o.insert(o.end(), m.leaf_data.begin(), m.leaf_data.end());
        o.insert(o.end(), m.path.begin(), m.path.end());
    }
    return o;
}
void parse_proof(const uint32_t* blob, VerifierState& vs) {
    size_t k = 0;
    uint32_t t = blob[k++];
    vs.transcript.assign(blob + k, blob + k + t);
    k += t;
    uint32_t m = blob[k++];
    for (uint32_t i = 0; i < m; i++) {
        MerkleOpening o;
        o.leaf_index = (size_t)blob[k] | ((size_t)blob[k + 1] << 32);
        uint32_t ll = blob[k + 2], pl = blob[k + 3];
        k += 4;
        o.leaf_data.assign(blob + k, blob + k + ll);
        k += ll;
        o.path.assign(blob + k, blob + k + pl);
        k += pl;
        vs.merkle_openings.push_back(std::move(o));
    }
}
std::vector<uint32_t> g_last_proof;
}  // namespace

extern "C" {

// out: [commitment_ood_samples, starting_folding_pow_bits, n_rounds, final_queries, final_query_pow_bits,
//       final_sumcheck_rounds, final_log_inv_rate] then per round [query_pow_bits, folding_pow_bits, num_queries,
//       ood_samples, log_inv_rate, num_variables]; returns number of words written
uint32_t orc_whir_config(const uint32_t* builder, uint32_t num_variables, uint32_t* out) {
    WhirConfig c = WhirConfig::make(make_builder(builder), num_variables);
    uint32_t k = 0;
    out[k++] = (uint32_t)c.commitment_ood_samples;
    out[k++] = (uint32_t)c.starting_folding_pow_bits;
    out[k++] = (uint32_t)c.n_rounds();
    out[k++] = (uint32_t)c.final_queries;
    out[k++] = (uint32_t)c.final_query_pow_bits;
    out[k++] = (uint32_t)c.final_sumcheck_rounds;
    out[k++] = (uint32_t)c.final_log_inv_rate;
    for (const RoundConfig& r : c.round_parameters) {
        out[k++] = (uint32_t)r.query_pow_bits;
        out[k++] = (uint32_t)r.folding_pow_bits;
        out[k++] = (uint32_t)r.num_queries;
        out[k++] = (uint32_t)r.ood_samples;
        out[k++] = (uint32_t)r.log_inv_rate;
        out[k++] = (uint32_t)r.num_variables;
    }
    return k;
}

// commit + prove.  `prefix` words are absorbed with add_base_scalars first (binds the test transcript to its context).
// Returns the proof size in words (fetch with orc_last_proof); out_point = num_variables x 5.
uint64_t orc_whir_prove(const uint32_t* builder, uint32_t num_variables, const uint32_t* poly, uint64_t actual_len,
                        const uint32_t* statements_blob, const uint32_t* prefix, uint32_t n_prefix, uint32_t* out_point,
                        uint64_t* out_pow_perms) {
    WhirConfig c = WhirConfig::make(make_builder(builder), num_variables);
    ProverState ps;
    if (n_prefix) ps.add_base_scalars(prefix, n_prefix);
    Witness w = whir_commit(c, ps, poly, actual_len);
    std::vector<SparseStatement> st = parse_statements(statements_blob, num_variables);
    std::vector<EF> pt = whir_prove(c, ps, st, std::move(w), poly);
    std::memcpy(out_point, pt.data(), pt.size() * 20);
    if (out_pow_perms) *out_pow_perms = ps.pow_permutations;
    g_last_proof = serialize_proof(ps);
    return g_last_proof.size();
}
void orc_last_proof(uint32_t* out) { std::memcpy(out, g_last_proof.data(), g_last_proof.size() * 4); }

// returns 1 if the proof verifies (and writes the folding randomness), 0 otherwise (message in orc_last_error)
static char g_verr[256];
const char* orc_last_error() { return g_verr; }
int orc_whir_verify(const uint32_t* builder, uint32_t num_variables, const uint32_t* proof_blob,
                    const uint32_t* statements_blob, const uint32_t* prefix, uint32_t n_prefix, uint32_t* out_point) {
    try {
        WhirConfig c = WhirConfig::make(make_builder(builder), num_variables);
        VerifierState vs;
        parse_proof(proof_blob, vs);
        if (n_prefix) {
            std::vector<uint32_t> got = vs.next_base_scalars_vec(n_prefix);
            if (std::memcmp(got.data(), prefix, n_prefix * 4) != 0) throw std::runtime_error("prefix mismatch");
        }
        ParsedCommitment pc = parse_commitment(vs, num_variables, c.commitment_ood_samples);
        std::vector<SparseStatement> st = parse_statements(statements_blob, num_variables);
        std::vector<EF> pt = whir_verify(c, vs, pc, st);
        if (vs.off != vs.transcript.size()) throw std::runtime_error("trailing transcript data");
        if (vs.merkle_idx != vs.merkle_openings.size()) throw std::runtime_error("unused merkle openings");
        if (out_point) std::memcpy(out_point, pt.data(), pt.size() * 20);
        g_verr[0] = 0;
        return 1;
    } catch (const std::exception& e) {
        snprintf(g_verr, sizeof g_verr, "%s", e.what());
        return 0;
    }
}

}  // extern "C"

// ================================================================================================
// GKR quotient (gkr_oracle.hpp)
// ================================================================================================
#include "gkr_oracle.hpp"
extern "C" {
// nums: 2^n base words; dens: 2^n x 5.  out: quotient[5], point[n*5], claims[10].  Returns proof words (orc_last_proof).
uint64_t orc_gkr_prove(const uint32_t* nums, const uint32_t* dens, uint32_t n_vars, uint32_t* out_quotient,
                       uint32_t* out_point, uint32_t* out_claims) {
    ProverState ps;
    EF q, cn, cd;
    std::vector<EF> pt;
    gkr_prove(ps, nums, (const EF*)dens, n_vars, q, pt, cn, cd);
    std::memcpy(out_quotient, q.v, 20);
    std::memcpy(out_point, pt.data(), pt.size() * 20);
    std::memcpy(out_claims, cn.v, 20);
    std::memcpy(out_claims + 5, cd.v, 20);
    g_last_proof = serialize_proof(ps);
    return g_last_proof.size();
}
int orc_gkr_verify(const uint32_t* proof_blob, uint32_t n_vars, uint32_t* out_quotient, uint32_t* out_point,
                   uint32_t* out_claims) {
    try {
        VerifierState vs;
        parse_proof(proof_blob, vs);
        EF q, cn, cd;
        std::vector<EF> pt;
        gkr_verify(vs, n_vars, q, pt, cn, cd);
        if (vs.off != vs.transcript.size()) throw std::runtime_error("trailing transcript data");
        std::memcpy(out_quotient, q.v, 20);
        std::memcpy(out_point, pt.data(), pt.size() * 20);
        std::memcpy(out_claims, cn.v, 20);
        std::memcpy(out_claims + 5, cd.v, 20);
        g_verr[0] = 0;
        return 1;
    } catch (const std::exception& e) {
        snprintf(g_verr, sizeof g_verr, "%s", e.what());
        return 0;
    }
}
}

// ================================================================================================
// AIR sumcheck (air_oracle.hpp)
// ================================================================================================
#include "air_oracle.hpp"
namespace {
// blob (u32 words): [n_sessions] [alpha x5] [bus_beta x5] [eta x5] [logup_alphas_eq_poly 16x5]
// per session: [table, n_vars] [eq_point n_vars x5] [sum x5] [columns: n_columns x 2^n_vars base words, column major]
struct AirProblem {
    std::vector<AirSession> sessions;
    EF eta;
};
AirProblem parse_air_problem(const uint32_t* b) {
    AirProblem pr;
    size_t k = 0;
    uint32_t ns = b[k++];
    EF alpha, beta;
    std::memcpy(alpha.v, b + k, 20); k += 5;
    std::memcpy(beta.v, b + k, 20); k += 5;
    std::memcpy(pr.eta.v, b + k, 20); k += 5;
    AirExtra ex;
    ex.bus_beta = beta;
    ex.logup_alphas_eq_poly.resize(16);
    for (int i = 0; i < 16; i++) { std::memcpy(ex.logup_alphas_eq_poly[i].v, b + k, 20); k += 5; }
    ex.alpha_powers.resize(101);
    ex.alpha_powers[0] = ef_one();
    for (int i = 1; i < 101; i++) ex.alpha_powers[i] = ef_mul(ex.alpha_powers[i - 1], alpha);
    for (uint32_t s = 0; s < ns; s++) {
        AirSession se;
        se.table = (int)b[k++];
        se.n_vars = b[k++];
        se.eq_factor.resize(se.n_vars);
        for (size_t i = 0; i < se.n_vars; i++) { std::memcpy(se.eq_factor[i].v, b + k, 20); k += 5; }
        std::memcpy(se.sum.v, b + k, 20); k += 5;
        se.mmf = ef_one();
        se.extra = ex;
        size_t n = (size_t)1 << se.n_vars, nc = air_n_columns(se.table), nsft = air_n_shift(se.table);
        se.cols.resize(nc + nsft);
        for (size_t c = 0; c < nc; c++) {
            se.cols[c].resize(n);
            for (size_t i = 0; i < n; i++) se.cols[c][i] = ef_from_base(b[k + c * n + i]);
        }
        for (size_t c = 0; c < nsft; c++) {
            std::vector<uint32_t> sh = shifted_column(b + k + c * n, n);
            se.cols[nc + c].resize(n);
            for (size_t i = 0; i < n; i++) se.cols[nc + c][i] = ef_from_base(sh[i]);
        }
        k += nc * n;
        pr.sessions.push_back(std::move(se));
    }
    return pr;
}
}  // namespace
extern "C" {
// Runs prove_batched_air_sumcheck then sends the final column evals (prove_execution.rs:209-214).
// out_point: n_max x 5 challenges (sumcheck order, LSB first); out_evals: concatenated final column evals.
uint64_t orc_air_prove(const uint32_t* blob, uint32_t* out_point, uint32_t* out_evals) {
    AirProblem pr = parse_air_problem(blob);
    ProverState ps;
    std::vector<EF> ch = prove_batched_air_sumcheck(ps, pr.sessions, pr.eta);
    std::memcpy(out_point, ch.data(), ch.size() * 20);
    size_t k = 0;
    for (auto& s : pr.sessions) {
        std::vector<EF> ev = s.final_column_evals();
        ps.add_extension_scalars(ev);
        std::memcpy(out_evals + k, ev.data(), ev.size() * 20);
        k += ev.size() * 5;
    }
    g_last_proof = serialize_proof(ps);
    return g_last_proof.size();
}
// Verifier side of the same slice (verify_execution.rs:109-170): the blob carries the same public data (columns ignored).
int orc_air_verify(const uint32_t* blob, const uint32_t* proof_blob) {
    try {
        AirProblem pr = parse_air_problem(blob);
        VerifierState vs;
        parse_proof(proof_blob, vs);
        size_t n_max = 0, max_full_degree = 1;
        for (auto& s : pr.sessions) { n_max = std::max(n_max, s.n_vars); max_full_degree = std::max(max_full_degree, s.degree() + 1); }
        EF target = ef_zero(), ep = ef_one();
        std::vector<EF> eta_p;
        for (auto& s : pr.sessions) { target = ef_add(target, ef_mul(ep, s.sum)); eta_p.push_back(ep); ep = ef_mul(ep, pr.eta); }
        std::vector<EF> point;
        for (size_t r = 0; r < n_max; r++) {  // sumcheck_verify, sumcheck/src/verify.rs:5-27
            std::vector<EF> coeffs = vs.next_sumcheck_polynomial(max_full_degree + 1, target, nullptr);
            EF c = vs.sample();
            point.push_back(c);
            target = poly_eval(coeffs, c);
        }
        EF mine = ef_zero();
        for (size_t i = 0; i < pr.sessions.size(); i++) {
            AirSession& s = pr.sessions[i];
            size_t nct = air_n_columns(s.table) + air_n_shift(s.table);
            std::vector<EF> ev = vs.next_extension_scalars_vec(nct);
            EF ce = air_eval(s.table, ev.data(), s.extra);
            // back_loaded_table_contribution, verify_execution.rs:236-251
            size_t suffix_start = n_max - s.n_vars;
            EF eqv = ef_one();
            for (size_t j = 0; j < s.n_vars; j++) {
                EF nat = point[n_max - 1 - j];  // natural_ordering_point_for_session
                EF b = s.eq_factor[j];
                eqv = ef_mul(eqv, ef_add(ef_mul(b, nat), ef_mul(ef_sub(ef_one(), b), ef_sub(ef_one(), nat))));
            }
            EF kt = ef_one();
            for (size_t j = 0; j < suffix_start; j++) kt = ef_mul(kt, point[j]);
            mine = ef_add(mine, ef_mul(ef_mul(ef_mul(eta_p[i], kt), eqv), ce));
        }
        if (!ef_eq(mine, target)) throw std::runtime_error("InvalidProof (air final value)");
        if (vs.off != vs.transcript.size()) throw std::runtime_error("trailing transcript data");
        g_verr[0] = 0;
        return 1;
    } catch (const std::exception& e) {
        snprintf(g_verr, sizeof g_verr, "%s", e.what());
        return 0;
    }
}
// single-point AIR evaluation (EF columns): values = (n_columns + n_shift) x 5; uses the extra data of `blob` header
void orc_air_eval(const uint32_t* blob_header, uint32_t table, const uint32_t* values, uint32_t* out5) {
    // header = [0] [alpha][beta][eta][eqpoly]: reuse the parser with zero sessions
    AirProblem pr;
    std::vector<uint32_t> tmp(blob_header, blob_header + 1 + 15 + 80);
    tmp[0] = 0;
    parse_air_problem(tmp.data());
    AirExtra ex;
    size_t k = 1;
    EF alpha;
    std::memcpy(alpha.v, blob_header + k, 20); k += 5;
    std::memcpy(ex.bus_beta.v, blob_header + k, 20); k += 10;
    ex.logup_alphas_eq_poly.resize(16);
    for (int i = 0; i < 16; i++) { std::memcpy(ex.logup_alphas_eq_poly[i].v, blob_header + k, 20); k += 5; }
    ex.alpha_powers.resize(101);
    ex.alpha_powers[0] = ef_one();
    for (int i = 1; i < 101; i++) ex.alpha_powers[i] = ef_mul(ex.alpha_powers[i - 1], alpha);
    EF r = air_eval((int)table, (const EF*)values, ex);
    std::memcpy(out5, r.v, 20);
}
// fill a Poseidon table row (109 words; first 25 given) — trace_gen.rs:44-112
void orc_poseidon16_fill_rows(uint32_t* rows, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) poseidon16_fill_row(rows + 109 * i);
}
// trace_gen.rs:118-147: permute = 0 rows take their unconstrained output columns from memory.  rows: n x 109 row-major.
void orc_poseidon16_outputs_from_memory(uint32_t* rows, uint64_t n, const uint32_t* memory, uint64_t mem_len) {
    for (uint64_t i = 0; i < n; i++) {
        uint32_t* row = rows + 109 * i;
        if (row[8] != 0) continue;  // POSEIDON_16_COL_FLAG_PERMUTE
        const uint64_t base = from_monty(row[2]);  // POSEIDON_16_COL_INDEX_INPUT_RES
        auto mem = [&](uint64_t a) { return a < mem_len ? memory[a] : 0u; };
        if (row[3] == ONE)  // POSEIDON_16_COL_FLAG_HALF_OUTPUT
            for (int j = 0; j < 4; j++) row[97 + j] = mem(base + 4 + j);
        for (int j = 0; j < 8; j++) row[101 + j] = mem(base + 8 + j);
    }
}
// get_execution_trace, main loop (lean_prover/src/trace_gen.rs:27-100), statement by statement.  pcs / fps: canonical
// integers; bytecode: rows x 16 Montgomery words (12 used); memory: padded image; out: 24 columns x n_cycles, column-major.
// "Instruction::Precompile" is recognised from the decoded fields: the only instruction kind with aux = mul = jump = 0.
void orc_execution_table_fill(const uint32_t* pcs, const uint32_t* fps, uint64_t n_cycles, const uint32_t* bytecode,
                              uint64_t bytecode_rows, const uint32_t* memory, uint64_t mem_len, uint32_t* out) {
    const uint32_t TWO = add(ONE, ONE);
    auto mem = [&](uint32_t addr_m) -> uint32_t {
        const uint64_t a = from_monty(addr_m);
        return a < mem_len ? memory[a] : 0u;
    };
    for (uint64_t i = 0; i < n_cycles; i++) {
        uint32_t f[12] = {0};
        if (pcs[i] < bytecode_rows) std::memcpy(f, bytecode + (uint64_t)pcs[i] * 16, 48);
        const uint32_t operand_a = f[0], operand_b = f[1], operand_c = f[2], flag_a = f[3], flag_b = f[4], flag_c = f[5];
        const uint32_t flag_c_fp = f[6], flag_ab_fp = f[7], mul_ = f[8], jump = f[9], aux = f[10];
        const uint32_t fp = to_monty(fps[i]);
        const bool is_deref = aux == TWO;
        uint32_t addr_a = 0;
        if (flag_a == 0 && flag_ab_fp == 0) addr_a = add(fp, operand_a);
        const uint32_t value_a = mem(addr_a);
        uint32_t addr_b = 0;
        if (flag_b == 0 && flag_ab_fp == 0)
            addr_b = add(fp, operand_b);
        else if (is_deref)
            addr_b = add(value_a, operand_b);
        const uint32_t value_b = mem(addr_b);
        uint32_t addr_c = 0;
        if (flag_c == 0 && flag_c_fp == 0) addr_c = add(fp, operand_c);
        const uint32_t value_c = mem(addr_c);
        auto nu = [&](uint32_t flag, uint32_t flag_fp, uint32_t operand, uint32_t value) {
            return add(add(mul(flag, operand), mul(sub(sub(ONE, flag), flag_fp), value)), mul(flag_fp, add(fp, operand)));
        };
        auto col = [&](int c) -> uint32_t& { return out[(uint64_t)c * n_cycles + i]; };
        for (int j = 0; j < 12; j++) col(8 + j) = f[j];
        col(20) = (aux == 0 && mul_ == 0 && jump == 0) ? ONE : 0;
        col(21) = nu(flag_a, flag_ab_fp, operand_a, value_a);
        col(22) = nu(flag_b, flag_ab_fp, operand_b, value_b);
        col(23) = nu(flag_c, flag_c_fp, operand_c, value_c);
        col(5) = value_a, col(6) = value_b, col(7) = value_c;
        col(0) = to_monty(pcs[i]), col(1) = fp;
        col(2) = addr_a, col(3) = addr_b, col(4) = addr_c;
    }
}
}

// ================================================================================================
// logup fill (logup_oracle.hpp)
// ================================================================================================
#include "logup_oracle.hpp"
extern "C" {
// tables_desc: n_tables x [table, log_rows]; tables_cols: concatenated column-major traces (n_columns_total x rows each).
// out_nums / out_dens sized for the next power of two of the active length; returns total_active_len.
uint64_t orc_logup_fill(const uint32_t* memory, const uint32_t* memory_acc, uint32_t log_mem, const uint32_t* bytecode,
                        const uint32_t* bytecode_acc, uint32_t log_bytecode, const uint32_t* tables_desc, uint32_t n_tables,
                        const uint32_t* tables_cols, const uint32_t* c5, const uint32_t* alphas16, uint32_t* out_nums,
                        uint32_t* out_dens) {
    std::vector<VmTableTrace> tabs;
    size_t off = 0;
    for (uint32_t i = 0; i < n_tables; i++) {
        VmTableTrace t{(int)tables_desc[2 * i], tables_desc[2 * i + 1], tables_cols + off};
        off += vm_table_def(t.table).n_columns_total << t.log_rows;
        tabs.push_back(t);
    }
    EF c;
    std::memcpy(c.v, c5, 20);
    std::vector<uint32_t> nums;
    std::vector<EF> dens;
    size_t total = logup_fill(memory, memory_acc, log_mem, bytecode, bytecode_acc, log_bytecode, tabs, c, (const EF*)alphas16, nums, dens);
    if (out_nums) std::memcpy(out_nums, nums.data(), nums.size() * 4);
    if (out_dens) std::memcpy(out_dens, dens.data(), dens.size() * 20);
    return total;
}
}

// ================================================================================================
// prove_execution / verify_execution after witness generation (execution_oracle.hpp)
// ================================================================================================
#include "execution_oracle.hpp"
#include "pruning_oracle.hpp"
extern "C" {
// hdr = [log_inv_rate, log_memory, log_bytecode, ending_pc, public_memory_size, n_public_input, log_rows x3]
// builder: 8 words as in make_builder, or NULL for default_whir_config(log_inv_rate)
// tables[t]: n_columns_total x 2^log_rows[t] words, column major.  Returns proof words (orc_last_proof), 0 on failure.
uint64_t orc_prove_execution(const uint32_t* hdr, const uint32_t* builder, const uint32_t* bytecode_hash, const uint32_t* public_input,
                             const uint32_t* bytecode, const uint32_t* bytecode_acc, const uint32_t* memory, const uint32_t* memory_acc,
                             const uint32_t* t_exec, const uint32_t* t_ext, const uint32_t* t_pos) {
    try {
        ExecutionInput in;
        in.log_inv_rate = hdr[0];
        in.log_memory = hdr[1];
        in.log_bytecode = hdr[2];
        in.ending_pc = hdr[3];
        in.public_memory_size = hdr[4];
        in.public_input.assign(public_input, public_input + hdr[5]);
        std::memcpy(in.bytecode_hash, bytecode_hash, 32);
        in.bytecode = bytecode;
        in.bytecode_acc = bytecode_acc;
        in.memory = memory;
        in.memory_acc = memory_acc;
        const uint32_t* tp[3] = {t_exec, t_ext, t_pos};
        for (int t = 0; t < 3; t++) in.tables[t] = VmTableTrace{t, hdr[6 + t], tp[t]};
        WhirConfigBuilder b = builder ? make_builder(builder) : default_whir_config(in.log_inv_rate);
        ProverState ps;
        prove_execution(ps, in, b);
        g_last_proof = serialize_proof(ps);
        g_verr[0] = 0;
        return g_last_proof.size();
    } catch (const std::exception& e) {
        snprintf(g_verr, sizeof g_verr, "%s", e.what());
        return 0;
    }
}
int orc_verify_execution(const uint32_t* proof_blob, const uint32_t* builder, const uint32_t* bytecode_hash, const uint32_t* public_input,
                         uint32_t n_public_input, const uint32_t* bytecode, uint32_t log_bytecode, uint32_t ending_pc) {
    try {
        VerifierState vs;
        parse_proof(proof_blob, vs);
        std::vector<uint32_t> pi(public_input, public_input + n_public_input);
        WhirConfigBuilder b;
        if (builder) b = make_builder(builder);
        verify_execution(vs, pi, bytecode_hash, bytecode, log_bytecode, ending_pc, builder ? &b : nullptr);
        g_verr[0] = 0;
        return 1;
    } catch (const std::exception& e) {
        snprintf(g_verr, sizeof g_verr, "%s", e.what());
        return 0;
    }
}

// ---- Merkle-path pruning (pruning_oracle.hpp); results through orc_last_proof ------------------------------------------
uint64_t orc_prune_proof(const uint32_t* blob, const uint32_t* batch_sizes, uint32_t n_batches) {
    try {
        g_last_proof = prune_blob(blob, batch_sizes, n_batches);
        g_verr[0] = 0;
        return g_last_proof.size();
    } catch (const std::exception& e) {
        snprintf(g_verr, sizeof g_verr, "%s", e.what());
        return 0;
    }
}
uint64_t orc_restore_proof(const uint32_t* pruned, uint64_t n_words) {
    try {
        g_last_proof = restore_blob(pruned, n_words);
        g_verr[0] = 0;
        return g_last_proof.size();
    } catch (const std::exception& e) {
        snprintf(g_verr, sizeof g_verr, "%s", e.what());
        return 0;
    }
}
uint64_t orc_pruned_size_fe(const uint32_t* pruned, uint64_t n_words) {
    try {
        g_verr[0] = 0;
        return pruned_size_fe(pruned, n_words);
    } catch (const std::exception& e) {
        snprintf(g_verr, sizeof g_verr, "%s", e.what());
        return 0;
    }
}
// OpenMP width of the oracle's data-parallel loops (bench.py's cpu_baseline leg); results do not depend on it
int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
}

// ================================================================================================
// leanVM runner + get_execution_trace (vm_oracle.hpp)
// ================================================================================================
#include "vm_oracle.hpp"
namespace {
struct OrcVmHint {  // layout of lm_vm_hint (include/leanmultisig_host.h)
    uint32_t pc, kind, args[4];
    uint8_t mode[4];
};
struct OrcVmRun {
    orc::vm::Bytecode bc;
    orc::vm::ExecutionResult er;
    orc::vm::ExecutionTrace tr;
    std::vector<uint32_t> pcs, fps;
    bool have_trace = false;
};
std::string g_vm_err;
}  // namespace
extern "C" {
const char* orc_vm_last_error() { return g_vm_err.c_str(); }
// Runs the program; returns a handle (NULL + orc_vm_last_error on a RunnerError).  Inputs as lmh_bytecode_new / lm_vm_witness.
void* orc_vm_execute(const uint32_t* instructions_multilinear, uint32_t log_size, uint64_t n_instructions, uint32_t ending_pc,
                     uint32_t starting_frame_memory, const void* hints_raw, uint64_t n_hints, uint32_t n_names, const uint32_t* public_input,
                     uint32_t n_public_input, uint32_t preamble_memory_len, const uint64_t* name_entry_begin, const uint64_t* entry_offset,
                     const uint32_t* data) {
    using namespace orc::vm;
    auto* run = new OrcVmRun();
    try {
        Bytecode& bc = run->bc;
        bc.instructions_multilinear = instructions_multilinear;
        bc.log_size = log_size, bc.ending_pc = ending_pc, bc.starting_frame_memory = starting_frame_memory, bc.n_names = n_names;
        for (uint64_t pc = 0; pc < n_instructions; pc++) bc.code.push_back(decode(instructions_multilinear + 16 * pc));
        bc.hints.resize(n_instructions);
        const OrcVmHint* hs = (const OrcVmHint*)hints_raw;
        for (uint64_t h = 0; h < n_hints; h++) {
            Hint x;
            x.kind = hs[h].kind;
            std::memcpy(x.args, hs[h].args, 16);
            std::memcpy(x.mode, hs[h].mode, 4);
            bc.hints.at(hs[h].pc).push_back(x);
        }
        WitnessHints w{preamble_memory_len, name_entry_begin, entry_offset, data};
        run->er = execute_bytecode(bc, public_input, n_public_input, w);
        for (size_t x : run->er.trace.pcs) run->pcs.push_back((uint32_t)x);
        for (size_t x : run->er.trace.fps) run->fps.push_back((uint32_t)x);
    } catch (const std::exception& e) {
        g_vm_err = e.what();
        delete run;
        return nullptr;
    }
    return run;
}
void orc_vm_free(void* h) { delete (OrcVmRun*)h; }
// sizes: [n_cycles, memory_len, n_poseidon, n_extension_rows, public_memory_size, runtime_memory_size, add, mul, deref, jump]
void orc_vm_sizes(void* h, uint64_t* out) {
    auto* r = (OrcVmRun*)h;
    out[0] = r->pcs.size(), out[1] = r->er.memory.cells.size(), out[2] = r->er.trace.poseidon[0].size(), out[3] = r->er.trace.extension[0].size();
    out[4] = r->er.public_memory_size, out[5] = r->er.runtime_memory_size;
    out[6] = r->er.trace.add, out[7] = r->er.trace.mul, out[8] = r->er.trace.deref, out[9] = r->er.trace.jump;
}
void orc_vm_log(void* h, uint32_t* pcs, uint32_t* fps, uint32_t* memory, uint8_t* defined) {
    auto* r = (OrcVmRun*)h;
    std::memcpy(pcs, r->pcs.data(), r->pcs.size() * 4);
    std::memcpy(fps, r->fps.data(), r->fps.size() * 4);
    for (size_t i = 0; i < r->er.memory.cells.size(); i++) {
        defined[i] = r->er.memory.cells[i].has_value();
        memory[i] = r->er.memory.cells[i].value_or(0);
    }
}
// get_execution_trace: sizes out = [log_memory, log_rows x 3, non_padded x 3, zero_vec_ptr, null_hash_ptr]
void orc_vm_trace(void* h, uint64_t* out) {
    auto* r = (OrcVmRun*)h;
    if (!r->have_trace) {
        r->tr = orc::vm::get_execution_trace(r->bc, r->er);
        r->have_trace = true;
    }
    size_t lm = 0;
    while (((size_t)1 << lm) < r->tr.memory.size()) lm++;
    out[0] = lm;
    for (int t = 0; t < 3; t++) out[1 + t] = r->tr.log_n_rows[t], out[4 + t] = r->tr.non_padded_n_rows[t];
    out[7] = r->tr.zero_vec_ptr, out[8] = r->tr.null_hash_ptr;
}
void orc_vm_trace_memory(void* h, uint32_t* out) {
    auto* r = (OrcVmRun*)h;
    std::memcpy(out, r->tr.memory.data(), r->tr.memory.size() * 4);
}
// table t as n_columns_total x 2^log_rows, column-major
void orc_vm_trace_table(void* h, uint32_t t, uint32_t* out) {
    auto* r = (OrcVmRun*)h;
    const auto& cols = r->tr.tables[t];
    const size_t n = (size_t)1 << r->tr.log_n_rows[t];
    for (size_t c = 0; c < cols.size(); c++) std::memcpy(out + c * n, cols[c].data(), n * 4);
}
}

// ================================================================================================
// transcript primitives and the product sumcheck with fixed challenges, for the second pin (tests/golden/twin_r03.py)
// ================================================================================================
extern "C" {
void* orc_ps_new() { return new orc::ProverState(); }
void orc_ps_free(void* h) { delete (orc::ProverState*)h; }
void orc_ps_add_base(void* h, const uint32_t* s, uint64_t n) { ((orc::ProverState*)h)->add_base_scalars(s, n); }
void orc_ps_duplex(void* h) { ((orc::ProverState*)h)->duplex(); }
void orc_ps_sample_vec(void* h, uint64_t n, uint32_t* out) {
    auto v = ((orc::ProverState*)h)->sample_vec(n);
    for (uint64_t i = 0; i < n; i++) std::memcpy(out + 5 * i, v[i].v, 20);
}
void orc_ps_sample_in_range(void* h, uint32_t bits, uint64_t n, uint64_t* out) {
    auto v = ((orc::ProverState*)h)->sample_in_range(bits, n);
    for (uint64_t i = 0; i < n; i++) out[i] = v[i];
}
void orc_ps_add_sumcheck_polynomial(void* h, const uint32_t* coeffs, uint32_t n, const uint32_t* eq_alpha) {
    std::vector<orc::EF> c(n);
    for (uint32_t i = 0; i < n; i++) std::memcpy(c[i].v, coeffs + 5 * i, 20);
    orc::EF a;
    if (eq_alpha) std::memcpy(a.v, eq_alpha, 20);
    ((orc::ProverState*)h)->add_sumcheck_polynomial(c, eq_alpha ? &a : nullptr);
}
void orc_ps_pow_grinding(void* h, uint32_t bits) { ((orc::ProverState*)h)->pow_grinding(bits); }
void orc_ps_state(void* h, uint32_t* out16) { std::memcpy(out16, ((orc::ProverState*)h)->ch.state, 64); }
uint64_t orc_ps_transcript(void* h, uint32_t* out) {
    auto& t = ((orc::ProverState*)h)->transcript;
    if (out) std::memcpy(out, t.data(), t.size() * 4);
    return t.size();
}
// run_product_sumcheck's rounds (product_computation.rs:37-315) with GIVEN challenges: out = n_rounds x (c0, c1, c2); the folded
// tables are returned in f_out / w_out (2^(n_vars - n_rounds) EF each)
void orc_product_sumcheck_fixed(const uint32_t* f_base, const uint32_t* w_ef, uint32_t n_vars, const uint32_t* challenges, uint32_t n_rounds,
                                uint32_t* out, uint32_t* f_out, uint32_t* w_out) {
    using namespace orc;
    size_t n = (size_t)1 << n_vars;
    std::vector<EF> f(n), w(n);
    EF sum = ef_zero();
    for (size_t i = 0; i < n; i++) {
        f[i] = ef_from_base(f_base[i]);
        std::memcpy(w[i].v, w_ef + 5 * i, 20);
        sum = ef_add(sum, ef_mul(f[i], w[i]));
    }
    for (uint32_t r = 0; r < n_rounds; r++) {
        const size_t half = f.size() / 2;
        EF c0 = ef_zero(), c2 = ef_zero();
        for (size_t i = 0; i < half; i++) {
            c0 = ef_add(c0, ef_mul(f[i], w[i]));
            c2 = ef_add(c2, ef_mul(ef_sub(f[i + half], f[i]), ef_sub(w[i + half], w[i])));
        }
        const EF c1 = ef_sub(ef_sub(sum, ef_add(c0, c0)), c2);
        std::memcpy(out + 15 * r, c0.v, 20);
        std::memcpy(out + 15 * r + 5, c1.v, 20);
        std::memcpy(out + 15 * r + 10, c2.v, 20);
        EF ch;
        std::memcpy(ch.v, challenges + 5 * r, 20);
        sum = ef_add(c0, ef_mul(ch, ef_add(c1, ef_mul(ch, c2))));
        for (size_t i = 0; i < half; i++) {
            f[i] = ef_add(f[i], ef_mul(ch, ef_sub(f[i + half], f[i])));
            w[i] = ef_add(w[i], ef_mul(ch, ef_sub(w[i + half], w[i])));
        }
        f.resize(half);
        w.resize(half);
    }
    for (size_t i = 0; i < f.size(); i++) {
        std::memcpy(f_out + 5 * i, f[i].v, 20);
        std::memcpy(w_out + 5 * i, w[i].v, 20);
    }
}
}
