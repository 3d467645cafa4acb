// ORACLE — TEST INFRASTRUCTURE ONLY (see kb_oracle.hpp header).  C entry points for ctypes.
#include "kb_oracle.hpp"
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
