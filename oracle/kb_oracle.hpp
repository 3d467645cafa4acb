// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Scalar CPU restatement of the leanMultisig proving hot path (reference checkout: /root/reference,
// citations below are relative to it).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, load or call anything under oracle/.  The product
// (leanmultisig_amd/) never includes or links this code.
//
// Parity pinning: Poseidon1-16 is pinned by the reference's own known-answer test
// (crates/backend/koala-bear/src/poseidon1_koalabear_16.rs:1066-1092); the field / extension / DFT
// parameters are pinned by the reference's in-source constants (koala_bear.rs:22-64,
// quintic_extension/mod.rs:19-50) and by the algebraic identities the reference's own tests
// use (whir/src/dft.rs:583-603: DFT row i == MLE at expand_from_univariate(g^i)).  Everything
// else on the path has NO stored vectors in the reference ("parity unpinned" beyond those —
// see DESIGN.md); it is written here in the most literal textbook form so it is an
// independent check of the optimised device code.
//
// Representation (F7 in SURVEY.md): every base-field value is a u32 in Montgomery form, R = 2^32
// (monty_31/monty_31.rs:33-41); EF = 5 consecutive u32 (quintic_extension/extension.rs:25-35).
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <vector>
#include <array>
#include <cassert>

namespace orc {

// ---------------------------------------------------------------------------------------------
// Base field: KoalaBear p = 2^31 - 2^24 + 1 (koala_bear.rs:22-26)
// ---------------------------------------------------------------------------------------------
static const uint32_t P = 0x7f000001u;
static const uint32_t MONTY_MU = 0x81000001u;  // +p^{-1} mod 2^32

// monty_31/utils.rs:107-127
static inline uint32_t monty_reduce(uint64_t x) {
    uint64_t t = (x * (uint64_t)MONTY_MU) & 0xffffffffull;
    uint64_t u = t * (uint64_t)P;
    uint64_t d = x - u;
    uint32_t hi = (uint32_t)(d >> 32);
    return (x < u) ? hi + P : hi;
}
// monty_31/utils.rs:8-10
static inline uint32_t to_monty(uint32_t x) { return (uint32_t)((((uint64_t)x) << 32) % P); }
static inline uint32_t from_monty(uint32_t x) { return monty_reduce((uint64_t)x); }
// monty_31/utils.rs:65-90
static inline uint32_t add(uint32_t a, uint32_t b) {
    uint32_t s = a + b;
    return s >= P ? s - P : s;
}
static inline uint32_t sub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + P - b; }
static inline uint32_t neg(uint32_t a) { return a ? P - a : 0; }
static inline uint32_t mul(uint32_t a, uint32_t b) { return monty_reduce((uint64_t)a * b); }
static inline uint32_t dbl(uint32_t a) { return add(a, a); }
static const uint32_t ONE = 0x01fffffeu;  // to_monty(1) = 2^32 mod p

static inline uint32_t pow_u64(uint32_t a, uint64_t e) {
    uint32_t r = ONE;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = mul(a, a);
        e >>= 1;
    }
    return r;
}
static inline uint32_t inv(uint32_t a) { return pow_u64(a, (uint64_t)P - 2); }

// koala_bear.rs:50-54 — canonical two-adic generators, index = bits.
static const uint32_t TWO_ADIC_GENERATORS[25] = {
    0x1,        0x7f000000, 0x7e010002, 0x6832fe4a, 0x8dbd69c,  0xa28f031,  0x5c4a5b99, 0x29b75a80, 0x17668b8a,
    0x27ad539b, 0x334d48c7, 0x7744959c, 0x768fc6fa, 0x303964b2, 0x3e687d4d, 0x45a60e61, 0x6e2f4d7a, 0x163bd499,
    0x6c4a8a45, 0x143ef899, 0x514ddcad, 0x484ef19b, 0x205d63c3, 0x68e7dd49, 0x6ac49f88};
static inline uint32_t two_adic_generator(unsigned bits) { return to_monty(TWO_ADIC_GENERATORS[bits]); }

// ---------------------------------------------------------------------------------------------
// Quintic extension F_p[X]/(X^5 + X^2 - 1)  (quintic_extension/extension.rs:531-548)
// ---------------------------------------------------------------------------------------------
struct EF {
    uint32_t v[5];
};
static inline EF ef_zero() { return EF{{0, 0, 0, 0, 0}}; }
static inline EF ef_one() { return EF{{ONE, 0, 0, 0, 0}}; }
static inline EF ef_from_base(uint32_t a) { return EF{{a, 0, 0, 0, 0}}; }
static inline bool ef_eq(const EF& a, const EF& b) { return std::memcmp(a.v, b.v, 20) == 0; }
static inline EF ef_add(const EF& a, const EF& b) {
    EF r;
    for (int i = 0; i < 5; i++) r.v[i] = add(a.v[i], b.v[i]);
    return r;
}
// OpenMP sums of extension-field elements (exact arithmetic: the order of a sum does not matter)
#pragma omp declare reduction(efsum : EF : omp_out = ef_add(omp_out, omp_in)) initializer(omp_priv = ef_zero())
static inline EF ef_sub(const EF& a, const EF& b) {
    EF r;
    for (int i = 0; i < 5; i++) r.v[i] = sub(a.v[i], b.v[i]);
    return r;
}
static inline EF ef_neg(const EF& a) {
    EF r;
    for (int i = 0; i < 5; i++) r.v[i] = neg(a.v[i]);
    return r;
}
static inline EF ef_mul_base(const EF& a, uint32_t b) {
    EF r;
    for (int i = 0; i < 5; i++) r.v[i] = mul(a.v[i], b);
    return r;
}
// Schoolbook product then reduction with X^5 = 1 - X^2 (i.e. X^5 + X^2 - 1 = 0):
//   X^5 = 1 - X^2, X^6 = X - X^3, X^7 = X^2 - X^4, X^8 = X^3 - X^5 = X^3 - 1 + X^2.
static inline EF ef_mul(const EF& a, const EF& b) {
    uint32_t c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) c[i + j] = add(c[i + j], mul(a.v[i], b.v[j]));
    // fold degree 8 down to 5
    // X^8 = X^3 + X^2 - 1
    c[3] = add(c[3], c[8]);
    c[2] = add(c[2], c[8]);
    c[0] = sub(c[0], c[8]);
    // X^7 = X^2 - X^4
    c[2] = add(c[2], c[7]);
    c[4] = sub(c[4], c[7]);
    // X^6 = X - X^3
    c[1] = add(c[1], c[6]);
    c[3] = sub(c[3], c[6]);
    // X^5 = 1 - X^2
    c[0] = add(c[0], c[5]);
    c[2] = sub(c[2], c[5]);
    return EF{{c[0], c[1], c[2], c[3], c[4]}};
}
static inline EF ef_square(const EF& a) { return ef_mul(a, a); }
static inline EF ef_pow_u64(EF a, uint64_t e) {
    EF r = ef_one();
    while (e) {
        if (e & 1) r = ef_mul(r, a);
        a = ef_square(a);
        e >>= 1;
    }
    return r;
}
// Inverse: a^{-1} = a^{p^5 - 2}.  p^5 - 2 does not fit in 64 bits, so use
// a^{-1} = conj(a) / Norm(a), conj(a) = a^{p + p^2 + p^3 + p^4} computed with plain exponentiation by p
// (the reference uses its Frobenius matrix, quintic_extension/extension.rs:585-607; the value is unique).
static inline EF ef_frobenius(const EF& a) { return ef_pow_u64(a, (uint64_t)P); }
static inline EF ef_inv(const EF& a) {
    EF f1 = ef_frobenius(a);
    EF f2 = ef_frobenius(f1);
    EF f3 = ef_frobenius(f2);
    EF f4 = ef_frobenius(f3);
    EF conj = ef_mul(ef_mul(f1, f2), ef_mul(f3, f4));
    EF norm = ef_mul(a, conj);
    assert(norm.v[1] == 0 && norm.v[2] == 0 && norm.v[3] == 0 && norm.v[4] == 0);
    return ef_mul_base(conj, inv(norm.v[0]));
}

// ---------------------------------------------------------------------------------------------
// Poseidon1-16, textbook schedule (poseidon1_koalabear_16.rs:11-15,22,580-581,699-815):
//   4 full rounds, 20 partial rounds, 4 full rounds; S-box x^3;
//   full round    = add RC (16 lanes), cube all lanes, dense circulant MDS
//   partial round = add RC (16 lanes), cube lane 0,    dense circulant MDS
// The reference evaluates the partial rounds through a sparse factorisation (:399-480, :873-912);
// that is an optimisation of this schedule and must reproduce the same KAT (:1083-1091).
// ---------------------------------------------------------------------------------------------
static const uint32_t MDS_CIRC_COL[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
static const uint32_t POSEIDON1_RC_CANON[28][16] = {
#include "params/poseidon1_rc.inc"
};

struct PoseidonTables {
    uint32_t rc[28][16];   // Montgomery form
    uint32_t mds[16][16];  // Montgomery form, M[i][j] = col[(16 + i - j) % 16]
    PoseidonTables() {
        for (int r = 0; r < 28; r++)
            for (int i = 0; i < 16; i++) rc[r][i] = to_monty(POSEIDON1_RC_CANON[r][i]);
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) mds[i][j] = to_monty(MDS_CIRC_COL[(16 + i - j) % 16]);
    }
};
static inline const PoseidonTables& poseidon_tables() {
    static PoseidonTables t;
    return t;
}

static inline void mds_dense(uint32_t s[16]) {
    const PoseidonTables& t = poseidon_tables();
    uint32_t o[16];
    for (int i = 0; i < 16; i++) {
        uint32_t acc = 0;
        for (int j = 0; j < 16; j++) acc = add(acc, mul(t.mds[i][j], s[j]));
        o[i] = acc;
    }
    std::memcpy(s, o, sizeof o);
}
static inline uint32_t cube(uint32_t x) { return mul(mul(x, x), x); }

static inline void poseidon16_permute(uint32_t s[16]) {
    const PoseidonTables& t = poseidon_tables();
    int r = 0;
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < 16; i++) s[i] = cube(add(s[i], t.rc[r][i]));
        mds_dense(s);
    }
    for (int k = 0; k < 20; k++, r++) {
        for (int i = 0; i < 16; i++) s[i] = add(s[i], t.rc[r][i]);
        s[0] = cube(s[0]);
        mds_dense(s);
    }
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < 16; i++) s[i] = cube(add(s[i], t.rc[r][i]));
        mds_dense(s);
    }
}
// compression mode: out = permute(x) + x  (poseidon1_koalabear_16.rs:1018-1030)
static inline void poseidon16_compress(uint32_t s[16]) {
    uint32_t in[16];
    std::memcpy(in, s, sizeof in);
    poseidon16_permute(s);
    for (int i = 0; i < 16; i++) s[i] = add(s[i], in[i]);
}

// 2-to-1 compression of two 8-word digests (symetric/src/compression.rs:5-15)
static inline void compress_pair(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]) {
    uint32_t s[16];
    std::memcpy(s, l, 32);
    std::memcpy(s + 8, r, 32);
    poseidon16_compress(s);
    std::memcpy(out, s, 32);
}

// Right-to-left overwrite-mode sponge (symetric/src/sponge.rs:7-24 `hash_slice`): data length is a
// multiple of 8 and >= 16.
static inline void hash_slice(const uint32_t* data, size_t len, uint32_t out[8]) {
    assert(len % 8 == 0 && len >= 16);
    size_t n_chunks = len / 8;
    uint32_t s[16];
    std::memcpy(s, data + len - 16, 64);
    poseidon16_compress(s);
    for (size_t c = n_chunks - 2; c-- > 0;) {
        std::memcpy(s + 8, data + c * 8, 32);
        poseidon16_compress(s);
    }
    std::memcpy(out, s, 32);
}

// ---------------------------------------------------------------------------------------------
// Merkle tree: leaf digests = hash_slice(row zero-padded to full width); levels = compress_pair
// (whir/src/merkle.rs:59-88,215-287 ; symetric/src/merkle.rs:21-90).  The reference's
// zero-suffix-state shortcut (sponge.rs:27-49) is an optimisation of exactly this.
// ---------------------------------------------------------------------------------------------
struct MerkleTree {
    size_t height = 0;       // number of leaves (power of two)
    size_t full_width = 0;   // leaf width in base words after zero padding
    std::vector<uint32_t> leaves;                 // row-major height x full_width
    std::vector<std::vector<uint32_t>> layers;    // layers[0] = leaf digests (height*8), ..., last = root (8)
    const uint32_t* root() const { return layers.back().data(); }
};
static inline MerkleTree merkle_build(const uint32_t* rows, size_t height, size_t width, size_t full_width) {
    MerkleTree t;
    t.height = height;
    t.full_width = full_width;
    t.leaves.assign(height * full_width, 0);
    for (size_t r = 0; r < height; r++) std::memcpy(&t.leaves[r * full_width], rows + r * width, width * 4);
    std::vector<uint32_t> lay(height * 8);
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < height; r++) hash_slice(&t.leaves[r * full_width], full_width, &lay[r * 8]);
    t.layers.push_back(lay);
    while (t.layers.back().size() > 8) {
        const std::vector<uint32_t>& prev = t.layers.back();
        size_t n = prev.size() / 16;
        std::vector<uint32_t> next(n * 8);
#pragma omp parallel for schedule(static) if (n >= 256)
        for (size_t i = 0; i < n; i++) compress_pair(&prev[16 * i], &prev[16 * i + 8], &next[8 * i]);
        t.layers.push_back(next);
    }
    return t;
}
// (whir/src/merkle.rs:205-211, symetric/src/merkle.rs:43-47): leaf row + siblings bottom-up.
static inline void merkle_open(const MerkleTree& t, size_t index, uint32_t* leaf_out, uint32_t* siblings_out) {
    std::memcpy(leaf_out, &t.leaves[index * t.full_width], t.full_width * 4);
    size_t log_h = t.layers.size() - 1;
    for (size_t i = 0; i < log_h; i++) std::memcpy(siblings_out + 8 * i, &t.layers[i][(((index >> i) ^ 1)) * 8], 32);
}
// symetric/src/merkle.rs:92-121
static inline bool merkle_verify(const uint32_t root[8], size_t log_height, size_t index, const uint32_t* leaf,
                                 size_t leaf_len, const uint32_t* siblings) {
    uint32_t cur[8];
    hash_slice(leaf, leaf_len, cur);
    for (size_t i = 0; i < log_height; i++) {
        uint32_t nxt[8];
        const uint32_t* sib = siblings + 8 * i;
        if ((index & 1) == 0)
            compress_pair(cur, sib, nxt);
        else
            compress_pair(sib, cur, nxt);
        std::memcpy(cur, nxt, 32);
        index >>= 1;
    }
    return std::memcmp(cur, root, 32) == 0;
}

// ---------------------------------------------------------------------------------------------
// LDE: gather/replicate (whir/src/utils.rs:128-150) + evaluation-domain radix-2 DFT
// (whir/src/dft.rs:79-144, butterflies :548-568).  Row-major h x w matrix of base words.
// ---------------------------------------------------------------------------------------------
// matrix[r][c] = evals[((c << log_block) + r) >> log_inv_rate], block = len*2^rate / 2^fold
template <typename T>
static inline std::vector<T> prepare_evals_for_fft(const T* evals, size_t len, unsigned folding_factor,
                                                   unsigned log_inv_rate, size_t dft_n_cols) {
    size_t n_blocks = (size_t)1 << folding_factor;
    size_t full_len = len << log_inv_rate;
    size_t block = full_len / n_blocks;
    unsigned log_block = 0;
    while (((size_t)1 << log_block) < block) log_block++;
    std::vector<T> out(block * dft_n_cols);
    for (size_t i = 0; i < out.size(); i++) {
        size_t c = i % dft_n_cols, r = i / dft_n_cols;
        out[i] = evals[((c << log_block) + r) >> log_inv_rate];
    }
    return out;
}
// In-place, natural order in and out, per column.  Layer l = 1..log h: blocks of 2^l rows,
// j < 2^(l-1): a = v[j], b = v[j + 2^(l-1)], d = (b - a) * w^j with w the 2^l-th root generator,
// v[j] = a + d, v[j + 2^(l-1)] = a - d.
static inline void dft_batch_by_evals(uint32_t* mat, size_t h, size_t w) {
    unsigned log_h = 0;
    while (((size_t)1 << log_h) < h) log_h++;
    for (unsigned l = 1; l <= log_h; l++) {
        size_t half = (size_t)1 << (l - 1);
        uint32_t g = two_adic_generator(l);
        std::vector<uint32_t> tw(half);
        tw[0] = ONE;
        for (size_t j = 1; j < half; j++) tw[j] = mul(tw[j - 1], g);
        const size_t n_bfly = h / 2;  // butterfly rows of this layer: index -> (block, j)
#pragma omp parallel for schedule(static) if (n_bfly * w >= 4096)
        for (size_t bj = 0; bj < n_bfly; bj++) {
            {
                const size_t blk = (bj / half) * 2 * half, j = bj % half;
                uint32_t* ra = mat + (blk + j) * w;
                uint32_t* rb = mat + (blk + j + half) * w;
                for (size_t c = 0; c < w; c++) {
                    uint32_t a = ra[c], b = rb[c];
                    uint32_t d = mul(sub(b, a), tw[j]);
                    ra[c] = add(a, d);
                    rb[c] = sub(a, d);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Multilinear helpers (poly/src/point.rs:51-61, evals.rs:142-347, eq_mle.rs:85-147)
// point[0] <-> most significant index bit.
// ---------------------------------------------------------------------------------------------
static inline std::vector<EF> expand_from_univariate(EF a, size_t n) {
    std::vector<EF> r(n);
    for (size_t i = 0; i < n; i++) {
        r[i] = a;
        a = ef_square(a);
    }
    return r;
}
// eq table: out[i] = scalar * prod_j (i_j p_j + (1-i_j)(1-p_j))
static inline std::vector<EF> eq_table(const EF* point, size_t n, EF scalar) {
    std::vector<EF> t(1, scalar);
    for (size_t k = 0; k < n; k++) {
        std::vector<EF> nt(t.size() * 2);
        for (size_t i = 0; i < t.size(); i++) {
            EF hi = ef_mul(t[i], point[k]);
            nt[2 * i] = ef_sub(t[i], hi);
            nt[2 * i + 1] = hi;
        }
        t.swap(nt);
    }
    return t;
}
// MLE evaluation by successive folding of the MOST significant variable first.
static inline EF mle_eval_base(const uint32_t* v, size_t n_vars, const EF* point) {
    size_t len = (size_t)1 << n_vars;
    std::vector<EF> cur(len);
    for (size_t i = 0; i < len; i++) cur[i] = ef_from_base(v[i]);
    for (size_t k = 0; k < n_vars; k++) {
        size_t half = len >> 1;
        for (size_t i = 0; i < half; i++) cur[i] = ef_add(cur[i], ef_mul(point[k], ef_sub(cur[i + half], cur[i])));
        len = half;
    }
    return cur[0];
}
static inline EF mle_eval_ext(const EF* v, size_t n_vars, const EF* point) {
    size_t len = (size_t)1 << n_vars;
    std::vector<EF> cur(v, v + len);
    for (size_t k = 0; k < n_vars; k++) {
        size_t half = len >> 1;
        for (size_t i = 0; i < half; i++) cur[i] = ef_add(cur[i], ef_mul(point[k], ef_sub(cur[i + half], cur[i])));
        len = half;
    }
    return cur[0];
}

// ---------------------------------------------------------------------------------------------
// Duplex challenger (fiat-shamir/src/challenger.rs:9-76) — overwrite mode, plain permutation.
// ---------------------------------------------------------------------------------------------
struct Challenger {
    uint32_t state[16];
    bool rate_fresh = false;
    Challenger() { std::memset(state, 0, sizeof state); }
    void observe(const uint32_t v[8]) {
        std::memcpy(state + 8, v, 32);
        poseidon16_permute(state);
        rate_fresh = true;
    }
    void observe_many(const uint32_t* s, size_t n) {
        for (size_t off = 0; off < n; off += 8) {
            uint32_t buf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            size_t k = n - off < 8 ? n - off : 8;
            std::memcpy(buf, s + off, k * 4);
            observe(buf);
        }
    }
    void duplex() {
        uint32_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        observe(z);
    }
    void sample(uint32_t out[8]) {
        assert(rate_fresh);
        std::memcpy(out, state + 8, 32);
        rate_fresh = false;
    }
    std::vector<uint32_t> sample_many(size_t n) {  // n rate blocks
        std::vector<uint32_t> out;
        for (size_t i = 0; i < n; i++) {
            if (i) duplex();
            uint32_t b[8];
            sample(b);
            out.insert(out.end(), b, b + 8);
        }
        return out;
    }
    // fiat-shamir/src/utils.rs:43-58
    std::vector<EF> sample_vec(size_t len) {
        std::vector<uint32_t> fe = sample_many((len * 5 + 7) / 8);
        std::vector<EF> r(len);
        for (size_t i = 0; i < len; i++) std::memcpy(r[i].v, &fe[5 * i], 20);
        return r;
    }
    EF sample_ef() { return sample_vec(1)[0]; }
    // challenger.rs:66-75
    std::vector<size_t> sample_in_range(unsigned bits, size_t n) {
        std::vector<uint32_t> fe = sample_many((n + 7) / 8);
        std::vector<size_t> r(n);
        for (size_t i = 0; i < n; i++) r[i] = (size_t)from_monty(fe[i]) & (((size_t)1 << bits) - 1);
        return r;
    }
};

}  // namespace orc
