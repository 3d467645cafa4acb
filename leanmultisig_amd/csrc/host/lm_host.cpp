// Host-side mirror of the reference's transcript and WHIR driver (see include/leanmultisig_host.h).
// Everything heavy goes through the device ABI (include/leanmultisig.h); this file only sequences the protocol:
// Fiat–Shamir state, round polynomials, query sampling, tiny leaf evaluations.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <vector>
#include <hip/hip_runtime.h>
#include "lm_host_internal.h"

using kb::EF;
using kb::u32;
using kb::u64;
using lmh::Challenger;
using lmh::Opening;
using lmh::ColVal;
using lmh::VmTableDef;
using lmh::kVmTables;
using lmh::kSnarkDomainSep;
using lmh::log2_ceil_u64;

namespace {

// Wall clock per stage of prove_execution.  Always recorded into the prover object (lmh_prover_stage_times: no synchronisation, the
// boundaries are where the host holds the stage's result); LM_STAGE_TIMES=1 additionally synchronises the stream at every mark and
// prints the stage on stderr (the total is then slightly pessimistic).
struct StageClock {
    const char* prefix = "";
    lm_ctx* ctx;
    bool on;
    double* sink = nullptr;  // LMH_N_STAGES slots of the prover (top-level clock only)
    std::chrono::steady_clock::time_point t0;
    explicit StageClock(lm_ctx* c, double* stage_ms = nullptr) : ctx(c), on(getenv("LM_STAGE_TIMES") != nullptr), sink(stage_ms) {
        if (sink)
            for (int i = 0; i < LMH_N_STAGES; i++) sink[i] = 0;
        if (on) lm_sync(ctx);
        t0 = std::chrono::steady_clock::now();
    }
    void mark(const char* name, int slot = -1) {
        if (!on && !sink) return;
        if (on) lm_sync(ctx);
        const auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (sink && slot >= 0 && slot < LMH_N_STAGES) sink[slot] += ms;
        if (on) fprintf(stderr, "# stage %s%-28s %8.3f ms\n", prefix, name, ms);
        t0 = t1;
    }
};


EF ef_load(const u32* p) {
    EF r;
    memcpy(r.v, p, 20);
    return r;
}

std::vector<EF> expand_from_univariate(EF a, u32 n) {  // poly/src/point.rs:51-61
    std::vector<EF> r(n);
    for (u32 i = 0; i < n; i++) {
        r[i] = a;
        a = kb::ef_sqr(a);
    }
    return r;
}

// evals (2^k values, base or AoS EF) at an EF point, point[0] <-> MSB (poly/src/evals.rs:142-347)
EF eval_leaf(const u32* leaf, bool is_ext, u32 k, const EF* point) {
    u64 len = 1ull << k;
    std::vector<EF> cur(len);
    for (u64 i = 0; i < len; i++) cur[i] = is_ext ? ef_load(leaf + 5 * i) : kb::ef_from_base(leaf[i]);
    for (u32 j = 0; j < k; j++) {
        u64 half = len >> 1;
        for (u64 i = 0; i < half; i++) cur[i] = kb::ef_add(cur[i], kb::ef_mul(point[j], kb::ef_sub(cur[i + half], cur[i])));
        len = half;
    }
    return cur[0];
}

u32 ilog2(u64 x) {
    u32 l = 0;
    while ((1ull << (l + 1)) <= x) l++;
    return l;
}

}  // namespace

struct lmh_witness {
    lm_tree* tree = nullptr;
    u32 root[8];
    std::vector<EF> ood_points, ood_answers;
};

namespace {

void add_base(lmh_prover* p, const u32* s, u64 n) {
    p->ch.observe_many(s, n);
    p->transcript.insert(p->transcript.end(), s, s + n);
}
void add_ext(lmh_prover* p, const std::vector<EF>& v) {
    std::vector<u32> f(v.size() * 5);
    for (size_t i = 0; i < v.size(); i++) memcpy(&f[5 * i], v[i].v, 20);
    add_base(p, f.data(), f.size());
}
bool sample_vec(lmh_prover* p, u64 n, std::vector<EF>& out) {  // fiat-shamir/src/utils.rs:43-58
    std::vector<u32> fe;
    if (!p->ch.sample_many((n * 5 + 7) / 8, fe)) return false;
    out.resize(n);
    for (u64 i = 0; i < n; i++) memcpy(out[i].v, &fe[5 * i], 20);
    return true;
}
bool sample_in_range(lmh_prover* p, u32 bits, u64 n, std::vector<u64>& out) {  // challenger.rs:66-75
    std::vector<u32> fe;
    if (!p->ch.sample_many((n + 7) / 8, fe)) return false;
    out.resize(n);
    for (u64 i = 0; i < n; i++) out[i] = (u64)kb::from_monty(fe[i]) & ((1ull << bits) - 1);
    return true;
}
// prover.rs:100-114 + utils.rs:30-41
void add_sumcheck_poly(lmh_prover* p, const std::vector<EF>& coeffs, const EF* eq_alpha) {
    std::vector<u32> bare(coeffs.size() * 5);
    for (size_t i = 0; i < coeffs.size(); i++) memcpy(&bare[5 * i], coeffs[i].v, 20);
    if (!eq_alpha) {
        p->ch.observe_many(bare.data(), bare.size());
    } else {
        EF oma = kb::ef_sub(kb::ef_one(), *eq_alpha);
        EF tam = kb::ef_sub(kb::ef_dbl(*eq_alpha), kb::ef_one());
        size_t d = coeffs.size() - 1;
        std::vector<EF> full;
        full.push_back(kb::ef_mul(oma, coeffs[0]));
        for (size_t k = 1; k <= d; k++) full.push_back(kb::ef_add(kb::ef_mul(oma, coeffs[k]), kb::ef_mul(tam, coeffs[k - 1])));
        full.push_back(kb::ef_mul(tam, coeffs[d]));
        std::vector<u32> f(full.size() * 5);
        for (size_t i = 0; i < full.size(); i++) memcpy(&f[5 * i], full[i].v, 20);
        p->ch.observe_many(f.data(), f.size());
    }
    p->transcript.insert(p->transcript.end(), bare.begin() + 5, bare.end());
}
int pow_grinding(lm_ctx* ctx, lmh_prover* p, u32 bits) {  // prover.rs:120-177
    if (bits == 0) return LM_OK;
    u32 w;
    int rc = lm_pow_grind(ctx, p->ch.state, bits, &w);
    if (rc) return rc;
    p->ch.observe_many(&w, 1);
    if ((kb::from_monty(p->ch.state[8]) & ((1u << bits) - 1)) != 0) return LM_E_INVALID;
    p->transcript.push_back(w);
    return LM_OK;
}

struct Sumcheck {  // SumcheckSingle, open.rs:322-330 — device resident
    lm_ctx* ctx = nullptr;
    const u32* f = nullptr;  // current evals (base at the very beginning, SoA EF afterwards)
    bool f_is_ext = false;
    u32* W = nullptr;        // current weights (SoA EF)
    u32 n_vars = 0;
    EF sum;
    u32 *f_buf[2] = {nullptr, nullptr}, *w_buf[2] = {nullptr, nullptr};
    int f_cur = -1, w_cur = 0;
    ~Sumcheck() {
        for (int i = 0; i < 2; i++) {
            if (f_buf[i]) lm_free(ctx, f_buf[i]);
            if (w_buf[i]) lm_free(ctx, w_buf[i]);
        }
    }
};

// run_product_sumcheck / run_sumcheck_many_rounds (product_computation.rs:37-125, open.rs:384-409).
// Two rounds per pass over the tables (lm_prod_round2 / lm_fold2_round): the device returns round t's (c0, c2) together with
// round t+1's as quadratics in the challenge of round t, which are evaluated here once that challenge is sampled; the two
// challenges are folded in by the next pass.  The weights change between calls (add_new_equality), so the look-ahead never
// crosses a call, and the tables are brought up to date before returning.
int sumcheck_rounds(lm_ctx* ctx, lmh_prover* p, Sumcheck& sc, u32 n_rounds, u32 pow_bits, std::vector<EF>& challenges) {
    std::vector<EF> pending;  // sampled, not yet folded into the tables
    bool la_valid = false;
    EF la[6];
    auto advance = [&](int fn, int wn, u32 vars) {
        sc.f = sc.f_buf[fn];
        sc.f_is_ext = true;
        sc.f_cur = fn;
        sc.W = sc.w_buf[wn];
        sc.w_cur = wn;
        sc.n_vars -= vars;
        pending.clear();
    };
    for (u32 r = 0; r < n_rounds; r++) {
        int rc;
        EF c0, c2;
        const u32 remaining = n_rounds - r;
        const int fn = sc.f_cur < 0 ? 0 : 1 - sc.f_cur, wn = 1 - sc.w_cur;
        if (la_valid) {
            const EF& x = pending.back();
            c0 = kb::ef_add(la[0], kb::ef_mul(x, kb::ef_add(la[1], kb::ef_mul(x, la[2]))));
            c2 = kb::ef_add(la[3], kb::ef_mul(x, kb::ef_add(la[4], kb::ef_mul(x, la[5]))));
            la_valid = false;
        } else {
            u32 s[40];
            bool eight = false;
            if (pending.empty()) {
                eight = remaining >= 2 && sc.n_vars >= 2;
                rc = eight ? lm_prod_round2(ctx, sc.f, sc.f_is_ext, sc.W, sc.n_vars, s) : lm_prod_round(ctx, sc.f, sc.f_is_ext, sc.W, sc.n_vars, s);
                if (rc) return rc;
            } else if (pending.size() == 1) {
                if ((rc = lm_fold_round(ctx, sc.f, sc.f_is_ext, sc.W, sc.n_vars, pending[0].v, sc.f_buf[fn], sc.w_buf[wn], s))) return rc;
                advance(fn, wn, 1);
            } else {
                eight = remaining >= 2 && sc.n_vars >= 4;
                rc = lm_fold2_round(ctx, sc.f, sc.f_is_ext, sc.W, sc.n_vars, pending[0].v, pending[1].v, sc.f_buf[fn], sc.w_buf[wn], eight ? 2 : 1, s);
                if (rc) return rc;
                advance(fn, wn, 2);
            }
            if (eight) {
                const EF P00 = ef_load(s), P01 = ef_load(s + 5), P10 = ef_load(s + 10), Q0 = ef_load(s + 15), Q1 = ef_load(s + 20);
                const EF T0 = ef_load(s + 25), T2 = ef_load(s + 30), T3 = ef_load(s + 35);
                c0 = kb::ef_add(P00, P01);
                c2 = kb::ef_add(Q0, Q1);
                la[0] = P00, la[1] = kb::ef_sub(kb::ef_sub(P10, P00), Q0), la[2] = Q0;
                la[3] = T0, la[4] = kb::ef_sub(kb::ef_sub(T3, T0), T2), la[5] = T2;
                la_valid = true;
            } else {
                c0 = ef_load(s), c2 = ef_load(s + 5);
            }
        }
        const EF c1 = kb::ef_sub(kb::ef_sub(sc.sum, kb::ef_dbl(c0)), c2);
        add_sumcheck_poly(p, {c0, c1, c2}, nullptr);
        rc = pow_grinding(ctx, p, pow_bits);
        if (rc) return rc;
        std::vector<EF> chv;
        if (!sample_vec(p, 1, chv)) return LM_E_INVALID;
        const EF ch = chv[0];
        challenges.push_back(ch);
        sc.sum = kb::ef_add(c0, kb::ef_mul(ch, kb::ef_add(c1, kb::ef_mul(ch, c2))));
        pending.push_back(ch);
    }
    // bring the tables up to date (the caller commits to / reads the folded polynomial next)
    if (!pending.empty()) {
        int rc;
        const int fn = sc.f_cur < 0 ? 0 : 1 - sc.f_cur, wn = 1 - sc.w_cur;
        if (pending.size() == 1) {
            if ((rc = lm_fold(ctx, sc.f, sc.f_is_ext, sc.n_vars, pending[0].v, sc.f_buf[fn]))) return rc;
            if ((rc = lm_fold(ctx, sc.W, 1, sc.n_vars, pending[0].v, sc.w_buf[wn]))) return rc;
            advance(fn, wn, 1);
        } else {
            if ((rc = lm_fold2_round(ctx, sc.f, sc.f_is_ext, sc.W, sc.n_vars, pending[0].v, pending[1].v, sc.f_buf[fn], sc.w_buf[wn], 0, nullptr))) return rc;
            advance(fn, wn, 2);
        }
    }
    return LM_OK;
}

int open_and_hint(lm_ctx* ctx, lmh_prover* p, const lm_tree* tree, const std::vector<u64>& idx, std::vector<u32>& leaves,
                  u32& leaf_words) {
    leaf_words = lm_tree_leaf_words(tree);
    u32 log_h = lm_tree_log_height(tree);
    leaves.resize((u64)idx.size() * leaf_words);
    std::vector<u32> sib((u64)idx.size() * log_h * 8 + 1);
    int rc = lm_tree_open(ctx, tree, idx.data(), (u32)idx.size(), leaves.data(), sib.data());
    if (rc) return rc;
    p->batch_sizes.push_back((u32)idx.size());
    for (size_t q = 0; q < idx.size(); q++) {
        Opening o;
        o.index = idx[q];
        o.leaf.assign(leaves.begin() + q * leaf_words, leaves.begin() + (q + 1) * leaf_words);
        o.path.assign(sib.begin() + q * log_h * 8, sib.begin() + (q + 1) * log_h * 8);
        p->openings.push_back(std::move(o));
    }
    return LM_OK;
}

// the second half of open_and_hint for an opening begun with lm_tree_open_begin (consumes the handle)
int collect_opening(lm_ctx* ctx, lmh_prover* p, const lm_tree* tree, lm_tree_opening* opening, const std::vector<u64>& idx,
                    std::vector<u32>& leaves, u32& leaf_words) {
    leaf_words = lm_tree_leaf_words(tree);
    const u32 log_h = lm_tree_log_height(tree);
    leaves.resize((u64)idx.size() * leaf_words);
    std::vector<u32> sib((u64)idx.size() * log_h * 8 + 1);
    int rc = lm_tree_open_end(ctx, opening, leaves.data(), sib.data());
    if (rc) return rc;
    p->batch_sizes.push_back((u32)idx.size());
    for (size_t q = 0; q < idx.size(); q++) {
        Opening o;
        o.index = idx[q];
        o.leaf.assign(leaves.begin() + q * leaf_words, leaves.begin() + (q + 1) * leaf_words);
        o.path.assign(sib.begin() + q * log_h * 8, sib.begin() + (q + 1) * log_h * 8);
        p->openings.push_back(std::move(o));
    }
    return LM_OK;
}

u32 fold_at(const lm_whir_config* c, u32 round) { return round == 0 ? c->folding_factor_first : c->folding_factor_subsequent; }
u32 total_fold(const lm_whir_config* c, u32 n_rounds) { return c->folding_factor_first + c->folding_factor_subsequent * n_rounds; }

// sample_ood_points (whir/src/utils.rs:30-57) on a device polynomial
int sample_ood(lm_ctx* ctx, lmh_prover* p, u32 n_samples, u32 num_variables, const u32* d_poly, bool is_ext,
               std::vector<EF>& pts, std::vector<EF>& ans) {
    pts.clear();
    ans.clear();
    if (!n_samples) return LM_OK;
    if (!sample_vec(p, n_samples, pts)) return LM_E_INVALID;
    // all samples in one call: the polynomial is read once per PAIR of points (lm_mle_eval_points)
    std::vector<u32> coords;
    coords.reserve((size_t)n_samples * num_variables * 5);
    for (EF z : pts)
        for (const EF& e : expand_from_univariate(z, num_variables)) coords.insert(coords.end(), e.v, e.v + 5);
    ans.resize(n_samples);
    int rc = lm_mle_eval_points(ctx, d_poly, is_ext, num_variables, n_samples, num_variables ? coords.data() : nullptr, ans[0].v);
    if (rc) return rc;
    add_ext(p, ans);
    return LM_OK;
}

}  // namespace

extern "C" {

lmh_prover* lmh_prover_new(void) { return new lmh_prover(); }
void lmh_prover_free(lmh_prover* p) { delete p; }
void lmh_add_base_scalars(lmh_prover* p, const uint32_t* s, uint64_t n) { add_base(p, s, n); }
void lmh_observe_scalars(lmh_prover* p, const uint32_t* s, uint64_t n) { p->ch.observe_many(s, n); }
void lmh_add_extension_scalars(lmh_prover* p, const uint32_t* ef, uint64_t n) { add_base(p, ef, n * 5); }
void lmh_duplex(lmh_prover* p) { p->ch.duplex(); }
int lmh_sample_vec(lmh_prover* p, uint64_t n, uint32_t* out) {
    std::vector<EF> v;
    if (!sample_vec(p, n, v)) return LM_E_INVALID;
    for (u64 i = 0; i < n; i++) memcpy(out + 5 * i, v[i].v, 20);
    return LM_OK;
}
int lmh_sample_in_range(lmh_prover* p, uint32_t bits, uint64_t n, uint64_t* out) {
    std::vector<u64> v;
    if (bits >= 31 || !sample_in_range(p, bits, n, v)) return LM_E_INVALID;
    memcpy(out, v.data(), n * 8);
    return LM_OK;
}
void lmh_add_sumcheck_polynomial(lmh_prover* p, const uint32_t* coeffs, uint32_t n, const uint32_t* eq_alpha) {
    std::vector<EF> c(n);
    for (u32 i = 0; i < n; i++) c[i] = ef_load(coeffs + 5 * i);
    if (eq_alpha) {
        EF a = ef_load(eq_alpha);
        add_sumcheck_poly(p, c, &a);
    } else {
        add_sumcheck_poly(p, c, nullptr);
    }
}
int lmh_pow_grinding(lm_ctx* ctx, lmh_prover* p, uint32_t bits) { return pow_grinding(ctx, p, bits); }
void lmh_challenger_state(const lmh_prover* p, uint32_t out16[16]) { memcpy(out16, p->ch.state, 64); }

// ---- Merkle-path pruning (crates/backend/fiat-shamir/src/merkle_pruning.rs:18-86) ---------------------------------------
// Per batch: sort by leaf index, drop repeated leaves, cut the all-zero tail shared by every leaf, and keep for path i only
// the siblings below its divergence from path i-1, except the one level where path i+1 joins it (that hash is recomputed
// by the verifier from path i+1).  Blob layout: include/leanmultisig_host.h.
}  // extern "C"
namespace lmh {
static unsigned diverge_level(u64 a, u64 b) {  // lca_level: number of low bits to drop until a == b
    unsigned l = 0;
    for (u64 x = a ^ b; x; x >>= 1) l++;
    return l;
}
std::vector<PrunedBatch> prune(const lmh_prover* p) {
    std::vector<PrunedBatch> out;
    size_t first = 0;
    for (u32 bs : p->batch_sizes) {
        const Opening* ops = p->openings.data() + first;
        first += bs;
        std::vector<u32> by_index(bs);
        for (u32 i = 0; i < bs; i++) by_index[i] = i;
        std::stable_sort(by_index.begin(), by_index.end(), [&](u32 a, u32 b) { return ops[a].index < ops[b].index; });
        std::vector<u32> kept;  // distinct leaves in index order
        PrunedBatch pb;
        pb.original_order.resize(bs);
        for (u32 i : by_index) {
            if (kept.empty() || ops[kept.back()].index != ops[i].index) kept.push_back(i);
            pb.original_order[i] = (u32)kept.size() - 1;
        }
        const size_t leaf_len = ops[kept[0]].leaf.size();
        size_t live = leaf_len;  // length after cutting the common zero tail
        while (live > 0) {
            bool zero = true;
            for (u32 i : kept) zero = zero && ops[i].leaf[live - 1] == 0;
            if (!zero) break;
            live--;
        }
        pb.merkle_height = (u32)(ops[kept[0]].path.size() / 8);
        pb.n_trailing_zeros = (u32)(leaf_len - live);
        for (size_t k = 0; k < kept.size(); k++) {
            const Opening& me = ops[kept[k]];
            const unsigned top = k == 0 ? pb.merkle_height : diverge_level(ops[kept[k - 1]].index, me.index);
            const int hole = k + 1 < kept.size() ? (int)diverge_level(me.index, ops[kept[k + 1]].index) - 1 : -1;
            PrunedPath pp;
            pp.leaf_index = me.index;
            pp.leaf.assign(me.leaf.begin(), me.leaf.begin() + live);
            for (unsigned lvl = 0; lvl < top; lvl++) {
                if ((int)lvl == hole) continue;
                pp.siblings.insert(pp.siblings.end(), me.path.begin() + 8 * lvl, me.path.begin() + 8 * lvl + 8);
            }
            pb.paths.push_back(std::move(pp));
        }
        out.push_back(std::move(pb));
    }
    return out;
}
}  // namespace lmh
namespace {
const std::vector<u32>& pruned_blob(const lmh_prover* p) {
    const size_t key[3] = {p->transcript.size(), p->openings.size(), p->batch_sizes.size()};
    if (memcmp(key, p->pruned_key, sizeof key) == 0) return p->pruned_cache;
    std::vector<u32>& o = p->pruned_cache;
    o.clear();
    {
        size_t words = p->transcript.size() + 64;
        for (const lmh::Opening& op : p->openings) words += op.leaf.size() + op.path.size() + 8;
        o.reserve(words);  // upper bound: nothing pruned
    }
    o.push_back((u32)p->transcript.size());
    o.insert(o.end(), p->transcript.begin(), p->transcript.end());
    const std::vector<lmh::PrunedBatch> batches = lmh::prune(p);
    o.push_back((u32)batches.size());
    for (const lmh::PrunedBatch& pb : batches) {
        o.push_back(pb.merkle_height);
        o.push_back(pb.n_trailing_zeros);
        o.push_back((u32)pb.original_order.size());
        o.insert(o.end(), pb.original_order.begin(), pb.original_order.end());
        o.push_back((u32)pb.paths.size());
        for (const lmh::PrunedPath& pp : pb.paths) {
            o.push_back((u32)pp.leaf_index);
            o.push_back((u32)(pp.leaf_index >> 32));
            o.push_back((u32)pp.leaf.size());
            o.insert(o.end(), pp.leaf.begin(), pp.leaf.end());
            o.push_back((u32)(pp.siblings.size() / 8));
            o.insert(o.end(), pp.siblings.begin(), pp.siblings.end());
        }
    }
    memcpy(p->pruned_key, key, sizeof key);
    return o;
}
}  // namespace
extern "C" {

uint64_t lmh_proof_pruned_words(const lmh_prover* p) { return p ? pruned_blob(p).size() : 0; }
void lmh_proof_pruned_copy(const lmh_prover* p, uint32_t* out) {
    const std::vector<u32>& b = pruned_blob(p);
    memcpy(out, b.data(), b.size() * 4);
    // the cache only bridges the size query and this copy (its key is the three sizes, which every mutation of a prover changes:
    // transcript and openings are append-only; lmh_prover_load_raw resets it): released here, rebuilt by the next query.
    // A prover object is not thread-safe, const calls included.
    p->pruned_key[0] = ~(size_t)0;
    std::vector<u32>().swap(p->pruned_cache);
}
// Proof::proof_size_fe (fiat-shamir/src/transcript.rs:39-53): transcript + pruned leaf data + 8 words per kept sibling
uint64_t lmh_proof_size_fe(const lmh_prover* p) {
    if (!p) return 0;
    u64 fe = p->transcript.size();
    for (const lmh::PrunedBatch& pb : lmh::prune(p))
        for (const lmh::PrunedPath& pp : pb.paths) fe += pp.leaf.size() + pp.siblings.size();
    return fe;
}
uint32_t lmh_proof_n_batches(const lmh_prover* p) { return p ? (uint32_t)p->batch_sizes.size() : 0; }
void lmh_proof_batch_sizes(const lmh_prover* p, uint32_t* out) {
    if (p && !p->batch_sizes.empty()) memcpy(out, p->batch_sizes.data(), p->batch_sizes.size() * 4);
}

int lmh_prover_load_raw(lmh_prover* p, const uint32_t* blob, uint64_t n_words, const uint32_t* batch_sizes, uint32_t n_batches) {
    if (!p || !blob || n_words < 2 || (n_batches && !batch_sizes)) return LM_E_INVALID;
    p->pruned_key[0] = ~(size_t)0;  // (the cached pruned blob describes the previous contents)
    u64 k = 0;
    const u64 T = blob[k++];
    if (T + 2 > n_words) return LM_E_INVALID;
    p->transcript.assign(blob + k, blob + k + T);
    k += T;
    const u64 M = blob[k++];
    p->openings.clear();
    for (u64 i = 0; i < M; i++) {
        if (k + 4 > n_words) return LM_E_INVALID;
        Opening o;
        o.index = blob[k] | (u64)blob[k + 1] << 32;
        const u64 ll = blob[k + 2], pl = blob[k + 3];
        k += 4;
        if (k + ll + pl > n_words || pl % 8) return LM_E_INVALID;
        o.leaf.assign(blob + k, blob + k + ll);
        o.path.assign(blob + k + ll, blob + k + ll + pl);
        k += ll + pl;
        p->openings.push_back(std::move(o));
    }
    u64 total = 0;
    for (u32 b = 0; b < n_batches; b++) total += batch_sizes[b];
    if (total != M || k != n_words) return LM_E_INVALID;
    for (u32 b = 0; b < n_batches; b++)
        if (batch_sizes[b] == 0) return LM_E_INVALID;
    p->batch_sizes.assign(batch_sizes, batch_sizes + n_batches);
    return LM_OK;
}

uint64_t lmh_proof_words(const lmh_prover* p) {
    u64 n = 2 + p->transcript.size();
    for (const Opening& o : p->openings) n += 4 + o.leaf.size() + o.path.size();
    return n;
}
void lmh_proof_copy(const lmh_prover* p, uint32_t* out) {
    u64 k = 0;
    out[k++] = (u32)p->transcript.size();
    memcpy(out + k, p->transcript.data(), p->transcript.size() * 4);
    k += p->transcript.size();
    out[k++] = (u32)p->openings.size();
    for (const Opening& o : p->openings) {
        out[k++] = (u32)o.index;
        out[k++] = (u32)(o.index >> 32);
        out[k++] = (u32)o.leaf.size();
        out[k++] = (u32)o.path.size();
        memcpy(out + k, o.leaf.data(), o.leaf.size() * 4);
        k += o.leaf.size();
        memcpy(out + k, o.path.data(), o.path.size() * 4);
        k += o.path.size();
    }
}

int lmh_whir_commit(lm_ctx* ctx, lmh_prover* p, const lm_whir_config* c, const uint32_t* d_poly, uint64_t actual_len,
                    lmh_witness** out) {
    if (!ctx || !p || !c || !d_poly || !out) return LM_E_INVALID;
    lmh_witness* w = new lmh_witness();
    int rc = lm_commit(ctx, d_poly, 0, c->num_variables, c->folding_factor_first, c->starting_log_inv_rate, actual_len,
                       &w->tree, w->root);
    if (rc) {
        delete w;
        return rc;
    }
    add_base(p, w->root, 8);
    rc = sample_ood(ctx, p, c->commitment_ood_samples, c->num_variables, d_poly, false, w->ood_points, w->ood_answers);
    if (rc) {
        lmh_witness_free(ctx, w);
        return rc;
    }
    *out = w;
    return LM_OK;
}
void lmh_witness_free(lm_ctx* ctx, lmh_witness* w) {
    if (!w) return;
    if (w->tree) lm_tree_free(ctx, w->tree);
    delete w;
}
void lmh_witness_root(const lmh_witness* w, uint32_t root[8]) { memcpy(root, w->root, 32); }

int lmh_whir_prove(lm_ctx* ctx, lmh_prover* p, const lm_whir_config* c, const lm_sparse_statement* statements,
                   uint32_t n_statements, const uint32_t* points, uint64_t n_point_coords, const uint64_t* selectors,
                   const uint32_t* values, uint64_t n_values, lmh_witness* witness, const uint32_t* d_poly,
                   uint32_t* out_point) {
    // "Consumes the witness": on EVERY path out of this function the witness and the tree it owns are released
    struct WitnessGuard {
        lm_ctx* ctx;
        lmh_witness* w;
        ~WitnessGuard() {
            if (w) lmh_witness_free(ctx, w);
        }
    } witness_guard{ctx, witness};
    if (!ctx || !p || !c || !witness || !d_poly || !out_point) {
        witness_guard.w = ctx ? witness : nullptr;
        lm_set_error("lmh_whir_prove: null argument");
        return LM_E_INVALID;
    }
    const u32 n = c->num_variables;
    if (c->n_rounds > LM_MAX_WHIR_ROUNDS) {
        lm_set_error("lmh_whir_prove: more than LM_MAX_WHIR_ROUNDS rounds");
        return LM_E_INVALID;
    }
    // validate_parameters, open.rs:18-20
    if (n != total_fold(c, c->n_rounds) + c->final_sumcheck_rounds) {
        lm_set_error("lmh_whir_prove: folding factors and final sumcheck rounds do not add up to num_variables (validate_parameters)");
        return LM_E_INVALID;
    }
    int rc;

    // ---- initialize_first_round_state (open.rs:467-510): OOD statements first, then the caller's --------------
    std::vector<lm_weight_item> items;
    std::vector<u32> pts;      // EF coordinates
    std::vector<u32> scalars;  // EF per item
    p->ch.duplex();
    std::vector<EF> gv;
    if (!sample_vec(p, 1, gv)) return LM_E_INVALID;
    const EF gamma = gv[0];
    EF gp = kb::ef_one(), sum = kb::ef_zero();
    auto push_item = [&](u64 offset, u32 inner, u32 is_next, u64 point_off, const EF& scalar) {
        lm_weight_item it;
        it.offset = offset;
        it.inner_n = inner;
        it.is_next = is_next;
        it.point_offset = point_off;
        items.push_back(it);
        scalars.insert(scalars.end(), scalar.v, scalar.v + 5);
    };
    for (size_t i = 0; i < witness->ood_points.size(); i++) {
        std::vector<EF> pt = expand_from_univariate(witness->ood_points[i], n);
        u64 off = pts.size() / 5;
        for (const EF& e : pt) pts.insert(pts.end(), e.v, e.v + 5);
        push_item(0, n, 0, off, gp);
        sum = kb::ef_add(sum, kb::ef_mul(witness->ood_answers[i], gp));
        gp = kb::ef_mul(gp, gamma);
    }
    const u64 user_pt_base = pts.size() / 5;
    if (n_point_coords) pts.insert(pts.end(), points, points + n_point_coords * 5);
    for (u32 s = 0; s < n_statements; s++) {
        const lm_sparse_statement& st = statements[s];
        if (st.point_len > n || st.n_values == 0 || st.point_offset + st.point_len > n_point_coords ||
            st.values_offset + st.n_values > n_values)
        {
            lm_set_error("lmh_whir_prove: statement %u is malformed (validate_statement)", s);
            return LM_E_INVALID;  // open.rs:22-28
        }
        for (u32 v = 0; v < st.n_values; v++) {
            u64 sel = selectors[st.values_offset + v];
            if (sel >= (1ull << (n - st.point_len))) {
                lm_set_error("lmh_whir_prove: selector of statement %u does not fit", s);
                return LM_E_INVALID;
            }
            push_item(sel << st.point_len, st.point_len, st.is_next, user_pt_base + st.point_offset, gp);
            sum = kb::ef_add(sum, kb::ef_mul(ef_load(values + (st.values_offset + v) * 5), gp));
            gp = kb::ef_mul(gp, gamma);
        }
    }

    StageClock wclk(ctx);
    wclk.prefix = "  whir.";
    Sumcheck sc;
    sc.ctx = ctx;
    const u64 len = 1ull << n;
    if ((rc = lm_malloc(ctx, 5 * len, &sc.w_buf[0]))) return rc;
    if ((rc = lm_malloc(ctx, 5 * (len / 2) + 8, &sc.w_buf[1]))) return rc;
    if ((rc = lm_malloc(ctx, 5 * (len / 2) + 8, &sc.f_buf[0]))) return rc;
    if ((rc = lm_malloc(ctx, 5 * (len / 4) + 8, &sc.f_buf[1]))) return rc;
    // combine_statement, open.rs:518-584
    rc = lm_weights_init(ctx, sc.w_buf[0], n, items.data(), (u32)items.size(), pts.data(), pts.size() / 5, scalars.data());
    if (rc) return rc;
    sc.f = d_poly;
    sc.f_is_ext = false;
    sc.W = sc.w_buf[0];
    sc.w_cur = 0;
    sc.n_vars = n;
    sc.sum = sum;
    // f_buf[1] must hold the second fold (len/4) and later ones; f_buf[0] the first (len/2): ping-pong sizes shrink.

    wclk.mark("weights_init");
    std::vector<EF> randomness;
    if ((rc = sumcheck_rounds(ctx, p, sc, fold_at(c, 0), c->starting_folding_pow_bits, randomness))) return rc;
    wclk.mark("sumcheck_0");

    u64 domain_size = 1ull << (n + c->starting_log_inv_rate);
    u32 next_domain_gen_log = ilog2(domain_size) - fold_at(c, 0);  // two_adic_generator(bits)
    lm_tree* tree = witness->tree;
    witness->tree = nullptr;
    bool tree_is_ext = false;
    const u32 g24 = kb::to_monty(0x6ac49f88u);  // generator of the 2^24-th roots (koala_bear.rs:50-54)
    auto two_adic_generator = [&](u32 bits) {
        u32 g = g24;
        for (u32 i = bits; i < 24; i++) g = kb::sqr(g);
        return g;
    };
    auto fail = [&](int code) {
        if (tree) lm_tree_free(ctx, tree);
        return code;  // (the witness itself goes with witness_guard)
    };

    for (u32 round = 0; round <= c->n_rounds; round++) {
        const u32 num_variables = n - total_fold(c, round);
        if (round == c->n_rounds) {
            // ---- final_round (open.rs:182-248) ----
            const u64 m = 1ull << num_variables;
            std::vector<u32> soa(5 * m);
            if ((rc = lm_download(ctx, soa.data(), sc.f, 5 * m))) return fail(rc);
            std::vector<EF> coeffs(m);
            for (u64 i = 0; i < m; i++)
                for (int k = 0; k < 5; k++) coeffs[i].v[k] = soa[(u64)k * m + i];
            // evals_to_coeffs, poly/src/evals.rs:44-56
            for (u64 half = 1; half < m; half <<= 1)
                for (u64 i = 0; i < m; i += 2 * half)
                    for (u64 j = 0; j < half; j++) coeffs[i + j + half] = kb::ef_sub(coeffs[i + j + half], coeffs[i + j]);
            for (u64 i = 0; i < m; i++) {
                u64 j = 0;
                for (u32 b = 0; b < num_variables; b++)
                    if ((i >> b) & 1) j |= 1ull << (num_variables - 1 - b);
                if (i < j) std::swap(coeffs[i], coeffs[j]);
            }
            add_ext(p, coeffs);
            wclk.mark("final.coeffs");
            if ((rc = pow_grinding(ctx, p, c->final_query_pow_bits))) return fail(rc);
            std::vector<u64> idx;
            if (!sample_in_range(p, ilog2(domain_size >> fold_at(c, round)), c->final_queries, idx)) return fail(LM_E_INVALID);
            std::vector<u32> leaves;
            u32 lw;
            if ((rc = open_and_hint(ctx, p, tree, idx, leaves, lw))) return fail(rc);
            wclk.mark("final.pow+queries");
            if (c->final_sumcheck_rounds > 0)
                if ((rc = sumcheck_rounds(ctx, p, sc, c->final_sumcheck_rounds, 0, randomness))) return fail(rc);
            wclk.mark("final.sumcheck");
            break;
        }
        const u32 fnext = fold_at(c, round + 1);
        const u32 rs_red = round == 0 ? c->rs_domain_initial_reduction_factor : 1;
        const u64 new_domain_size = domain_size >> rs_red;
        const u64 inv_rate = new_domain_size >> num_variables;
        // reorder_and_dft + MerkleData::build on the folded polynomial (open.rs:75-90)
        lm_tree* new_tree = nullptr;
        u32 root[8];
        rc = lm_commit(ctx, sc.f, 1, num_variables, fnext, ilog2(inv_rate), 1ull << num_variables, &new_tree, root);
        if (rc) return fail(rc);
        wclk.mark("round.commit");
        add_base(p, root, 8);
        std::vector<EF> ood_points, ood_answers;
        rc = sample_ood(ctx, p, c->rounds[round].ood_samples, num_variables, sc.f, true, ood_points, ood_answers);
        wclk.mark("round.ood");
        if (!rc) rc = pow_grinding(ctx, p, c->rounds[round].query_pow_bits);
        wclk.mark("round.pow");
        std::vector<u64> idx;
        if (!rc && !sample_in_range(p, ilog2(domain_size >> fold_at(c, round)), c->rounds[round].num_queries, idx)) rc = LM_E_INVALID;
        // The openings are hints: they do not enter the challenger, so nothing up to the next sumcheck depends on the opened LEAVES
        // except the claimed sum.  The opening kernel is enqueued, the combination randomness is sampled and the weight kernels —
        // which depend on the query INDICES only — are enqueued behind it; leaves and paths are collected, and the STIR evaluations
        // computed, while those run (open.rs:92-181 in transcript order: hint_merkle_paths, then sample gamma).
        lm_tree_opening* opening = nullptr;
        if (!rc && !idx.empty()) rc = lm_tree_open_begin(ctx, tree, idx.data(), (u32)idx.size(), &opening);
        if (rc) {
            lm_tree_free(ctx, new_tree);
            return fail(rc);
        }
        auto drop = [&](int code) {
            (void)lm_tree_open_end(ctx, opening, nullptr, nullptr);
            lm_tree_free(ctx, new_tree);
            return fail(code);
        };
        wclk.mark("round.open_queries(launch)");
        p->ch.duplex();
        std::vector<EF> g1;
        if (!sample_vec(p, 1, g1)) return drop(LM_E_INVALID);
        const EF g = g1[0];
        // add_new_equality + add_new_base_equality (open.rs:337-382)
        items.clear();
        pts.clear();
        scalars.clear();
        EF gpw = kb::ef_one();
        for (size_t i = 0; i < ood_points.size(); i++) {
            std::vector<EF> pt = expand_from_univariate(ood_points[i], num_variables);
            u64 off = pts.size() / 5;
            for (const EF& e : pt) pts.insert(pts.end(), e.v, e.v + 5);
            push_item(0, num_variables, 0, off, gpw);
            sc.sum = kb::ef_add(sc.sum, kb::ef_mul(gpw, ood_answers[i]));
            gpw = kb::ef_mul(gpw, g);
        }
        const u32 dom_gen = two_adic_generator(next_domain_gen_log);
        std::vector<EF> stir_scalar(idx.size());
        {
            // the query points z, z^2, z^4, .. are base-field: squarings in the base field, embedded (expand_from_univariate on
            // ef_from_base(z) computes the same words with 19 extension-field squarings per query)
            const size_t at = pts.size();
            pts.resize(at + idx.size() * (size_t)num_variables * 5, 0u);
            u32* w = pts.data() + at;
            for (size_t q = 0; q < idx.size(); q++) {
                u32 z = kb::pow(dom_gen, idx[q]);
                const u64 off = (at + q * (size_t)num_variables * 5) / 5;
                for (u32 j = 0; j < num_variables; j++, w += 5) {
                    w[0] = z;
                    z = kb::sqr(z);
                }
                push_item(0, num_variables, 0, off, gpw);
                stir_scalar[q] = gpw;
                gpw = kb::ef_mul(gpw, g);
            }
        }
        wclk.mark("round.items(host)");
        rc = lm_weights_accumulate(ctx, sc.W, num_variables, items.data(), (u32)items.size(), pts.data(), pts.size() / 5,
                                   scalars.data());
        if (rc) return drop(rc);
        wclk.mark("round.weights(launch)");
        std::vector<u32> leaves;
        u32 lw = 0;
        if ((rc = collect_opening(ctx, p, tree, opening, idx, leaves, lw))) {
            lm_tree_free(ctx, new_tree);
            return fail(rc);
        }
        wclk.mark("round.open_queries(collect)");
        const u32 ff = fold_at(c, round);
        const EF* folding_randomness = randomness.data() + (randomness.size() - ff);
        // leaf value at the folding randomness = <leaf, eq(randomness, .)>: the eq table is shared by all queries (a base
        // leaf then costs 5 multiplications per word instead of a chain of EF folds; same field element either way)
        {
            const u64 m = 1ull << ff;
            std::vector<EF> eq(m);
            eq[0] = kb::ef_one();
            for (u32 j = 0; j < ff; j++) {  // after step j: eq over the first j+1 coordinates, coordinate 0 <-> MSB
                const u64 cur = 1ull << j;
                for (u64 i = cur; i-- > 0;) {
                    const EF hi = kb::ef_mul(eq[i], folding_randomness[j]);
                    eq[2 * i + 1] = hi;
                    eq[2 * i] = kb::ef_sub(eq[i], hi);
                }
            }
            for (size_t q = 0; q < idx.size(); q++) {
                const u32* leaf = &leaves[q * lw];
                EF acc = kb::ef_zero();
                if (tree_is_ext)
                    for (u64 i = 0; i < m; i++) acc = kb::ef_add(acc, kb::ef_mul(eq[i], ef_load(leaf + 5 * i)));
                else {  // five dot products with delayed reduction (kb::dot_n's schedule: 4 products, then 3 per fold)
                    u64 a64[5] = {0, 0, 0, 0, 0};
                    int room = 4;
                    for (u64 i = 0; i < m; i++) {
                        if (room == 0) {
                            for (int k = 0; k < 5; k++) a64[k] = kb::fold32(a64[k]);
                            room = 3;
                        }
                        const u64 x = leaf[i];
                        for (int k = 0; k < 5; k++) a64[k] += x * eq[i].v[k];
                        room--;
                    }
                    for (int k = 0; k < 5; k++) acc.v[k] = kb::reduce(kb::fold32(a64[k]));
                }
                sc.sum = kb::ef_add(sc.sum, kb::ef_mul(stir_scalar[q], acc));
            }
        }
        wclk.mark("round.stir_evals(host)");
        if (!rc) rc = sumcheck_rounds(ctx, p, sc, fnext, c->rounds[round].folding_pow_bits, randomness);
        wclk.mark("round.sumcheck");
        if (rc) {
            lm_tree_free(ctx, new_tree);
            return fail(rc);
        }
        domain_size = new_domain_size;
        next_domain_gen_log = ilog2(new_domain_size) - fnext;
        lm_tree_free(ctx, tree);
        tree = new_tree;
        tree_is_ext = true;
    }
    if (tree) lm_tree_free(ctx, tree);
    if (randomness.size() != n) return LM_E_INVALID;
    for (u32 i = 0; i < n; i++) memcpy(out_point + 5 * i, randomness[i].v, 20);
    return LM_OK;
}


// prove_gkr_quotient (quotient_gkr/mod.rs:31-141) — transcript order of SURVEY.md App. B step 4.
int lmh_prove_gkr_quotient(lm_ctx* ctx, lmh_prover* p, const uint32_t* d_nums, const uint32_t* d_dens, uint32_t n_vars,
                           uint32_t out_quotient[5], uint32_t* out_point, uint32_t out_claims[10]) {
    return lmh_prove_gkr_quotient_active(ctx, p, d_nums, d_dens, n_vars, n_vars <= 30 ? 1ull << n_vars : 0, out_quotient, out_point, out_claims);
}
int lmh_prove_gkr_quotient_active(lm_ctx* ctx, lmh_prover* p, const uint32_t* d_nums, const uint32_t* d_dens, uint32_t n_vars,
                                  uint64_t active_len, uint32_t out_quotient[5], uint32_t* out_point, uint32_t out_claims[10]) {
    if (!ctx || !p || !d_nums || !d_dens || !out_quotient || !out_point || !out_claims) return LM_E_INVALID;
    lm_gkr* g = nullptr;
    int rc = lm_gkr_build_active(ctx, d_nums, d_dens, n_vars, active_len, &g);
    if (rc) return rc;
    auto fail = [&](int code) {
        lm_gkr_free(ctx, g);
        return code;
    };
    // LM_STAGE_TIMES: where the wall clock of the layer loop goes (device round trips vs host transcript work)
    const bool clk_on = getenv("LM_STAGE_TIMES") != nullptr;
    double t_round = 0, t_host = 0, t_begin = 0, t_end = 0;
    u32 n_rounds_total = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    u32 tn[160], td[160];
    if ((rc = lm_gkr_top(ctx, g, tn, td))) return fail(rc);
    add_base(p, tn, 160);
    add_base(p, td, 160);
    std::vector<EF> top_n(32), top_d(32);
    EF quotient = kb::ef_zero();
    for (int i = 0; i < 32; i++) {
        top_n[i] = ef_load(tn + 5 * i);
        top_d[i] = ef_load(td + 5 * i);
        quotient = kb::ef_add(quotient, kb::ef_mul(top_n[i], kb::ef_inv(top_d[i])));  // compute_quotient, mod.rs:143-145
    }
    std::vector<EF> point;
    if (!sample_vec(p, 5, point)) return fail(LM_E_INVALID);
    EF claim_num = eval_leaf(tn, true, 5, point.data());
    EF claim_den = eval_leaf(td, true, 5, point.data());
    for (u32 K = 5; K < n_vars; K++) {
        // prove_gkr_layer
        p->ch.duplex();
        std::vector<EF> av;
        if (!sample_vec(p, 1, av)) return fail(LM_E_INVALID);
        const EF alpha = av[0];
        EF sum = kb::ef_add(claim_num, kb::ef_mul(alpha, claim_den));
        EF mmf = kb::ef_one();
        auto tb0 = now();
        if ((rc = lm_gkr_layer_begin(ctx, g, K, point[0].v, alpha.v))) return fail(rc);
        if (clk_on) t_begin += ms(tb0, now());
        std::vector<EF> q;
        EF r_prev;
        // 1 / point[j] for every round of the layer with ONE inversion (the per-round inverse sat between two device exchanges)
        std::vector<EF> inv_pt(K);
        {
            std::vector<EF> pre(K + 1);
            pre[0] = kb::ef_one();
            for (u32 j = 0; j < K; j++) pre[j + 1] = kb::ef_mul(pre[j], point[j]);
            bool zero = true;
            for (int k = 0; k < 5; k++) zero = zero && pre[K].v[k] == 0;
            if (zero) {
                for (u32 j = 0; j < K; j++) inv_pt[j] = kb::ef_inv(point[j]);
            } else {
                EF run = kb::ef_inv(pre[K]);
                for (u32 j = K; j-- > 0;) {
                    inv_pt[j] = kb::ef_mul(run, pre[j]);
                    run = kb::ef_mul(run, point[j]);
                }
            }
        }
        for (u32 t = 0; t < K; t++) {
            u32 c[10];
            auto tr0 = now();
            if ((rc = lm_gkr_round(ctx, g, t == 0 ? nullptr : r_prev.v, c))) return fail(rc);
            auto tr1 = now();
            const EF eq_alpha = point[K - 1 - t];
            // build_bare_from_coeffs, sumcheck_utils.rs:491-503
            const EF c0 = kb::ef_mul(ef_load(c), mmf), c2 = kb::ef_mul(ef_load(c + 5), mmf);
            const EF h1 = kb::ef_mul(kb::ef_sub(sum, kb::ef_mul(kb::ef_sub(kb::ef_one(), eq_alpha), c0)), inv_pt[K - 1 - t]);
            const EF c1 = kb::ef_sub(kb::ef_sub(h1, c0), c2);
            add_sumcheck_poly(p, {c0, c1, c2}, &eq_alpha);
            std::vector<EF> rv;
            if (!sample_vec(p, 1, rv)) return fail(LM_E_INVALID);
            const EF r = rv[0];
            const EF eq_eval = kb::ef_add(kb::ef_mul(kb::ef_sub(kb::ef_one(), eq_alpha), kb::ef_sub(kb::ef_one(), r)),
                                          kb::ef_mul(eq_alpha, r));
            const EF bare_r = kb::ef_add(c0, kb::ef_mul(r, kb::ef_add(c1, kb::ef_mul(r, c2))));
            sum = kb::ef_mul(eq_eval, bare_r);
            mmf = kb::ef_mul(mmf, eq_eval);
            q.push_back(r);
            r_prev = r;
            if (clk_on) {
                t_round += ms(tr0, tr1);
                t_host += ms(tr1, now());
                n_rounds_total++;
            }
        }
        u32 ie[20];
        auto te0 = now();
        if ((rc = lm_gkr_layer_end(ctx, g, r_prev.v, ie))) return fail(rc);
        if (clk_on) t_end += ms(te0, now());
        add_base(p, ie, 20);
        std::vector<EF> bv;
        if (!sample_vec(p, 1, bv)) return fail(LM_E_INVALID);
        const EF beta = bv[0], omb = kb::ef_sub(kb::ef_one(), beta);
        claim_num = kb::ef_add(kb::ef_mul(omb, ef_load(ie)), kb::ef_mul(beta, ef_load(ie + 5)));
        claim_den = kb::ef_add(kb::ef_mul(omb, ef_load(ie + 10)), kb::ef_mul(beta, ef_load(ie + 15)));
        std::vector<EF> np(q.rbegin(), q.rend());
        np.push_back(beta);
        point = np;
    }
    if (clk_on)
        fprintf(stderr, "#   gkr: %u rounds: device round trips %.3f ms, host transcript %.3f ms, layer_begin %.3f ms, layer_end %.3f ms\n",
                n_rounds_total, t_round, t_host, t_begin, t_end);
    lm_gkr_free(ctx, g);
    memcpy(out_quotient, quotient.v, 20);
    for (u32 i = 0; i < n_vars; i++) memcpy(out_point + 5 * i, point[i].v, 20);
    memcpy(out_claims, claim_num.v, 20);
    memcpy(out_claims + 5, claim_den.v, 20);
    return LM_OK;
}


// Coefficients of the Lagrange basis polynomials of the nodes 0..d (row a: L_a = prod_{b != a} (X - b) / (a - b), base field),
// built once per degree: the batched AIR sumcheck interpolates every table's round polynomial every round, between two
// device exchanges.
static const std::vector<u32>& lagrange_basis(u32 d) {
    static std::mutex mu;
    static std::map<u32, std::vector<u32>> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(d);
    if (it != cache.end()) return it->second;
    std::vector<u32> L((size_t)(d + 1) * (d + 1), 0);
    for (u32 a = 0; a <= d; a++) {
        std::vector<u32> num{kb::ONE};  // prod_{b != a} (X - b)
        u32 den = kb::ONE;
        for (u32 b = 0; b <= d; b++) {
            if (b == a) continue;
            std::vector<u32> nn(num.size() + 1, 0);
            const u32 mb = kb::neg(kb::to_monty(b));
            for (size_t c = 0; c < num.size(); c++) {
                nn[c + 1] = kb::add(nn[c + 1], num[c]);
                nn[c] = kb::add(nn[c], kb::mul(num[c], mb));
            }
            num.swap(nn);
            den = kb::mul(den, kb::sub(kb::to_monty(a), kb::to_monty(b)));
        }
        const u32 inv_den = kb::inv(den);
        for (size_t c = 0; c < num.size(); c++) L[(size_t)a * (d + 1) + c] = kb::mul(num[c], inv_den);
    }
    return cache.emplace(d, std::move(L)).first->second;
}

// prove_batched_air_sumcheck (air_sumcheck.rs:636-681) + compute_bare_round_poly / process_challenge (:225-292) host side
int lmh_prove_batched_air_sumcheck(lm_ctx* ctx, lmh_prover* p, const lm_air_table* tables, uint32_t n_tables,
                                   const uint32_t alpha[5], const uint32_t* logup_eq16, const uint32_t bus_beta[5],
                                   const uint32_t eta[5], uint32_t* out_point, uint32_t* out_col_evals) {
    if (!ctx || !p || !tables || !n_tables || !alpha || !logup_eq16 || !bus_beta || !eta || !out_point || !out_col_evals)
        return LM_E_INVALID;
    // a session's slice of the pinned result buffer, its flag word and its side stream are keyed by the TABLE id (lm_air_new): the
    // sessions of one batch are launched back to back, so two sessions of the same table would overwrite each other's round sums
    {
        bool seen[3] = {false, false, false};
        for (u32 i = 0; i < n_tables; i++) {
            if (n_tables > 3 || tables[i].table > 2 || seen[tables[i].table]) {
                lm_set_error("lmh_prove_batched_air_sumcheck: at most one session per table (execution, extension_op, poseidon16) in a batch");
                return LM_E_INVALID;
            }
            seen[tables[i].table] = true;
        }
    }
    struct Session {
        lm_air* h = nullptr;
        u32 n_vars = 0, deg = 0;
        std::vector<EF> eq_factor, inv_eq_factor;
        EF sum, mmf;
    };
    std::vector<Session> ss(n_tables);
    auto cleanup = [&](int code) {
        for (Session& s : ss)
            if (s.h) lm_air_free(ctx, s.h);
        return code;
    };
    u32 n_rounds = 0, max_full_degree = 1;
    std::vector<bool> launched(n_tables, false);
    static const bool early_first = getenv("LM_AIR_NO_EARLY_FIRST") == nullptr;
    u32 tallest = 0;
    for (u32 i = 0; i < n_tables; i++) tallest = std::max(tallest, tables[i].log_rows);
    for (u32 i = 0; i < n_tables; i++) {
        const lm_air_table& t = tables[i];
        int rc = lm_air_new(ctx, t.table, t.d_cols, t.log_rows, t.eq_point, alpha, logup_eq16, bus_beta, &ss[i].h);
        if (rc) return cleanup(rc);
        if (t.non_padded_n_rows) {
            if (t.non_padded_n_rows > (1ull << t.log_rows)) {
                lm_set_error("lmh_prove_batched_air_sumcheck: non_padded_n_rows exceeds the table");
                return cleanup(LM_E_INVALID);
            }
            if ((rc = lm_air_set_active_rows(ss[i].h, t.non_padded_n_rows))) return cleanup(rc);
        }
        // the tallest table's first round is the first thing the batch waits for: it goes out the moment its session exists, beside the
        // set-up of the other sessions (virtual columns, eq tables, uploads: ~60 us of stream time in front of it otherwise)
        if (early_first && t.log_rows == tallest) {
            if ((rc = lm_air_round_launch(ctx, ss[i].h))) return cleanup(rc);
            launched[i] = true;
        }
        ss[i].n_vars = t.log_rows;
        ss[i].deg = lm_air_degree(ss[i].h);
        ss[i].eq_factor.resize(t.log_rows);
        for (u32 j = 0; j < t.log_rows; j++) ss[i].eq_factor[j] = ef_load(t.eq_point + 5 * j);
        // 1 / eq_factor[j] for every round of the session with ONE inversion (the per-round inverse sat between two device exchanges)
        {
            const u32 n = t.log_rows;
            std::vector<EF> pre(n + 1);
            pre[0] = kb::ef_one();
            for (u32 j = 0; j < n; j++) pre[j + 1] = kb::ef_mul(pre[j], ss[i].eq_factor[j]);
            ss[i].inv_eq_factor.resize(n);
            bool zero = true;
            for (int k = 0; k < 5; k++) zero = zero && pre[n].v[k] == 0;
            if (zero) {
                for (u32 j = 0; j < n; j++) ss[i].inv_eq_factor[j] = kb::ef_inv(ss[i].eq_factor[j]);
            } else {
                EF run = kb::ef_inv(pre[n]);
                for (u32 j = n; j-- > 0;) {
                    ss[i].inv_eq_factor[j] = kb::ef_mul(run, pre[j]);
                    run = kb::ef_mul(run, ss[i].eq_factor[j]);
                }
            }
        }
        ss[i].sum = ef_load(t.sum);
        ss[i].mmf = kb::ef_one();
        n_rounds = std::max(n_rounds, t.log_rows);
        max_full_degree = std::max(max_full_degree, ss[i].deg + 1);
    }
    const EF eta_e = ef_load(eta);
    std::vector<EF> eta_p(n_tables, kb::ef_one()), k(n_tables, kb::ef_one());
    for (u32 i = 1; i < n_tables; i++) eta_p[i] = kb::ef_mul(eta_p[i - 1], eta_e);
    std::vector<EF> challenges;
    // LM_STAGE_TIMES: how the wall clock of the round loop splits into waiting for the device and host work between exchanges
    const bool clk_on = getenv("LM_STAGE_TIMES") != nullptr;
    double t_wait = 0, t_launch = 0;
    const auto t_loop0 = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    std::vector<u32> launch_order(n_tables);
    for (u32 i = 0; i < n_tables; i++) launch_order[i] = i;
    std::stable_sort(launch_order.begin(), launch_order.end(), [&](u32 a, u32 b) { return tables[a].table > tables[b].table; });
    // A session's FIRST round depends on no challenge of the batch (its eq point and alpha are known before round 0), only its later
    // ones do: the base-field rounds of the tables that join late (Poseidon16: five launches, 0.42 ms on the default workload) are
    // enqueued at once, tallest table first, and run beside the rounds the tallest table does alone; their sums wait in the session's
    // slice of the pinned buffer until the session joins.  (LM_AIR_NO_EARLY_FIRST=1: launch when the session joins, for A/B.)
    if (early_first) {
        std::vector<u32> by_height(n_tables);
        for (u32 i = 0; i < n_tables; i++) by_height[i] = i;
        std::stable_sort(by_height.begin(), by_height.end(), [&](u32 a, u32 b) { return ss[a].n_vars > ss[b].n_vars; });
        for (u32 i : by_height) {
            if (launched[i]) continue;  // (enqueued right behind its lm_air_new, above)
            int rc = lm_air_round_launch(ctx, ss[i].h);
            if (rc) return cleanup(rc);
            launched[i] = true;
        }
    }
    for (u32 round = 0; round < n_rounds; round++) {
        std::vector<EF> combined(max_full_degree + 1, kb::ef_zero());
        std::vector<std::vector<EF>> bare(n_tables);
        // the sessions are independent until the challenge: enqueue every active table's round, then collect
        const auto tl0 = std::chrono::steady_clock::now();
        for (u32 i = 0; i < n_tables; i++)
            if (round >= n_rounds - ss[i].n_vars && !launched[i]) {  // (first round of a session: later ones were launched behind their fold)
                int rc = lm_air_round_launch(ctx, ss[i].h);
                if (rc) return cleanup(rc);
            }
        if (clk_on) t_launch += since(tl0);
        for (u32 i = 0; i < n_tables; i++) {
            Session& s = ss[i];
            const u32 join = n_rounds - s.n_vars;
            const EF w = kb::ef_mul(eta_p[i], k[i]);
            if (round < join) {
                combined[1] = kb::ef_add(combined[1], kb::ef_mul(w, s.sum));
                continue;
            }
            // compute_bare_round_poly: raw sums at z = 0, 2, .., deg from the device
            std::vector<u32> raw((size_t)s.deg * 5);
            const auto tw0 = std::chrono::steady_clock::now();
            int rc = lm_air_round_wait(ctx, s.h, raw.data());
            if (rc) return cleanup(rc);
            if (clk_on) {
                t_wait += since(tw0);
                if (getenv("LM_AIR_DBG")) fprintf(stderr, "#     air round %u table %u: wait %.1f us\n", round, i, since(tw0) * 1e3);
            }
            const u32 d = s.deg;
            std::vector<EF> ev(d + 1);
            ev[0] = kb::ef_mul(ef_load(&raw[0]), s.mmf);
            for (u32 z = 2; z <= d; z++) ev[z] = kb::ef_mul(ef_load(&raw[(size_t)(z - 1) * 5]), s.mmf);
            const EF eq_alpha = s.eq_factor.back();
            ev[1] = kb::ef_mul(kb::ef_sub(s.sum, kb::ef_mul(kb::ef_sub(kb::ef_one(), eq_alpha), ev[0])), s.inv_eq_factor[s.eq_factor.size() - 1]);
            // DensePolynomial::lagrange_interpolation on the points 0..d: the basis polynomials depend on d only (lagrange_basis)
            std::vector<EF> coeffs(d + 1, kb::ef_zero());
            {
                const std::vector<u32>& L = lagrange_basis(d);
                for (u32 a = 0; a <= d; a++)
                    for (u32 c = 0; c <= d; c++) coeffs[c] = kb::ef_add(coeffs[c], kb::ef_mul_base(ev[a], L[(size_t)a * (d + 1) + c]));
            }
            bare[i] = coeffs;
            // expand_bare_to_full (fiat-shamir/src/utils.rs:30-41)
            const EF oma = kb::ef_sub(kb::ef_one(), eq_alpha), tam = kb::ef_sub(kb::ef_dbl(eq_alpha), kb::ef_one());
            std::vector<EF> full(d + 2);
            full[0] = kb::ef_mul(oma, coeffs[0]);
            for (u32 c = 1; c <= d; c++) full[c] = kb::ef_add(kb::ef_mul(oma, coeffs[c]), kb::ef_mul(tam, coeffs[c - 1]));
            full[d + 1] = kb::ef_mul(tam, coeffs[d]);
            for (u32 c = 0; c < d + 2; c++) combined[c] = kb::ef_add(combined[c], kb::ef_mul(w, full[c]));
        }
        add_sumcheck_poly(p, combined, nullptr);
        std::vector<EF> cv;
        if (!sample_vec(p, 1, cv)) return cleanup(LM_E_INVALID);
        const EF ch = cv[0];
        challenges.push_back(ch);
        // The sessions' chains (fold -> next round kernel) run on their own streams, but their launch calls are issued by this one
        // thread, ~4-5 us each: a session's next round is enqueued right behind its fold, the session with the longest small-round
        // kernel first (Poseidon16, ExtensionOp, execution) — with all folds first and all rounds after, the last round kernel started
        // six launch calls after the challenge was known.
        for (u32 oi = 0; oi < n_tables; oi++) {
            const u32 i = launch_order[oi];
            Session& s = ss[i];
            const u32 join = n_rounds - s.n_vars;
            if (round < join) {  // (not joined yet: its first round may already be in flight)
                k[i] = kb::ef_mul(k[i], ch);
                continue;
            }
            launched[i] = false;
            // process_challenge
            const EF a = s.eq_factor.back();
            const EF eq_eval = kb::ef_add(kb::ef_mul(kb::ef_sub(kb::ef_one(), a), kb::ef_sub(kb::ef_one(), ch)), kb::ef_mul(a, ch));
            EF bv = kb::ef_zero();
            for (size_t c = bare[i].size(); c-- > 0;) bv = kb::ef_add(kb::ef_mul(bv, ch), bare[i][c]);
            s.sum = kb::ef_mul(bv, eq_eval);
            s.mmf = kb::ef_mul(s.mmf, eq_eval);
            int rc = lm_air_bind(ctx, s.h, ch.v);
            if (rc) return cleanup(rc);
            s.eq_factor.pop_back();
            if (round + 1 < n_rounds) {
                if ((rc = lm_air_round_launch(ctx, s.h))) return cleanup(rc);
                launched[i] = true;
            }
        }
    }
    if (clk_on) {
        const double total = since(t_loop0);
        fprintf(stderr, "#   air: %u rounds %.3f ms: waiting for round sums %.3f ms, launch calls %.3f ms, host (interpolation, transcript, bind calls) %.3f ms\n",
                n_rounds, total, t_wait, t_launch, total - t_wait - t_launch);
    }
    u32* oe = out_col_evals;
    for (u32 i = 0; i < n_tables; i++) {  // (the three publications side by side, then collected in transcript order)
        int rc = lm_air_final_evals_begin(ctx, ss[i].h);
        if (rc) return cleanup(rc);
    }
    for (u32 i = 0; i < n_tables; i++) {
        const u32 ne = lm_air_n_evals(ss[i].h);
        int rc = lm_air_final_evals_end(ctx, ss[i].h, oe);
        if (rc) return cleanup(rc);
        add_base(p, oe, (u64)ne * 5);  // add_extension_scalars(&col_evals)
        oe += (size_t)ne * 5;
    }
    for (u32 i = 0; i < n_rounds; i++) memcpy(out_point + 5 * i, challenges[i].v, 20);
    return cleanup(LM_OK);
}


}  // extern "C" (helpers below have C++ linkage)

namespace {
struct DevBuf {  // RAII device allocation
    lm_ctx* ctx;
    u32* p = nullptr;
    DevBuf(lm_ctx* c) : ctx(c) {}
    ~DevBuf() {
        if (p) lm_free(ctx, p);
    }
};
}  // namespace

extern "C" {

uint32_t lmh_stacked_n_vars(const lm_execution_trace* t) {
    u32 mx = std::max(t->tables[0].log_rows, std::max(t->tables[1].log_rows, t->tables[2].log_rows));
    u64 total = (2ull << t->log_memory) + (1ull << std::max(t->log_bytecode, mx));
    for (int k = 0; k < 3; k++) total += (u64)kVmTables[k].n_columns << t->tables[k].log_rows;
    return log2_ceil_u64(total);
}

int lmh_prover_stage_times(const lmh_prover* p, double out_ms[LMH_N_STAGES]) {
    if (!p || !out_ms) return LM_E_INVALID;
    memcpy(out_ms, p->stage_ms, sizeof p->stage_ms);
    return LM_OK;
}

int lmh_prove_execution(lm_ctx* ctx, lmh_prover* p, const lm_execution_trace* tr, const lm_whir_config* cfg) {
    if (!ctx || !p || !tr || !cfg) return LM_E_INVALID;
    int rc;
    if ((rc = lm_bind_thread(ctx))) return rc;
    StageClock clk(ctx, p->stage_ms);
    int order[3];
    const u32 log_rows[3] = {tr->tables[0].log_rows, tr->tables[1].log_rows, tr->tables[2].log_rows};
    lmh::sorted_tables(log_rows, order);
    const u32 log_mem = tr->log_memory, log_bc = tr->log_bytecode;
    auto invalid = [](const char* why) {
        lm_set_error("lmh_prove_execution: %s", why);
        return LM_E_INVALID;
    };
    // assertions of stack_polynomials_and_commit (stacked_pcs.rs:108-112) and of prove_execution (prove_execution.rs:41-46,
    // 64-76: memory >= bytecode and >= 2^MIN_LOG_MEMORY_SIZE, table heights within [MIN_LOG_N_ROWS_PER_TABLE, limit])
    if (log_mem < tr->tables[0].log_rows || tr->tables[0].log_rows < tr->tables[order[0]].log_rows)
        return invalid("memory must be at least as tall as the execution table, which must be the tallest table");
    if (log_mem < log_bc) return invalid("memory must be at least as large as the bytecode (prove_execution.rs:41-45)");
    if (log_mem < lmh::MIN_LOG_MEMORY_SIZE || log_mem > lmh::MAX_LOG_MEMORY_SIZE) return invalid("log_memory outside [MIN_LOG_MEMORY_SIZE, MAX_LOG_MEMORY_SIZE]");
    if (log_bc < lmh::MIN_BYTECODE_LOG_SIZE) return invalid("bytecode smaller than 2^MIN_BYTECODE_LOG_SIZE");
    for (int t = 0; t < 3; t++)
        if (log_rows[t] < lmh::MIN_LOG_N_ROWS_PER_TABLE || log_rows[t] > lmh::max_log_n_rows_per_table(t))
            return invalid("table height outside [MIN_LOG_N_ROWS_PER_TABLE, max_log_n_rows_per_table]");
    if (tr->public_memory_size == 0 || (tr->public_memory_size & (tr->public_memory_size - 1)))
        return invalid("public_memory_size must be a power of two (log2_strict_usize, prove_execution.rs:225)");
    if (!lmh::rate_ok(tr->log_inv_rate)) return invalid("log_inv_rate outside [MIN_WHIR_LOG_INV_RATE, MAX_WHIR_LOG_INV_RATE]");
    const u32 stacked_n_vars = lmh_stacked_n_vars(tr);
    if (cfg->num_variables != stacked_n_vars || cfg->starting_log_inv_rate != tr->log_inv_rate)
        return invalid("the WhirConfig is not for this trace (num_variables / starting_log_inv_rate)");
    // ---- Fiat-Shamir preamble (prove_execution.rs:47-63) ----
    p->ch.observe_many(tr->public_input, tr->n_public_input);
    {
        u32 st[16];
        memcpy(st, tr->bytecode_hash, 32);
        for (int i = 0; i < 8; i++) st[8 + i] = kb::to_monty(kSnarkDomainSep[i]);
        lmh::host_compress(st);  // poseidon16_compress_pair(bytecode.hash, SNARK_DOMAIN_SEP)
        p->ch.observe_many(st, 8);
        u32 dims[6] = {kb::to_monty(tr->log_inv_rate), kb::to_monty(log_mem), kb::to_monty(tr->n_public_input),
                       kb::to_monty(tr->tables[0].log_rows), kb::to_monty(tr->tables[1].log_rows), kb::to_monty(tr->tables[2].log_rows)};
        add_base(p, dims, 6);
    }
    // ---- stack_polynomials_and_commit (stacked_pcs.rs:99-157) ----
    const u64 mem = 1ull << log_mem;
    DevBuf poly(ctx);
    u32* const in_place = tr->d_stacked;  // the trace already lives in its committed layout (leanmultisig_host.h)
    if (in_place) {
        u64 at = 2 * mem + std::max(1ull << tr->tables[order[0]].log_rows, 1ull << log_bc);
        bool ok = tr->d_memory == in_place && !tr->d_memory_acc && !tr->d_bytecode_acc;
        for (int k = 0; k < 3 && ok; k++) {
            const int t = order[k];
            for (u32 c = 0; c < kVmTables[t].n_columns; c++, at += 1ull << tr->tables[t].log_rows) ok = ok && tr->tables[t].d_cols[c] == in_place + at;
        }
        if (!ok) return invalid("d_stacked: memory / committed columns are not at their stacked offsets (or access counters were supplied)");
    } else if ((rc = lm_malloc(ctx, 1ull << stacked_n_vars, &poly.p)))
        return rc;
    u32* const d_poly = in_place ? in_place : poly.p;
    std::vector<const u32*> srcs;
    std::vector<u64> offs, lens;
    auto place = [&](const u32* src, u64 at, u64 n) {
        srcs.push_back(src);
        offs.push_back(at);
        lens.push_back(n);
    };
    // access counters (prove_execution.rs:90-110) unless the caller already has them
    (void)lm_access_errors(ctx, 1);  // reset the sticky count of lookup rows that leave the image
    DevBuf mem_acc(ctx), bc_acc(ctx);
    const u32* d_memory_acc = tr->d_memory_acc;
    const u32* d_bytecode_acc = tr->d_bytecode_acc;
    if (!d_memory_acc) {
        std::vector<const u32*> idx;
        std::vector<u64> rows;
        std::vector<u32> nv;
        for (int t = 0; t < 3; t++)
            for (u32 l = 0; l < kVmTables[t].n_lookups; l++) {
                idx.push_back(tr->tables[t].d_cols[kVmTables[t].lookups[l].index]);
                rows.push_back(1ull << tr->tables[t].log_rows);
                nv.push_back(kVmTables[t].lookups[l].n_values);
            }
        u32* dst = in_place ? in_place + mem : nullptr;
        if (!dst) {
            if ((rc = lm_malloc(ctx, mem, &mem_acc.p))) return rc;
            dst = mem_acc.p;
        }
        if ((rc = lm_access_counts(ctx, dst, mem, (u32)idx.size(), idx.data(), rows.data(), nv.data()))) return rc;
        d_memory_acc = dst;
    }
    if (!d_bytecode_acc) {
        const u32* pc = tr->tables[0].d_cols[0];  // COL_PC
        const u64 rows = 1ull << tr->tables[0].log_rows;
        const u32 one = 1;
        u32* dst = in_place ? in_place + 2 * mem : nullptr;
        if (!dst) {
            if ((rc = lm_malloc(ctx, 1ull << log_bc, &bc_acc.p))) return rc;
            dst = bc_acc.p;
        }
        if ((rc = lm_access_counts(ctx, dst, 1ull << log_bc, 1, &pc, &rows, &one))) return rc;
        d_bytecode_acc = dst;
    }
    place(tr->d_memory, 0, mem);
    place(d_memory_acc, mem, mem);
    u64 off = 2 * mem;
    place(d_bytecode_acc, off, 1ull << log_bc);
    off += std::max(1ull << tr->tables[order[0]].log_rows, 1ull << log_bc);
    for (int k = 0; k < 3; k++) {
        const int t = order[k];
        for (u32 c = 0; c < kVmTables[t].n_columns; c++) {
            place(tr->tables[t].d_cols[c], off, 1ull << tr->tables[t].log_rows);
            off += 1ull << tr->tables[t].log_rows;
        }
    }
    if (!in_place && (rc = lm_stack_columns(ctx, poly.p, 1ull << stacked_n_vars, (u32)srcs.size(), srcs.data(), offs.data(), lens.data()))) return rc;
    lmh_witness* wit = nullptr;
    clk.mark("stack", LMH_STAGE_COMMIT);
    if ((rc = lmh_whir_commit(ctx, p, cfg, d_poly, off, &wit))) return rc;
    clk.mark("whir_commit", LMH_STAGE_COMMIT);
    // (the root has been published behind the counting kernels on the same stream: the stream is idle, the count final)
    if (const u32 n_bad = lm_access_errors(ctx, 0)) {
        lm_set_error("%u lookup rows address words outside the memory / bytecode image (the reference panics on these)", n_bad);
        lmh_witness_free(ctx, wit);
        return LM_E_INVALID;
    }
    auto fail = [&](int code) {
        if (wit) lmh_witness_free(ctx, wit);
        return code;
    };
    // ---- logup (prove_execution.rs:123-137, logup.rs:27-308) ----
    std::vector<EF> cv, alphas;
    if (!sample_vec(p, 1, cv)) return fail(LM_E_INVALID);
    const EF logup_c = cv[0];
    p->ch.duplex();
    if (!sample_vec(p, 4, alphas)) return fail(LM_E_INVALID);
    EF aeq[16];  // eval_eq(&logup_alphas): index bits MSB first
    for (u32 i = 0; i < 16; i++) {
        EF a = kb::ef_one();
        for (u32 j = 0; j < 4; j++) a = kb::ef_mul(a, ((i >> (3 - j)) & 1) ? alphas[j] : kb::ef_sub(kb::ef_one(), alphas[j]));
        aeq[i] = a;
    }
    std::vector<lm_logup_section> secs;
    auto new_sec = [&](u64 offset, u32 log_len, u32 num_mode, const u32* num_col, int den_sign, u32 domsep) {
        lm_logup_section s;
        memset(&s, 0, sizeof s);
        s.out_offset = offset;
        s.log_len = log_len;
        s.num_mode = num_mode;
        s.d_num_col = num_col;
        s.den_sign = den_sign;
        s.domsep = domsep;
        secs.push_back(s);
        return &secs.back();
    };
    auto add_data = [&](lm_logup_section* s, const u32* col, u32 stride, u32 add) {
        s->d_data[s->n_data] = col;
        s->stride[s->n_data] = stride;
        s->add[s->n_data] = add;
        s->n_data++;
    };
    u64 loff = 0;
    {
        lm_logup_section* s = new_sec(loff, log_mem, 3, d_memory_acc, -1, 0);  // logup.rs:94-109
        add_data(s, tr->d_memory, 1, 0);
        add_data(s, nullptr, 0, 0);
        loff += mem;
        s = new_sec(loff, log_bc, 3, d_bytecode_acc, -1, 2);  // :111-125
        for (u32 k = 0; k < 12; k++) add_data(s, tr->d_bytecode + k, 16, 0);
        add_data(s, nullptr, 0, 0);
        loff += std::max(1ull << log_bc, 1ull << tr->tables[order[0]].log_rows);
    }
    for (int k = 0; k < 3; k++) {
        const int t = order[k];
        const VmTableDef& def = kVmTables[t];
        const u32 lr = tr->tables[t].log_rows;
        const u32* const* cols = tr->tables[t].d_cols;
        if (t == 0) {  // :141-156
            lm_logup_section* s = new_sec(loff, lr, 1, nullptr, -1, 2);
            for (u32 c = 0; c < 12; c++) add_data(s, cols[8 + c], 1, 0);
            add_data(s, cols[0], 1, 0);
            loff += 1ull << lr;
        }
        {  // bus, :158-176
            lm_logup_section* s = new_sec(loff, lr, def.pull ? 3 : 2, cols[def.selector], +1, 1);
            for (u32 c = 0; c < 4; c++) add_data(s, cols[def.bus_data[c]], 1, 0);
            loff += 1ull << lr;
        }
        for (u32 l = 0; l < def.n_lookups; l++)  // :178-199
            for (u32 i = 0; i < def.lookups[l].n_values; i++) {
                lm_logup_section* s = new_sec(loff, lr, 1, nullptr, -1, 0);
                add_data(s, cols[def.lookups[l].first_value + i], 1, 0);
                add_data(s, cols[def.lookups[l].index], 1, i);
                loff += 1ull << lr;
            }
    }
    const u32 gkr_n_vars = log2_ceil_u64(loff);
    DevBuf nums(ctx), dens(ctx);
    if ((rc = lm_malloc(ctx, 1ull << gkr_n_vars, &nums.p))) return fail(rc);
    if ((rc = lm_malloc(ctx, 5ull << gkr_n_vars, &dens.p))) return fail(rc);
    u64 gkr_active = 0;  // = loff: the tail up to 2^gkr_n_vars is neutral padding, never materialised
    if ((rc = lm_logup_build_active(ctx, secs.data(), (u32)secs.size(), logup_c.v, aeq[0].v, gkr_n_vars, nums.p, dens.p, &gkr_active))) return fail(rc);
    clk.mark("logup_fill", LMH_STAGE_LOGUP_FILL);
    u32 quotient[5], claims[10];
    std::vector<u32> gkr_pt((size_t)gkr_n_vars * 5);
    if ((rc = lmh_prove_gkr_quotient_active(ctx, p, nums.p, dens.p, gkr_n_vars, gkr_active, quotient, gkr_pt.data(), claims))) return fail(rc);
    clk.mark("logup_gkr", LMH_STAGE_GKR);
    if (quotient[0] | quotient[1] | quotient[2] | quotient[3] | quotient[4]) {  // assert_eq!(sum, ZERO)
        lm_set_error("logup sum != 0: the witness is inconsistent (a lookup reads a value the memory / bytecode does not hold, or the access counters are wrong)");
        return fail(LM_E_INVALID);
    }
    auto from_end = [&](u32 n) { return gkr_pt.data() + (size_t)(gkr_n_vars - n) * 5; };
    // column evaluations (logup.rs:224-308)
    // All of them are evaluations at suffixes of the GKR point and nothing is sampled in between: the six device evaluations are
    // enqueued back to back (lm_results_defer_begin: results at increasing offsets of the pinned buffer, no wait per call), collected
    // once, and then observed in the reference's order.
    EF value_memory_acc, value_memory, value_bytecode_acc;
    std::vector<u32> want_of[3], ev_of[3];
    size_t i_sel_of[3], i_lk_of[3];
    if ((rc = lm_results_defer_begin(ctx))) return fail(rc);
    auto fail_deferred = [&](int code) {
        (void)lm_results_defer_end(ctx);
        return fail(code);
    };
    if ((rc = lm_mle_eval(ctx, d_memory_acc, 0, log_mem, 1, 0, from_end(log_mem), value_memory_acc.v))) return fail_deferred(rc);
    if ((rc = lm_mle_eval(ctx, tr->d_memory, 0, log_mem, 1, 0, from_end(log_mem), value_memory.v))) return fail_deferred(rc);
    if ((rc = lm_mle_eval(ctx, d_bytecode_acc, 0, log_bc, 1, 0, from_end(log_bc), value_bytecode_acc.v))) return fail_deferred(rc);
    for (int k = 0; k < 3; k++) {
        const int t = order[k];
        const VmTableDef& def = kVmTables[t];
        const u32 lr = tr->tables[t].log_rows;
        const u32* const* cols = tr->tables[t].d_cols;
        // gather the column list in transcript order, evaluate them in one batch
        std::vector<u32>& want = want_of[t];
        if (t == 0) {
            want.push_back(0);
            for (u32 c = 0; c < 12; c++) want.push_back(8 + c);
        }
        i_sel_of[t] = want.size();
        want.push_back(def.selector);
        for (u32 c = 0; c < 4; c++) want.push_back(def.bus_data[c]);
        i_lk_of[t] = want.size();
        for (u32 l = 0; l < def.n_lookups; l++) {
            want.push_back(def.lookups[l].index);
            for (u32 i = 0; i < def.lookups[l].n_values; i++) want.push_back(def.lookups[l].first_value + i);
        }
        std::vector<const u32*> ptrs;
        for (u32 c : want) ptrs.push_back(cols[c]);
        ev_of[t].resize(want.size() * 5);
        if ((rc = lm_mle_eval_cols(ctx, ptrs.data(), (u32)ptrs.size(), lr, from_end(lr), ev_of[t].data()))) return fail_deferred(rc);
    }
    if ((rc = lm_results_defer_end(ctx))) return fail(rc);
    add_base(p, value_memory_acc.v, 5);
    add_base(p, value_memory.v, 5);
    add_base(p, value_bytecode_acc.v, 5);
    std::vector<ColVal> columns_values[3];
    EF bus_num[3], bus_den[3];
    for (int k = 0; k < 3; k++) {
        const int t = order[k];
        const VmTableDef& def = kVmTables[t];
        const std::vector<u32>&want = want_of[t], &ev = ev_of[t];
        const size_t i_sel = i_sel_of[t], i_lk = i_lk_of[t];
        auto E = [&](size_t i) { return ef_load(&ev[5 * i]); };
        if (t == 0) {
            add_base(p, &ev[0], 5);       // eval_on_pc
            add_base(p, &ev[5], 12 * 5);  // instr_evals
            for (size_t i = 0; i < 13; i++) columns_values[t].push_back({want[i], E(i)});
        }
        EF esel = E(i_sel);
        if (def.pull) esel = kb::ef_neg(esel);  // * direction.to_field_flag()
        add_base(p, esel.v, 5);
        EF fp = aeq[15];  // finger_print(LOGUP_PRECOMPILE_DOMAINSEP = 1, bus data evals)
        for (u32 c = 0; c < 4; c++) fp = kb::ef_add(fp, kb::ef_mul(aeq[c], E(i_sel + 1 + c)));
        const EF edata = kb::ef_add(logup_c, fp);
        add_base(p, edata.v, 5);
        bus_num[t] = esel;
        bus_den[t] = edata;
        for (size_t i = i_lk; i < want.size(); i++) {
            add_base(p, &ev[5 * i], 5);
            columns_values[t].push_back({want[i], E(i)});
        }
    }
    clk.mark("column_evaluations", LMH_STAGE_COLUMN_EVALS);
    // ---- AIR (prove_execution.rs:152-223) ----
    std::vector<EF> tmp;
    if (!sample_vec(p, 1, tmp)) return fail(LM_E_INVALID);
    const EF bus_beta = tmp[0];
    p->ch.duplex();
    if (!sample_vec(p, 1, tmp)) return fail(LM_E_INVALID);
    const EF air_alpha = tmp[0];
    p->ch.duplex();
    if (!sample_vec(p, 1, tmp)) return fail(LM_E_INVALID);
    const EF air_eta = tmp[0];
    lm_air_table at[3];
    u32 n_evals_total = 0;
    for (int k = 0; k < 3; k++) {
        const int t = order[k];
        at[k].table = (u32)t;
        at[k].log_rows = tr->tables[t].log_rows;
        at[k].d_cols = tr->tables[t].d_cols;
        at[k].eq_point = from_end(tr->tables[t].log_rows);
        at[k].non_padded_n_rows = tr->tables[t].non_padded_n_rows;
        const EF dir = kVmTables[t].pull ? kb::ef_neg(kb::ef_one()) : kb::ef_one();
        const EF bfv = kb::ef_add(kb::ef_mul(bus_num[t], dir), kb::ef_mul(bus_beta, kb::ef_sub(bus_den[t], logup_c)));
        memcpy(at[k].sum, bfv.v, 20);
        n_evals_total += kVmTables[t].n_columns + kVmTables[t].n_shift;
    }
    const u32 n_max = tr->tables[order[0]].log_rows;
    std::vector<u32> air_point((size_t)n_max * 5), col_evals((size_t)n_evals_total * 5);
    if ((rc = lmh_prove_batched_air_sumcheck(ctx, p, at, 3, air_alpha.v, aeq[0].v, bus_beta.v, air_eta.v, air_point.data(), col_evals.data())))
        return fail(rc);
    clk.mark("batched_air_sumcheck", LMH_STAGE_AIR);
    // ---- public memory, statements (:225-260; stacked_pcs.rs:40-97) ----
    const u32 lpm = log2_ceil_u64(tr->public_memory_size);
    std::vector<EF> pm_pt;
    if (!sample_vec(p, lpm, pm_pt)) return fail(LM_E_INVALID);
    EF pm_eval;
    if ((rc = lm_mle_eval(ctx, tr->d_memory, 0, lpm, 1, 0, lpm ? pm_pt[0].v : nullptr, pm_eval.v))) return fail(rc);
    lmh::Statements S;
    lmh::assemble_statements(S, log_rows, log_mem, log_bc, tr->ending_pc, gkr_pt.data(), gkr_n_vars, value_memory, value_memory_acc,
                             value_bytecode_acc, lpm ? pm_pt[0].v : nullptr, lpm, pm_eval, columns_values, air_point.data(), col_evals.data());
    clk.mark("statement_assembly", LMH_STAGE_WHIR_PROVE);
    std::vector<u32> out_point((size_t)stacked_n_vars * 5);
    lmh_witness* w = wit;
    wit = nullptr;  // consumed by lmh_whir_prove
    rc = lmh_whir_prove(ctx, p, cfg, S.sts.data(), (u32)S.sts.size(), S.pts.data(), S.pts.size() / 5, S.sels.data(), S.vals.data(),
                        S.sels.size(), w, d_poly, out_point.data());
    clk.mark("whir_open", LMH_STAGE_WHIR_PROVE);
    return rc;
}


// ---- pad_table (lean_prover/src/trace_gen.rs:170-191) ---------------------------------------------------------------------
uint32_t lmh_table_log_rows(uint64_t n_rows) {  // log2_ceil(h + 1).max(MIN_LOG_N_ROWS_PER_TABLE)
    u32 l = 0;
    while ((1ull << l) < n_rows + 1) l++;
    return std::max<u32>(l, lmh::MIN_LOG_N_ROWS_PER_TABLE);
}
}  // extern "C"
int lmh::pad_table(lm_ctx* ctx, uint32_t table, uint32_t* const* d_cols, uint64_t n_rows, uint32_t log_rows, uint32_t zero_vec_ptr,
                   uint32_t null_hash_ptr, uint32_t ending_pc, bool with_virtual) {
    if (!ctx || !d_cols || table > 2 || log_rows > 30 || n_rows >= (1ull << log_rows)) {
        lm_set_error("lmh_pad_table: bad arguments (a table needs at least one padding row: n_rows < 2^log_rows)");
        return LM_E_INVALID;
    }
    const u32 n_cols = lmh::kVmTables[table].n_columns;
    std::vector<u32> row(n_cols, 0);  // canonical values
    if (table == 0) {  // execution/mod.rs:59-74 (committed columns; nu_a, nu_b are virtual)
        row[0] = ending_pc;                                   // COL_PC
        row[2] = row[3] = row[4] = zero_vec_ptr;             // COL_MEM_ADDRESS_A/B/C
        row[8] = 1;                                           // COL_OPERAND_A
        row[9] = ending_pc;                                   // COL_OPERAND_B: the jump destination
        row[11] = row[12] = 1;                                // COL_FLAG_A, COL_FLAG_B
        row[14] = 1;                                          // COL_FLAG_C_FP
        row[17] = 1;                                          // COL_JUMP
    } else if (table == 1) {  // extension_op/mod.rs:125-134
        row[1] = 1;                                           // COL_START
        row[2] = 1;                                           // COL_LEN
        row[6] = row[7] = row[13] = zero_vec_ptr;            // COL_IDX_A, COL_IDX_B, COL_IDX_RES
    } else {  // poseidon_16/mod.rs:182-205: flags 0, inputs 0; the 84 derived columns are left to lm_poseidon_trace
        row[1] = zero_vec_ptr;                                // index_b (POSEIDON_16_COL_INDEX_INPUT_RIGHT)
        row[2] = null_hash_ptr;                               // index_res
        row[6] = zero_vec_ptr;                                // effective_index_left_first
        row[7] = zero_vec_ptr + 4;                            // effective_index_left_second (+ HALF_DIGEST_LEN)
    }
    // the virtual columns behind the committed ones (bus data of the padding rows): execution nu_a / nu_b / ... (execution/mod.rs:59-74),
    // ExtensionOp and Poseidon16 precompile-data columns — filled when the caller's pointer array carries them (n_total entries)
    const u32 n_total = lmh::kVmTables[table].n_total;
    std::vector<u32> vals(n_total, 0);
    for (u32 c = 0; c < n_cols; c++) {
        if (row[c] >= kb::P) {
            lm_set_error("lmh_pad_table: value out of the field");
            return LM_E_INVALID;
        }
        vals[c] = kb::to_monty(row[c]);
    }
    u32 n_fill = n_cols;
    if (with_virtual) {
        if (table == 0) vals[21] = kb::to_monty(1), vals[22] = kb::to_monty(ending_pc);
        if (table == 1) vals[30] = kb::to_monty(64);
        if (table == 2) vals[109] = kb::to_monty(zero_vec_ptr), vals[110] = kb::to_monty(1);
        n_fill = n_total;
    }
    for (u32 c = 0; c < n_fill; c++)
        if (!d_cols[c]) {
            lm_set_error("lmh_pad_table: column %u is null", c);
            return LM_E_INVALID;
        }
    return lm_fill_columns(ctx, d_cols, vals.data(), n_fill, n_rows, (1ull << log_rows) - n_rows);
}
extern "C" {
int lmh_pad_table(lm_ctx* ctx, uint32_t table, uint32_t* const* d_cols, uint64_t n_rows, uint32_t log_rows, uint32_t zero_vec_ptr,
                  uint32_t null_hash_ptr, uint32_t ending_pc) {
    return lmh::pad_table(ctx, table, d_cols, n_rows, log_rows, zero_vec_ptr, null_hash_ptr, ending_pc, false);
}

}  // extern "C"
