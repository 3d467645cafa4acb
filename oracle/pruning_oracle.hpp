// TEST INFRASTRUCTURE ONLY — CPU restatement of Merkle-path pruning (crates/backend/fiat-shamir/src/merkle_pruning.rs:18-170):
// MerklePaths::prune (:18-86), PrunedMerklePaths::restore (:88-170) and Proof::proof_size_fe (transcript.rs:39-53).
// Parity: the reference's own tests for this file are round trips on synthetic trees (merkle_pruning.rs:172-400); the same
// round trips run in tests/test_oracle_pruning.py.  The reference serialises the pruned proof with postcard + lz4
// (type_1_aggregation.rs:81-89); that byte format is NOT restated — blobs here are u32 words:
//   un-pruned  [T][transcript x T][M] M x {idx_lo, idx_hi, leaf_len, path_len, leaf.., path..}
//   pruned     [T][transcript x T][B] B x {height, n_trailing_zeros, n_orig, original_order x n_orig, n_paths,
//                                          n_paths x {idx_lo, idx_hi, leaf_len, leaf.., n_sib, sib x 8 n_sib}}
// One batch = one hint_merkle_paths call (the openings of one commitment at one query set).
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>
#include "kb_oracle.hpp"

namespace orc {

struct PathU {  // MerklePath<F, F>
    uint64_t leaf_index;
    std::vector<uint32_t> leaf;
    std::vector<uint32_t> siblings;  // 8 words per level, bottom-up
};
struct PrunedBatch {  // PrunedMerklePaths<F, F>
    uint32_t height = 0, n_trailing_zeros = 0;
    std::vector<uint32_t> original_order;
    std::vector<uint64_t> index;
    std::vector<std::vector<uint32_t>> leaf, sib;
};

static inline size_t lca_level(uint64_t a, uint64_t b) {  // :14-16
    uint64_t x = a ^ b;
    size_t l = 0;
    while (x) {
        l++;
        x >>= 1;
    }
    return l;
}

static inline PrunedBatch prune_batch(const std::vector<PathU>& paths) {  // :18-86
    if (paths.empty()) throw std::runtime_error("prune: empty batch");
    PrunedBatch out;
    out.height = (uint32_t)(paths[0].siblings.size() / 8);
    std::vector<size_t> ord(paths.size());
    for (size_t i = 0; i < ord.size(); i++) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return paths[a].leaf_index < paths[b].leaf_index; });
    out.original_order.assign(paths.size(), 0);
    std::vector<const PathU*> ded;
    for (size_t o : ord) {
        if (!ded.empty() && ded.back()->leaf_index == paths[o].leaf_index) {
            out.original_order[o] = (uint32_t)(ded.size() - 1);
        } else {
            out.original_order[o] = (uint32_t)ded.size();
            ded.push_back(&paths[o]);
        }
    }
    const size_t leaf_len = ded[0]->leaf.size();
    size_t tz = 0;
    for (size_t off = leaf_len; off-- > 0;) {
        bool any = false;
        for (const PathU* p : ded) any = any || p->leaf[off] != 0;
        if (any) break;
        tz++;
    }
    out.n_trailing_zeros = (uint32_t)tz;
    for (size_t i = 0; i < ded.size(); i++) {
        const uint64_t idx = ded[i]->leaf_index;
        const size_t levels = i == 0 ? out.height : lca_level(ded[i - 1]->leaf_index, idx);
        const bool has_skip = i + 1 < ded.size();
        const size_t skip = has_skip ? lca_level(idx, ded[i + 1]->leaf_index) - 1 : 0;
        std::vector<uint32_t> s;
        for (size_t lvl = 0; lvl < levels; lvl++) {
            if (has_skip && lvl == skip) continue;
            s.insert(s.end(), ded[i]->siblings.begin() + 8 * lvl, ded[i]->siblings.begin() + 8 * lvl + 8);
        }
        out.index.push_back(idx);
        out.sib.push_back(std::move(s));
        out.leaf.emplace_back(ded[i]->leaf.begin(), ded[i]->leaf.end() - (ptrdiff_t)tz);
    }
    return out;
}

// :88-170; throws on malformed input (the reference returns None)
static inline std::vector<PathU> restore_batch(PrunedBatch b) {
    const size_t n = b.index.size(), h = b.height;
    if (h >= 32 || b.n_trailing_zeros > 1024 || n == 0) throw std::runtime_error("restore: bad header");
    for (auto& d : b.leaf) d.resize(d.size() + b.n_trailing_zeros, 0);
    auto levels = [&](size_t i) { return i == 0 ? h : lca_level(b.index[i - 1], b.index[i]); };
    auto has_skip = [&](size_t i) { return i + 1 < n; };
    auto skip = [&](size_t i) { return lca_level(b.index[i], b.index[i + 1]) - 1; };
    std::vector<std::vector<uint32_t>> sub(n);  // subtree hashes per path, 8 words per level (level 0 = leaf hash)
    auto sibling_at = [&](size_t i, size_t lvl, size_t& cursor, uint32_t out[8]) {
        if (has_skip(i) && skip(i) == lvl) {
            if (sub[i + 1].size() < 8 * (lvl + 1)) throw std::runtime_error("restore: missing subtree hash");
            std::copy(sub[i + 1].begin() + 8 * lvl, sub[i + 1].begin() + 8 * lvl + 8, out);
        } else {
            if (cursor + 8 > b.sib[i].size()) throw std::runtime_error("restore: not enough siblings");
            std::copy(b.sib[i].begin() + cursor, b.sib[i].begin() + cursor + 8, out);
            cursor += 8;
        }
    };
    for (size_t i = n; i-- > 0;) {  // backward pass
        if (b.index[i] >> h) throw std::runtime_error("restore: index out of range");
        uint32_t hash[8];
        hash_slice(b.leaf[i].data(), b.leaf[i].size(), hash);
        sub[i].insert(sub[i].end(), hash, hash + 8);
        size_t cursor = 0;
        for (size_t lvl = 0; lvl < levels(i); lvl++) {
            uint32_t s[8], nxt[8];
            sibling_at(i, lvl, cursor, s);
            if ((b.index[i] >> lvl) & 1)
                compress_pair(s, hash, nxt);
            else
                compress_pair(hash, s, nxt);
            std::copy(nxt, nxt + 8, hash);
            sub[i].insert(sub[i].end(), hash, hash + 8);
        }
    }
    std::vector<PathU> restored;
    for (size_t i = 0; i < n; i++) {  // forward pass
        PathU p;
        p.leaf_index = b.index[i];
        p.leaf = b.leaf[i];
        size_t cursor = 0;
        for (size_t lvl = 0; lvl < levels(i); lvl++) {
            uint32_t s[8];
            sibling_at(i, lvl, cursor, s);
            p.siblings.insert(p.siblings.end(), s, s + 8);
        }
        if (!restored.empty()) {
            const auto& prev = restored.back().siblings;
            if (prev.size() < 8 * levels(i)) throw std::runtime_error("restore: previous path too short");
            p.siblings.insert(p.siblings.end(), prev.begin() + 8 * levels(i), prev.end());
        }
        restored.push_back(std::move(p));
    }
    std::vector<PathU> out;
    for (uint32_t o : b.original_order) {
        if (o >= restored.size()) throw std::runtime_error("restore: bad original_order");
        out.push_back(restored[o]);
    }
    return out;
}

// ---- blobs ----
static inline std::vector<uint32_t> prune_blob(const uint32_t* blob, const uint32_t* batch_sizes, size_t n_batches) {
    size_t k = 0;
    const uint32_t T = blob[k++];
    std::vector<uint32_t> o(blob, blob + 1 + T);
    k += T;
    const uint32_t M = blob[k++];
    size_t used = 0;
    o.push_back((uint32_t)n_batches);
    for (size_t bi = 0; bi < n_batches; bi++) {
        std::vector<PathU> paths;
        for (uint32_t q = 0; q < batch_sizes[bi]; q++) {
            if (used++ >= M) throw std::runtime_error("prune: batch sizes exceed the openings");
            PathU p;
            p.leaf_index = (uint64_t)blob[k] | ((uint64_t)blob[k + 1] << 32);
            const uint32_t ll = blob[k + 2], pl = blob[k + 3];
            k += 4;
            p.leaf.assign(blob + k, blob + k + ll);
            k += ll;
            p.siblings.assign(blob + k, blob + k + pl);
            k += pl;
            paths.push_back(std::move(p));
        }
        const PrunedBatch b = prune_batch(paths);
        o.push_back(b.height);
        o.push_back(b.n_trailing_zeros);
        o.push_back((uint32_t)b.original_order.size());
        o.insert(o.end(), b.original_order.begin(), b.original_order.end());
        o.push_back((uint32_t)b.index.size());
        for (size_t i = 0; i < b.index.size(); i++) {
            o.push_back((uint32_t)b.index[i]);
            o.push_back((uint32_t)(b.index[i] >> 32));
            o.push_back((uint32_t)b.leaf[i].size());
            o.insert(o.end(), b.leaf[i].begin(), b.leaf[i].end());
            o.push_back((uint32_t)(b.sib[i].size() / 8));
            o.insert(o.end(), b.sib[i].begin(), b.sib[i].end());
        }
    }
    if (used != M) throw std::runtime_error("prune: batch sizes do not cover the openings");
    return o;
}
static inline std::vector<PrunedBatch> parse_pruned(const uint32_t* blob, size_t n_words, size_t& transcript_words) {
    size_t k = 0;
    auto need = [&](size_t m) {
        if (k + m > n_words) throw std::runtime_error("pruned blob truncated");
    };
    need(1);
    const uint32_t T = blob[k++];
    need(T + 1);
    k += T;
    transcript_words = T;
    const uint32_t B = blob[k++];
    std::vector<PrunedBatch> out;
    for (uint32_t bi = 0; bi < B; bi++) {
        PrunedBatch b;
        need(3);
        b.height = blob[k++];
        b.n_trailing_zeros = blob[k++];
        const uint32_t no = blob[k++];
        need(no + 1);
        b.original_order.assign(blob + k, blob + k + no);
        k += no;
        const uint32_t np = blob[k++];
        for (uint32_t i = 0; i < np; i++) {
            need(3);
            b.index.push_back((uint64_t)blob[k] | ((uint64_t)blob[k + 1] << 32));
            const uint32_t ll = blob[k + 2];
            k += 3;
            need(ll + 1);
            b.leaf.emplace_back(blob + k, blob + k + ll);
            k += ll;
            const uint32_t ns = blob[k++];
            need((size_t)ns * 8);
            b.sib.emplace_back(blob + k, blob + k + (size_t)ns * 8);
            k += (size_t)ns * 8;
        }
        out.push_back(std::move(b));
    }
    if (k != n_words) throw std::runtime_error("pruned blob has trailing words");
    return out;
}
static inline std::vector<uint32_t> restore_blob(const uint32_t* blob, size_t n_words) {
    size_t T = 0;
    const std::vector<PrunedBatch> bs = parse_pruned(blob, n_words, T);
    std::vector<uint32_t> o(blob, blob + 1 + T);
    std::vector<PathU> all;
    for (const PrunedBatch& b : bs)
        for (PathU& p : restore_batch(b)) all.push_back(std::move(p));
    o.push_back((uint32_t)all.size());
    for (const PathU& p : all) {
        o.push_back((uint32_t)p.leaf_index);
        o.push_back((uint32_t)(p.leaf_index >> 32));
        o.push_back((uint32_t)p.leaf.size());
        o.push_back((uint32_t)p.siblings.size());
        o.insert(o.end(), p.leaf.begin(), p.leaf.end());
        o.insert(o.end(), p.siblings.begin(), p.siblings.end());
    }
    return o;
}
// Proof::proof_size_fe, transcript.rs:39-53
static inline uint64_t pruned_size_fe(const uint32_t* blob, size_t n_words) {
    size_t T = 0;
    uint64_t s = 0;
    for (const PrunedBatch& b : parse_pruned(blob, n_words, T)) {
        for (const auto& l : b.leaf) s += l.size();
        for (const auto& x : b.sib) s += x.size();
    }
    return s + T;
}

}  // namespace orc
