// Internal to the host layer (lm_host.cpp, lm_wire.cpp, lm_verify.cpp): the transcript objects behind the opaque handles of
// include/leanmultisig_host.h.  Not part of the ABI.
#pragma once
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../../include/leanmultisig_host.h"
#include "../kb.h"
#include "../poseidon16.h"

namespace lmh {
using kb::EF;
using kb::u32;
using kb::u64;

// ---- Challenger (crates/backend/fiat-shamir/src/challenger.rs:9-76): overwrite-mode duplex, plain permutation -----
struct Challenger {
    u32 state[16];
    bool rate_fresh = false;
    Challenger() { memset(state, 0, sizeof state); }
    void observe(const u32 v[8]) {
        memcpy(state + 8, v, 32);
        kb::poseidon16_permute(state);
        rate_fresh = true;
    }
    void observe_many(const u32* s, u64 n) {
        for (u64 off = 0; off < n; off += 8) {
            u32 buf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            memcpy(buf, s + off, (size_t)std::min<u64>(8, n - off) * 4);
            observe(buf);
        }
    }
    void duplex() {
        const u32 z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        observe(z);
    }
    bool sample(u32 out[8]) {
        if (!rate_fresh) return false;  // "stale rate. insert a duplex() before."
        memcpy(out, state + 8, 32);
        rate_fresh = false;
        return true;
    }
    bool sample_many(u64 n_blocks, std::vector<u32>& out) {
        out.clear();
        for (u64 i = 0; i < n_blocks; i++) {
            if (i) duplex();
            u32 b[8];
            if (!sample(b)) return false;
            out.insert(out.end(), b, b + 8);
        }
        return true;
    }
};


struct Opening {  // one Merkle opening as the prover produced it (MerkleOpening, fiat-shamir/src/transcript.rs:8-12)
    u64 index;
    std::vector<u32> leaf, path;
};

// PrunedMerklePaths (fiat-shamir/src/merkle_pruning.rs:5-12): the openings of one query set after MerklePaths::prune
struct PrunedPath {
    u64 leaf_index;
    std::vector<u32> leaf;      // without the common zero tail
    std::vector<u32> siblings;  // 8 words per kept sibling, bottom-up
};
struct PrunedBatch {
    u32 merkle_height = 0, n_trailing_zeros = 0;
    std::vector<u32> original_order;
    std::vector<PrunedPath> paths;
};
}  // namespace lmh

struct lmh_prover {
    lmh::Challenger ch;
    std::vector<lmh::u32> transcript;
    std::vector<lmh::Opening> openings;
    std::vector<lmh::u32> batch_sizes;  // openings per hint_merkle_paths call (one query set of one commitment), in order
};

namespace lmh {
std::vector<PrunedBatch> prune(const lmh_prover* p);  // MerklePaths::prune per batch (lm_host.cpp)
}
