// Proof wire format of the reference:
//   ExecutionProof { proof: Proof<F>, #[serde(skip)] metadata }          crates/lean_prover/src/prove_execution.rs:12-18
//   Proof<F> { transcript: Vec<F>, merkle_paths: Vec<PrunedMerklePaths<F, F>> }     fiat-shamir/src/transcript.rs:33-36
//   PrunedMerklePaths { merkle_height: usize, original_order: Vec<usize>, leaf_data: Vec<Vec<F>>,
//                       paths: Vec<(usize, Vec<[F; 8]>)>, n_trailing_zeros: usize }  fiat-shamir/src/merkle_pruning.rs:5-12
// serialised with serde + postcard 1.1 (Cargo.toml:80) and framed with lz4_flex::compress_prepend_size
// (rec_aggregation/src/type_1_aggregation.rs:81-89).
// postcard's wire format: structs and tuples are their fields in order; fixed arrays [T; N] are tuples (no length); a
// sequence is varint(len) + elements; every integer wider than 8 bits — u32, usize — is an unsigned LEB128 varint.  A field
// element is `serialize_u32(self.value)`, its Montgomery word (monty_31.rs:152-157), hence a 1..5-byte varint; the
// deserialiser rejects words >= p (:159-168).
// LZ4: compress_prepend_size = u32 little-endian uncompressed length + one LZ4 block.  Compressed bytes are not canonical
// (any valid block decodes to the same bytes; the reference's decompress accepts any), so "bit-exact proof bytes" is a
// statement about the postcard stream.  The encoder below is a plain greedy hash-table LZ4 block compressor.
#include <string.h>
#include <vector>
#include "lm_host_internal.h"

void lm_set_error(const char* fmt, ...);

namespace {
using lmh::u32;
using lmh::u64;
typedef std::vector<uint8_t> Bytes;

void put_varint(Bytes& o, u64 v) {
    while (v >= 0x80) {
        o.push_back((uint8_t)(v | 0x80));
        v >>= 7;
    }
    o.push_back((uint8_t)v);
}
void put_fe(Bytes& o, const u32* w, size_t n) {
    for (size_t i = 0; i < n; i++) put_varint(o, w[i]);
}

Bytes postcard_proof(const lmh_prover* p) {
    Bytes o;
    o.reserve(p->transcript.size() * 5 + 1024);
    put_varint(o, p->transcript.size());
    put_fe(o, p->transcript.data(), p->transcript.size());
    const std::vector<lmh::PrunedBatch> batches = lmh::prune(p);
    put_varint(o, batches.size());
    for (const lmh::PrunedBatch& b : batches) {
        put_varint(o, b.merkle_height);
        put_varint(o, b.original_order.size());
        for (u32 x : b.original_order) put_varint(o, x);
        put_varint(o, b.paths.size());  // leaf_data: Vec<Vec<F>>
        for (const lmh::PrunedPath& pp : b.paths) {
            put_varint(o, pp.leaf.size());
            put_fe(o, pp.leaf.data(), pp.leaf.size());
        }
        put_varint(o, b.paths.size());  // paths: Vec<(usize, Vec<[F; 8]>)>
        for (const lmh::PrunedPath& pp : b.paths) {
            put_varint(o, pp.leaf_index);
            put_varint(o, pp.siblings.size() / 8);
            put_fe(o, pp.siblings.data(), pp.siblings.size());
        }
        put_varint(o, b.n_trailing_zeros);
    }
    return o;
}

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    u64 varint(unsigned max_bits) {  // postcard rejects encodings longer than ceil(bits / 7) bytes
        u64 v = 0;
        const unsigned max_bytes = (max_bits + 6) / 7;
        for (unsigned i = 0; i < max_bytes; i++) {
            if (p >= end) return ok = false, 0;
            const uint8_t b = *p++;
            if (7 * i + 7 > 64 && ((b & 0x7f) >> (64 - 7 * i))) return ok = false, 0;  // bits beyond the 64th (postcard: "bad varint")
            v |= (u64)(b & 0x7f) << (7 * i);
            if (!(b & 0x80)) {
                if (max_bits < 64 && (v >> max_bits)) ok = false;
                return v;
            }
        }
        return ok = false, 0;
    }
    u32 fe() {
        const u64 v = varint(32);
        if (v >= kb::P) ok = false;  // "non-canonical MontyField31 value"
        return (u32)v;
    }
    // a length that the remaining input can actually hold (each element takes at least min_bytes)
    u64 len(u64 min_bytes) {
        const u64 n = varint(64);
        if (!ok || n > (u64)(end - p) / (min_bytes ? min_bytes : 1)) return ok = false, 0;
        return n;
    }
};

// ---- LZ4 block format ----------------------------------------------------------------------------------------------------
// sequence = token (literal length : 4 | match length - 4 : 4), [extra literal length bytes], literals, offset (u16 LE),
// [extra match length bytes]; the last sequence has literals only; the last 5 bytes are literals and the last match starts
// at least 12 bytes before the end of the block.
void lz4_emit(Bytes& o, const uint8_t* lit, size_t n_lit, size_t match_len, size_t offset) {
    const size_t ml = match_len ? match_len - 4 : 0;
    o.push_back((uint8_t)((n_lit >= 15 ? 15 : n_lit) << 4 | (match_len ? (ml >= 15 ? 15 : ml) : 0)));
    if (n_lit >= 15) {
        size_t r = n_lit - 15;
        for (; r >= 255; r -= 255) o.push_back(255);
        o.push_back((uint8_t)r);
    }
    o.insert(o.end(), lit, lit + n_lit);
    if (!match_len) return;
    o.push_back((uint8_t)offset);
    o.push_back((uint8_t)(offset >> 8));
    if (ml >= 15) {
        size_t r = ml - 15;
        for (; r >= 255; r -= 255) o.push_back(255);
        o.push_back((uint8_t)r);
    }
}
Bytes lz4_block(const uint8_t* in, size_t n) {
    Bytes o;
    o.reserve(n + n / 255 + 16);
    if (n < 13) {
        lz4_emit(o, in, n, 0, 0);
        return o;
    }
    std::vector<int64_t> table(1 << 16, -1);
    auto rd32 = [&](size_t i) {
        u32 v;
        memcpy(&v, in + i, 4);
        return v;
    };
    const size_t match_limit = n - 12, last_literals = 5;
    size_t anchor = 0, i = 0;
    while (i < match_limit) {
        const u32 h = (rd32(i) * 2654435761u) >> 16;
        const int64_t cand = table[h];
        table[h] = (int64_t)i;
        if (cand >= 0 && i - (size_t)cand <= 0xffff && rd32((size_t)cand) == rd32(i)) {
            size_t len = 4;
            while (i + len < n - last_literals && in[(size_t)cand + len] == in[i + len]) len++;
            lz4_emit(o, in + anchor, i - anchor, len, i - (size_t)cand);
            i += len;
            anchor = i;
        } else {
            i++;
        }
    }
    lz4_emit(o, in + anchor, n - anchor, 0, 0);
    return o;
}
// returns the number of bytes written or -1 on malformed input / overflow of `cap`
int64_t lz4_block_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    size_t ip = 0, op = 0;
    while (ip < n) {
        const uint8_t tok = in[ip++];
        size_t ll = tok >> 4;
        if (ll == 15) {
            uint8_t b;
            do {
                if (ip >= n) return -1;
                b = in[ip++];
                ll += b;
            } while (b == 255);
        }
        if (ll > n - ip || ll > cap - op) return -1;
        memcpy(out + op, in + ip, ll);
        ip += ll;
        op += ll;
        if (ip == n) break;  // last sequence: literals only
        if (n - ip < 2) return -1;
        const size_t off = in[ip] | (size_t)in[ip + 1] << 8;
        ip += 2;
        if (off == 0 || off > op) return -1;
        size_t ml = (tok & 15);
        if (ml == 15) {
            uint8_t b;
            do {
                if (ip >= n) return -1;
                b = in[ip++];
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if (ml > cap - op) return -1;
        for (size_t k = 0; k < ml; k++, op++) out[op] = out[op - off];  // overlapping copies are the run-length case
    }
    return (int64_t)op;
}

}  // namespace

struct lmh_proof {  // a decoded Proof<F>
    std::vector<u32> transcript;
    std::vector<lmh::PrunedBatch> batches;
};

namespace lmh {
const std::vector<u32>& proof_transcript(const lmh_proof* p) { return p->transcript; }
const std::vector<PrunedBatch>& proof_batches(const lmh_proof* p) { return p->batches; }
}  // namespace lmh

extern "C" {

uint64_t lmh_proof_postcard_size(const lmh_prover* p) { return p ? postcard_proof(p).size() : 0; }
void lmh_proof_postcard(const lmh_prover* p, uint8_t* out) {
    const Bytes b = postcard_proof(p);
    memcpy(out, b.data(), b.size());
}

uint64_t lmh_lz4_compress_bound(uint64_t n) { return 4 + n + n / 255 + 16; }
uint64_t lmh_lz4_compress_prepend_size(const uint8_t* in, uint64_t n, uint8_t* out) {
    const Bytes b = lz4_block(in, (size_t)n);
    const u32 n32 = (u32)n;
    memcpy(out, &n32, 4);  // little-endian host
    memcpy(out + 4, b.data(), b.size());
    return 4 + b.size();
}
int64_t lmh_lz4_decompress_size_prepended(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap) {
    if (!in || n < 4) return -1;
    u32 size;
    memcpy(&size, in, 4);
    if (!out) return size;  // size query
    if (size > cap) return -1;
    const int64_t got = lz4_block_decode(in + 4, (size_t)n - 4, out, size);
    return got == (int64_t)size ? got : -1;
}
uint64_t lmh_proof_compressed_size(const lmh_prover* p) {
    if (!p) return 0;
    const Bytes b = postcard_proof(p);
    return 4 + lz4_block(b.data(), b.size()).size();
}
void lmh_proof_compressed(const lmh_prover* p, uint8_t* out) {
    const Bytes b = postcard_proof(p);
    lmh_lz4_compress_prepend_size(b.data(), b.size(), out);
}

// postcard::from_bytes::<Proof<F>>: NULL (with lm_last_error) on truncated input, over-long varints, lengths the input
// cannot hold, non-canonical field words or trailing bytes.
static lmh_proof* proof_from_postcard(const uint8_t* bytes, uint64_t n);
lmh_proof* lmh_proof_from_postcard(const uint8_t* bytes, uint64_t n) {
    try {  // nothing unwinds across the ABI
        return proof_from_postcard(bytes, n);
    } catch (...) {
        lm_set_error("lmh_proof_from_postcard: out of memory");
        return nullptr;
    }
}
static lmh_proof* proof_from_postcard(const uint8_t* bytes, uint64_t n) {
    if (!bytes) return nullptr;
    Reader r{bytes, bytes + n};
    lmh_proof* pf = new lmh_proof();
    auto fail = [&](const char* what) {
        lm_set_error("lmh_proof_from_postcard: %s at byte %llu", what, (unsigned long long)(r.p - bytes));
        delete pf;
        return (lmh_proof*)nullptr;
    };
    const u64 nt = r.len(1);
    if (!r.ok) return fail("transcript length");
    pf->transcript.resize(nt);
    for (u64 i = 0; i < nt; i++) pf->transcript[i] = r.fe();
    if (!r.ok) return fail("transcript word");
    const u64 nb = r.len(5);
    if (!r.ok) return fail("merkle_paths length");
    pf->batches.resize(nb);
    for (lmh::PrunedBatch& b : pf->batches) {
        const u64 h = r.varint(64);
        if (!r.ok || h > 64) return fail("merkle_height");
        b.merkle_height = (u32)h;
        const u64 no = r.len(1);
        if (!r.ok) return fail("original_order length");
        b.original_order.resize(no);
        for (u64 i = 0; i < no; i++) {
            const u64 v = r.varint(64);
            if (!r.ok || v > 0xffffffffull) return fail("original_order entry");
            b.original_order[i] = (u32)v;
        }
        const u64 nl = r.len(1);
        if (!r.ok) return fail("leaf_data length");
        b.paths.resize(nl);
        for (lmh::PrunedPath& pp : b.paths) {
            const u64 ll = r.len(1);
            if (!r.ok) return fail("leaf length");
            pp.leaf.resize(ll);
            for (u64 i = 0; i < ll; i++) pp.leaf[i] = r.fe();
            if (!r.ok) return fail("leaf word");
        }
        const u64 np = r.len(2);
        if (!r.ok || np != nl) return fail("paths length");  // prune() emits one path per leaf; restore() indexes both alike
        for (lmh::PrunedPath& pp : b.paths) {
            pp.leaf_index = r.varint(64);
            const u64 ns = r.len(8);
            if (!r.ok) return fail("sibling count");
            pp.siblings.resize(ns * 8);
            for (u64 i = 0; i < ns * 8; i++) pp.siblings[i] = r.fe();
            if (!r.ok) return fail("sibling word");
        }
        const u64 tz = r.varint(64);
        if (!r.ok || tz > 0xffffffffull) return fail("n_trailing_zeros");
        b.n_trailing_zeros = (u32)tz;
    }
    if (r.p != r.end) return fail("trailing bytes");
    return pf;
}
lmh_proof* lmh_proof_decompress(const uint8_t* bytes, uint64_t n) {
    const int64_t size = lmh_lz4_decompress_size_prepended(bytes, n, nullptr, 0);
    // an LZ4 block expands by at most 255x (one length byte per 255 output bytes): a larger size prefix cannot be honest, and
    // nothing is allocated for it
    if (size < 0 || size > (1ll << 30) || (u64)size > 255 * n + 64) {
        lm_set_error("lmh_proof_decompress: bad size prefix");
        return nullptr;
    }
    try {
        Bytes raw((size_t)size);
        if (lmh_lz4_decompress_size_prepended(bytes, n, raw.data(), raw.size()) != size) {
            lm_set_error("lmh_proof_decompress: malformed LZ4 block");
            return nullptr;
        }
        return lmh_proof_from_postcard(raw.data(), raw.size());
    } catch (...) {
        lm_set_error("lmh_proof_decompress: out of memory");
        return nullptr;
    }
}
void lmh_proof_free(lmh_proof* p) { delete p; }
// Proof::proof_size_fe (transcript.rs:39-53)
uint64_t lmh_proof_decoded_size_fe(const lmh_proof* p) {
    if (!p) return 0;
    u64 fe = p->transcript.size();
    for (const lmh::PrunedBatch& b : p->batches)
        for (const lmh::PrunedPath& pp : b.paths) fe += pp.leaf.size() + pp.siblings.size();
    return fe;
}
// the decoded proof in the u32-word layout of lmh_proof_pruned_copy (tests compare the two)
uint64_t lmh_proof_decoded_pruned_words(const lmh_proof* p, uint32_t* out) {
    std::vector<u32> o;
    o.push_back((u32)p->transcript.size());
    o.insert(o.end(), p->transcript.begin(), p->transcript.end());
    o.push_back((u32)p->batches.size());
    for (const lmh::PrunedBatch& pb : p->batches) {
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
    if (out) memcpy(out, o.data(), o.size() * 4);
    return o.size();
}

}  // extern "C"
