"""TEST INFRASTRUCTURE: an independent Python restatement of the reference's proof byte format, used to check the
library's serialiser (leanmultisig_amd/csrc/host/lm_wire.cpp).
  * postcard 1.1 of Proof { transcript: Vec<F>, merkle_paths: Vec<PrunedMerklePaths<F, F>> } (crates/backend/fiat-shamir/src/
    transcript.rs:33-36, merkle_pruning.rs:5-12; F = varint of the Montgomery u32, koala-bear/src/monty_31/monty_31.rs:152-157);
  * the LZ4 block format behind lz4_flex::{compress_prepend_size, decompress_size_prepended} (decoder only).
Input / output of the postcard functions is the pruned u32 blob of include/leanmultisig_host.h."""
import numpy as np

P = 0x7F000001


def _varint(v):
    out = bytearray()
    v = int(v)
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _read_varint(b, k):
    v, shift = 0, 0
    while True:
        x = b[k]
        k += 1
        v |= (x & 0x7F) << shift
        shift += 7
        if not x & 0x80:
            return v, k


def parse_pruned_blob(words):
    """pruned u32 blob -> (transcript, [dict(height, n_trailing_zeros, original_order, paths=[(index, leaf, siblings)])])"""
    w = [int(x) for x in np.asarray(words).reshape(-1)]
    k = 0
    T = w[k]; k += 1
    transcript = w[k:k + T]; k += T
    B = w[k]; k += 1
    batches = []
    for _ in range(B):
        height, tz, n_orig = w[k], w[k + 1], w[k + 2]; k += 3
        order = w[k:k + n_orig]; k += n_orig
        n_paths = w[k]; k += 1
        paths = []
        for _ in range(n_paths):
            idx = w[k] | (w[k + 1] << 32); ll = w[k + 2]; k += 3
            leaf = w[k:k + ll]; k += ll
            ns = w[k]; k += 1
            sib = w[k:k + 8 * ns]; k += 8 * ns
            paths.append((idx, leaf, sib))
        batches.append(dict(height=height, n_trailing_zeros=tz, original_order=order, paths=paths))
    assert k == len(w)
    return transcript, batches


def postcard_proof(words):
    transcript, batches = parse_pruned_blob(words)
    o = bytearray()
    o += _varint(len(transcript))
    for x in transcript:
        o += _varint(x)
    o += _varint(len(batches))
    for b in batches:
        o += _varint(b["height"])
        o += _varint(len(b["original_order"]))
        for x in b["original_order"]:
            o += _varint(x)
        o += _varint(len(b["paths"]))              # leaf_data: Vec<Vec<F>>
        for _, leaf, _ in b["paths"]:
            o += _varint(len(leaf))
            for x in leaf:
                o += _varint(x)
        o += _varint(len(b["paths"]))              # paths: Vec<(usize, Vec<[F; 8]>)>
        for idx, _, sib in b["paths"]:
            o += _varint(idx)
            o += _varint(len(sib) // 8)
            for x in sib:
                o += _varint(x)                    # [F; 8]: a tuple, no length
        o += _varint(b["n_trailing_zeros"])
    return bytes(o)


def postcard_decode(data):
    """bytes -> pruned u32 blob (inverse of postcard_proof)"""
    k = 0
    out = []
    T, k = _read_varint(data, k)
    out.append(T)
    for _ in range(T):
        v, k = _read_varint(data, k)
        assert v < P
        out.append(v)
    B, k = _read_varint(data, k)
    out.append(B)
    for _ in range(B):
        height, k = _read_varint(data, k)
        n_orig, k = _read_varint(data, k)
        order = []
        for _ in range(n_orig):
            v, k = _read_varint(data, k)
            order.append(v)
        n_leaf, k = _read_varint(data, k)
        leaves = []
        for _ in range(n_leaf):
            ll, k = _read_varint(data, k)
            leaf = []
            for _ in range(ll):
                v, k = _read_varint(data, k)
                leaf.append(v)
            leaves.append(leaf)
        n_paths, k = _read_varint(data, k)
        assert n_paths == n_leaf
        paths = []
        for _ in range(n_paths):
            idx, k = _read_varint(data, k)
            ns, k = _read_varint(data, k)
            sib = []
            for _ in range(8 * ns):
                v, k = _read_varint(data, k)
                sib.append(v)
            paths.append((idx, sib))
        tz, k = _read_varint(data, k)
        out += [height, tz, n_orig] + order + [n_paths]
        for (idx, sib), leaf in zip(paths, leaves):
            out += [idx & 0xFFFFFFFF, idx >> 32, len(leaf)] + leaf + [len(sib) // 8] + sib
    assert k == len(data)
    return np.array(out, dtype=np.uint32)


def lz4_block_decode(src):
    """LZ4 block format (token = literal length : match length - 4; lengths >= 15 continue in 255-steps; u16 LE offset)."""
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        tok = src[i]; i += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                b = src[i]; i += 1
                ll += b
                if b != 255:
                    break
        out += src[i:i + ll]; i += ll
        if i >= n:
            break
        off = src[i] | (src[i + 1] << 8); i += 2
        assert 0 < off <= len(out)
        ml = tok & 15
        if ml == 15:
            while True:
                b = src[i]; i += 1
                ml += b
                if b != 255:
                    break
        ml += 4
        for _ in range(ml):
            out.append(out[-off])
    return bytes(out)


def lz4_decompress_size_prepended(data):
    size = int.from_bytes(data[:4], "little")
    out = lz4_block_decode(data[4:])
    assert len(out) == size
    return out
