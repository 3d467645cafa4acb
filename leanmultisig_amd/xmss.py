"""The reference's XMSS / WOTS scheme (crates/xmss/src/{lib,wots,xmss}.rs) in numpy, batched over signers: produces REAL
signatures for the aggregation workload (leanmultisig_amd/programs/xmss_aggregate.py, bench.py) and verifies them.  Every hash is
Poseidon1-16 in compression mode (`poseidon16_compress(x)[..n]`); the permutation comes from the caller (`compress`: the
library's host thread pool by default, the CPU oracle in the tests).  All arrays hold Montgomery-form u32 words (the
reference's memory image); "canonical" integers are converted with `M`.

What is NOT restated: key derivation from a seed (Keccak-seeded StdRng, xmss.rs:39-66 — secret material and the off-path
Merkle nodes are simply random here, which is all a verifier ever sees) and the signers' cache."""
import numpy as np

P = 0x7F000001
V, W, CHAIN_LENGTH = 42, 3, 8          # lib.rs:21-23
NUM_CHAIN_HASHES = 110                  # lib.rs:24
TARGET_SUM = V * (CHAIN_LENGTH - 1) - NUM_CHAIN_HASHES   # 184
RANDOMNESS_LEN, MESSAGE_LEN, PP_LEN, XMSS_DIGEST_LEN = 6, 8, 4, 4
LOG_LIFETIME = 32
TWEAK_CHAIN, TWEAK_WOTS_PK, TWEAK_MERKLE, TWEAK_ENCODING = 0, 1, 2, 3


def to_monty(x):
    return (((np.asarray(x, dtype=np.uint64) % np.uint64(P)) << np.uint64(32)) % np.uint64(P)).astype(np.uint32)


def from_monty(x):
    rinv = pow(1 << 32, P - 2, P)
    return ((np.asarray(x, dtype=np.uint64) * np.uint64(rinv)) % np.uint64(P)).astype(np.uint32)


def rand_field(rng, shape):
    """uniform field elements (Montgomery words)"""
    return rng.integers(0, P, size=shape, dtype=np.uint64).astype(np.uint32)


def make_tweak(tweak_type, sub_position, index):
    """lib.rs:46-56: canonical pair [(type << 26) + (index_hi << 10) + sub_position, index_lo]"""
    assert tweak_type < 4 and np.all(np.asarray(sub_position) < (1 << 10))
    index = np.asarray(index, dtype=np.int64)
    return np.stack(np.broadcast_arrays((tweak_type << 26) + ((index >> 16) << 10) + np.asarray(sub_position, dtype=np.int64), index & 0xFFFF), axis=-1)


class Xmss:
    def __init__(self, compress=None):
        """compress: (n, 16) u32 -> (n, 16) u32, perm(x) + x; default: lmh_poseidon16_compress_many (host thread pool)"""
        if compress is None:
            from . import vm
            compress = vm.poseidon16_compress_many
        self._compress = compress
        self.M = lambda x: to_monty(x)
        self.canon = lambda x: from_monty(x).astype(np.int64)

    def compress(self, x16):
        """poseidon16_compress: (n, 16) -> (n, 16) (callers slice the digest they need)"""
        return self._compress(np.ascontiguousarray(x16, dtype=np.uint32).reshape(-1, 16))

    # ---- chain hash (wots.rs:117-133): left = [tweak(2) | 0 0 | data(4)], right = [pp(4) | 0000] -------------------------
    def chain_step(self, data, pp, slot, chain_index, step):
        n = data.shape[0]
        x = np.zeros((n, 16), dtype=np.uint32)
        x[:, 0:2] = self.M(make_tweak(TWEAK_CHAIN, np.asarray(chain_index) * CHAIN_LENGTH + np.asarray(step), slot))
        x[:, 4:8] = data
        x[:, 8:12] = pp
        return self.compress(x)[:, :4]

    # ---- wots_encode (wots.rs:151-183) on a batch of candidates -----------------------------------------------------------
    def encode(self, message, slot, pp, randomness):
        """-> (encoding (n, V) ints, valid (n,), pre_compressed (n, 8), compressed (n, 8))"""
        n = randomness.shape[0]
        x = np.zeros((n, 16), dtype=np.uint32)
        x[:, :8] = message
        x[:, 8:14] = randomness
        x[:, 14:16] = self.M(make_tweak(TWEAK_ENCODING, 0, slot))
        pre = self.compress(x)[:, :8]
        y = np.zeros((n, 16), dtype=np.uint32)
        y[:, :8] = pre
        y[:, 8:12] = pp
        comp = self.compress(y)[:, :8]
        c = self.canon(comp)
        ok = ~np.any(c == P - 1, axis=1)                       # "ensures uniformity of encoding"
        chunks = np.stack([(c[:, k] >> (W * j)) & (CHAIN_LENGTH - 1) for k in range(8) for j in range(24 // W)], axis=1)[:, :V]
        ok &= chunks.sum(axis=1) == TARGET_SUM                 # is_valid_encoding (wots.rs:185-199)
        return chunks, ok, pre, comp

    # ---- WotsPublicKey::hash (wots.rs:95-115) ---------------------------------------------------------------------------
    def wots_pk_hash(self, tips, pp, slot):
        """tips (n, V, 4) -> (leaf (n, 4), states (n, V/2 + 1, 8))"""
        n = tips.shape[0]
        x = np.zeros((n, 16), dtype=np.uint32)
        x[:, 0:2] = self.M(make_tweak(TWEAK_WOTS_PK, 0, slot))
        x[:, 4:8] = pp
        state = self.compress(x)[:, :8]
        states = [state]
        for i in range(0, V, 2):
            y = np.concatenate([state, tips[:, i], tips[:, i + 1]], axis=1)
            state = self.compress(y)[:, :8]
            states.append(state)
        return state[:, :4], np.stack(states, axis=1)

    # ---- Merkle path (xmss.rs:205-236) -----------------------------------------------------------------------------------
    def merkle_root(self, leaf, siblings, pp, slot):
        """siblings (n, 32, 4) -> (root (n, 4), nodes (n, 33, 4) with nodes[:, 0] = leaf)"""
        cur = leaf
        nodes = [cur]
        for level in range(LOG_LIFETIME):
            is_left = ((slot >> level) & 1) == 0
            parent = slot >> (level + 1)
            n = cur.shape[0]
            x = np.zeros((n, 16), dtype=np.uint32)
            x[:, 0:2] = self.M(make_tweak(TWEAK_MERKLE, level + 1, parent))
            x[:, 4:8] = pp
            x[:, 8:12] = cur if is_left else siblings[:, level]
            x[:, 12:16] = siblings[:, level] if is_left else cur
            cur = self.compress(x)[:, :4]
            nodes.append(cur)
        return cur, np.stack(nodes, axis=1)

    # ---- key generation + signing for a batch of independent signers, all at `slot`, all on `message` ------------------------
    def keygen_and_sign(self, rng, n, message, slot, rand_field):
        """-> dict(pp (n,4), root (n,4), randomness (n,6), encoding (n,V), chain_tips (n,V,4), merkle_proof (n,32,4))"""
        pp = rand_field(rng, (n, PP_LEN))
        pre_images = rand_field(rng, (n, V, 4))
        # public chain ends: CHAIN_LENGTH - 1 steps from the pre-images (WotsSecretKey::new, wots.rs:33-42)
        cur = pre_images.reshape(n * V, 4)
        ppv = np.repeat(pp, V, axis=0)
        cidx = np.tile(np.arange(V), n)
        levels = [cur]
        for step in range(CHAIN_LENGTH - 1):
            cur = self.chain_step(cur, ppv, slot, cidx, step)
            levels.append(cur)
        levels = np.stack(levels, axis=0).reshape(CHAIN_LENGTH, n, V, 4)   # levels[k] = value after k steps
        leaf, _ = self.wots_pk_hash(levels[CHAIN_LENGTH - 1], pp, slot)
        siblings = rand_field(rng, (n, LOG_LIFETIME, 4))                    # gen_random_node for every off-path node
        root, _ = self.merkle_root(leaf, siblings, pp, slot)
        # find_randomness_for_wots_encoding (wots.rs:135-149): batches of candidates per signer until one encodes
        randomness = np.zeros((n, RANDOMNESS_LEN), dtype=np.uint32)
        encoding = np.zeros((n, V), dtype=np.int64)
        todo = np.arange(n)
        while todo.size:
            per = 2048
            cand = rand_field(rng, (todo.size * per, RANDOMNESS_LEN))
            enc, ok, _, _ = self.encode(message, slot, np.repeat(pp[todo], per, axis=0), cand)
            ok = ok.reshape(todo.size, per)
            first = np.argmax(ok, axis=1)
            hit = ok[np.arange(todo.size), first]
            sel = np.arange(todo.size) * per + first
            randomness[todo[hit]] = cand[sel[hit]]
            encoding[todo[hit]] = enc[sel[hit]]
            todo = todo[~hit]
        tips = levels[encoding, np.arange(n)[:, None], np.arange(V)[None, :]]   # chain_tips[i] = pre_image after encoding[i] steps
        return dict(pp=pp, root=root, randomness=randomness, encoding=encoding, chain_tips=tips, merkle_proof=siblings)

    # ---- xmss_verify (xmss.rs:205-236) -----------------------------------------------------------------------------------
    def verify(self, sig, message, slot):
        n = sig["pp"].shape[0]
        enc, ok, _, _ = self.encode(message, slot, sig["pp"], sig["randomness"])
        cur = sig["chain_tips"].copy()
        for k in range(CHAIN_LENGTH - 1):  # chain i needs CHAIN_LENGTH - 1 - enc[i] more steps, starting at step enc[i]
            active = enc + k < CHAIN_LENGTH - 1
            s, c = np.nonzero(active)
            if s.size:
                cur[s, c] = self.chain_step(cur[s, c], sig["pp"][s], slot, c, enc[s, c] + k)
        leaf, _ = self.wots_pk_hash(cur, sig["pp"], slot)
        root, _ = self.merkle_root(leaf, sig["merkle_proof"], sig["pp"], slot)
        return ok & np.all(root == sig["root"], axis=1)
