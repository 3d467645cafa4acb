"""CPU: Merkle-path pruning restatement (oracle/pruning_oracle.hpp) — the round trips of the reference's own tests
(crates/backend/fiat-shamir/src/merkle_pruning.rs:172-400: single path, adjacent leaves, duplicates, all leaves, random
subsets) on real Poseidon trees: restore(prune(paths)) == paths, the restored paths verify against the root, and
Proof::proof_size_fe shrinks as expected."""
import numpy as np
import pytest

from tests import oracle_binding as ob
from tests.oracle_binding import rand_field


def _blob(transcript, openings):
    o = [len(transcript)] + list(transcript) + [len(openings)]
    for idx, leaf, path in openings:
        o += [idx & 0xFFFFFFFF, idx >> 32, len(leaf), len(path)] + list(leaf) + list(path)
    return np.array(o, dtype=np.uint32)


def _tree(orc, rng, log_h, width, zero_tail):
    rows = rand_field(rng, (1 << log_h, width))
    if zero_tail:
        rows[:, width - zero_tail:] = 0
    layers = orc.merkle_build(rows, width)
    return rows, layers


def _open(rows, layers, log_h, idx):
    path, off, n, i = [], 0, 1 << log_h, idx
    for _ in range(log_h):
        path += list(layers[off + (i ^ 1)])
        off += n
        n >>= 1
        i >>= 1
    return idx, list(rows[idx]), path


def _lca(a, b):
    return (a ^ b).bit_length()


def _expected_size_fe(n_transcript, idxs, log_h, width, zero_tail):
    """Proof::proof_size_fe of the pruned batch, straight from the definition (merkle_pruning.rs:18-86): path i keeps the
    siblings of levels < lca(prev, i) (all levels for the first) except level lca(i, next) - 1."""
    d = sorted(set(idxs))
    fe = n_transcript + len(d) * (width - zero_tail)
    for k, i in enumerate(d):
        levels = log_h if k == 0 else _lca(d[k - 1], i)
        skip = _lca(i, d[k + 1]) - 1 if k + 1 < len(d) else None
        fe += 8 * sum(1 for lvl in range(levels) if lvl != skip)
    return fe


CASES = {
    "single": [5],
    "adjacent": [6, 7],
    "duplicates": [3, 9, 3, 3, 12, 9],
    "siblings_far": [0, 15],
    "all": list(range(16)),
    "unsorted": [11, 2, 8, 3, 10],
}


@pytest.mark.parametrize("name", list(CASES))
def test_prune_restore_round_trip(orc, name):
    rng = np.random.default_rng(len(name))
    log_h, width, zero_tail = 4, 24, 8
    rows, layers = _tree(orc, rng, log_h, width, zero_tail)
    idxs = CASES[name]
    transcript = list(rand_field(rng, 13))
    blob = _blob(transcript, [_open(rows, layers, log_h, i) for i in idxs])
    pruned = ob.prune_proof(orc, blob, [len(idxs)])
    restored = ob.restore_proof(orc, pruned)
    assert np.array_equal(restored, blob)
    # every restored path authenticates against the root
    root = layers[-1]
    for i in idxs:
        _, leaf, path = _open(rows, layers, log_h, i)
        assert orc.merkle_verify(root, log_h, i, np.array(leaf, dtype=np.uint32), np.array(path, dtype=np.uint32).reshape(log_h, 8))
    # size: transcript + distinct leaves without the common zero tail + 8 words per kept sibling
    fe = ob.pruned_size_fe(orc, pruned)
    assert fe == _expected_size_fe(13, idxs, log_h, width, zero_tail)
    if name == "single":
        assert fe == 13 + (width - zero_tail) + 8 * log_h
    if name == "adjacent":  # path 6 drops level 0 (recomputed from path 7), path 7 keeps only level 0
        assert fe == 13 + 2 * (width - zero_tail) + 8 * (log_h - 1) + 8


def test_two_batches_and_tampering(orc):
    rng = np.random.default_rng(7)
    rows_a, lay_a = _tree(orc, rng, 5, 16, 0)
    rows_b, lay_b = _tree(orc, rng, 3, 40, 3)
    ia, ib = [1, 30, 17, 1, 16], [7, 0, 3]
    blob = _blob(list(rand_field(rng, 5)), [_open(rows_a, lay_a, 5, i) for i in ia] + [_open(rows_b, lay_b, 3, i) for i in ib])
    pruned = ob.prune_proof(orc, blob, [len(ia), len(ib)])
    assert np.array_equal(ob.restore_proof(orc, pruned), blob)
    with pytest.raises(RuntimeError):
        ob.prune_proof(orc, blob, [len(ia)])  # batch sizes must cover every opening
    with pytest.raises(RuntimeError):
        ob.restore_proof(orc, pruned[:-3])  # truncated
    bad = pruned.copy()
    bad[-1] ^= 1  # a kept sibling digest: restore still succeeds, but the path no longer authenticates
    rb = ob.restore_proof(orc, bad)
    assert not np.array_equal(rb, blob)
