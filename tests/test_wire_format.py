"""Proof wire format, CPU: the library's serialiser of the reference's ExecutionProof bytes —
postcard of Proof { transcript, merkle_paths: Vec<PrunedMerklePaths> } framed by lz4_flex's size-prepended block — against
an independent Python restatement of both formats (tests/wire_py.py), on the oracle's proof of the golden execution
instance and on Merkle batches with duplicates / zero tails; decoder round trips and rejection of malformed bytes."""
import hashlib
import json
import os

import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob
from tests import synth_witness, wire_py

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_instance(orc):
    """the prove_execution instance of tests/golden/vectors_r01.json (seed 14) -> (raw proof, batch sizes, builder)"""
    v = json.load(open(os.path.join(GOLD, "vectors_r01.json")))["prove_execution"]
    w = synth_witness.build(orc, np.random.default_rng(v["seed"]), n_calls=v["n_calls"])
    b = np.array(v["builder"], dtype=np.uint32)
    raw = ob.prove_execution(orc, w, synth_witness.header(w), b)
    cfg = lm.WhirConfig.new(lm.WhirBuilder.default(1, security_level=int(b[6]), pow_bits=int(b[7])), synth_witness.stacked_n_vars(w)).to_dict()
    sizes = [r["num_queries"] for r in cfg["rounds"]] + [cfg["final_queries"]]
    return w, raw, sizes, b


def test_postcard_bytes_match_the_python_restatement_and_golden(orc):
    w, raw, sizes, b = golden_instance(orc)
    pr = lm.Prover.from_raw(raw, sizes)
    pruned = pr.proof_pruned()
    assert np.array_equal(pruned, ob.prune_proof(orc, raw, sizes))
    data = pr.proof_bytes()
    assert data == wire_py.postcard_proof(pruned)
    gold = json.load(open(os.path.join(GOLD, "vectors_r02.json")))["proof_bytes"]
    assert len(data) == gold["postcard_len"] and hashlib.sha256(data).hexdigest() == gold["postcard_sha256"]
    # field elements are varints of Montgomery words: 1..5 bytes each (this tiny trace has many zero words), plus length prefixes
    fe = pr.proof_size_fe()
    assert fe == gold["proof_size_fe"] and fe < len(data) < 5 * fe + 64
    # decode: the library's decoder and the Python one give back the pruned proof; it restores to the raw proof
    dec = lm.DecodedProof(data)
    assert np.array_equal(dec.pruned_words(), pruned) and dec.size_fe() == fe
    assert np.array_equal(wire_py.postcard_decode(data), pruned)
    assert np.array_equal(ob.restore_proof(orc, dec.pruned_words()), raw)


def test_lz4_frame(orc):
    _, raw, sizes, _ = golden_instance(orc)
    pr = lm.Prover.from_raw(raw, sizes)
    data, comp = pr.proof_bytes(), pr.proof_bytes(compressed=True)
    assert int.from_bytes(comp[:4], "little") == len(data)
    assert wire_py.lz4_decompress_size_prepended(comp) == data          # an independent decoder accepts the block
    assert lm.lz4_decompress(comp) == data
    assert np.array_equal(lm.DecodedProof(comp, compressed=True).pruned_words(), pr.proof_pruned())
    assert len(comp) <= len(data) + len(data) // 255 + 24                # random field words: essentially incompressible
    rng = np.random.default_rng(5)
    for blob in (b"", b"a", bytes(11), bytes(12), bytes(13), bytes(70000), b"abcd" * 9000, rng.bytes(3000) + bytes(500) + rng.bytes(17),
                 (rng.bytes(64) * 40)[:2501]):
        c = lm.lz4_compress(blob)
        assert wire_py.lz4_decompress_size_prepended(c) == blob and lm.lz4_decompress(c) == blob
    assert len(lm.lz4_compress(bytes(70000))) < 400 and len(lm.lz4_compress(b"abcd" * 9000)) < 300
    # hand-assembled block: 1 literal + an overlapping match of 19 (offset 1) + 5 trailing literals
    block = bytes([0x1F, ord("x"), 1, 0, 0]) + bytes([0x50]) + b"tail!"
    framed = (25).to_bytes(4, "little") + block
    assert lm.lz4_decompress(framed) == b"x" * 20 + b"tail!" == wire_py.lz4_decompress_size_prepended(framed)
    # malformed: wrong size prefix, offset before the start, truncated literals
    assert lm.lz4_decompress((24).to_bytes(4, "little") + block) is None
    assert lm.lz4_decompress((25).to_bytes(4, "little") + bytes([0x1F, ord("x"), 2, 0, 0, 0x50]) + b"tail!") is None
    assert lm.lz4_decompress((25).to_bytes(4, "little") + block[:-2]) is None


def test_malformed_postcard_is_rejected(orc):
    _, raw, sizes, _ = golden_instance(orc)
    data = lm.Prover.from_raw(raw, sizes).proof_bytes()
    lm.DecodedProof(data)
    for bad in (data[:-1], data + b"\x00", data[:100], b"", b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01"):
        with pytest.raises(lm.LmError):
            lm.DecodedProof(bad)
    # a non-canonical field word (>= p) in the transcript: MontyField31::deserialize rejects it (monty_31.rs:159-168)
    k = 0
    _, k = wire_py._read_varint(data, k)          # transcript length
    v, k2 = wire_py._read_varint(data, k)         # first word
    bad = data[:k] + wire_py._varint(wire_py.P + 5) + data[k2:]
    with pytest.raises(lm.LmError):
        lm.DecodedProof(bad)
    # an over-long varint for a u32 (6 bytes)
    bad = data[:k] + b"\x80\x80\x80\x80\x80\x01" + data[k2:]
    with pytest.raises(lm.LmError):
        lm.DecodedProof(bad)


def test_batches_with_duplicates_and_zero_tails(orc):
    """the pruning shapes of the reference's own tests (merkle_pruning.rs:172-400) through the byte format"""
    from tests.test_oracle_pruning import _blob, _open, _tree
    rng = np.random.default_rng(9)
    rows_a, lay_a = _tree(orc, rng, 5, 16, 0)
    rows_b, lay_b = _tree(orc, rng, 3, 40, 3)
    ia, ib = [1, 30, 17, 1, 16], [7, 0, 3, 3]
    blob = _blob(list(ob.rand_field(rng, 5)), [_open(rows_a, lay_a, 5, i) for i in ia] + [_open(rows_b, lay_b, 3, i) for i in ib])
    pr = lm.Prover.from_raw(blob, [len(ia), len(ib)])
    pruned = pr.proof_pruned()
    assert np.array_equal(pruned, ob.prune_proof(orc, blob, [len(ia), len(ib)]))
    data = pr.proof_bytes()
    assert data == wire_py.postcard_proof(pruned)
    assert np.array_equal(lm.DecodedProof(data).pruned_words(), pruned)
    with pytest.raises(lm.LmError):
        lm.Prover.from_raw(blob, [len(ia)])  # batch sizes must cover every opening


def test_external_pin_fixture_is_current(orc, tmp_path):
    """tests/golden/external_pin/{proof,instance}.bin — what rust_shim's #[test] feeds to the REFERENCE's verify_execution
    (default_whir_config, 124-bit): the committed files are what tools/write_proof.py produces, and the library's own
    verifier accepts them with the builder read off the proof."""
    import subprocess
    import sys
    pin = os.path.join(GOLD, "external_pin")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "write_proof.py"), str(tmp_path), "--cpu"], cwd=root)
    for f in ("proof.bin", "instance.bin"):
        assert open(os.path.join(pin, f), "rb").read() == open(tmp_path / f, "rb").read(), f
    inst = np.frombuffer(open(os.path.join(pin, "instance.bin"), "rb").read(), dtype="<u4")
    log_bc, ending_pc, n_pub = int(inst[0]), int(inst[1]), int(inst[2])
    w = dict(log_bytecode=log_bc, ending_pc=ending_pc, bytecode_hash=inst[3:11], public_input=inst[11:11 + n_pub],
             bytecode=inst[11 + n_pub:].reshape(-1, 16))
    ok, err = lm.verify_execution(w, open(os.path.join(pin, "proof.bin"), "rb").read())
    assert ok, err


def test_decoder_rejects_overlong_varint_and_dishonest_size_prefix():
    """a 10-byte varint whose last byte carries bits beyond the 64th is malformed (postcard rejects it, it is not
    truncated); an LZ4 size prefix larger than any block of that length can expand to is refused before anything is allocated."""
    import leanmultisig_amd as lm
    bad = bytes([0xFF] * 9 + [0x02])          # 2^64: does not fit a u64
    with pytest.raises(lm.LmError):
        lm.DecodedProof(bad)
    ok_max = bytes([0xFF] * 9 + [0x01])       # 2^64 - 1 is a well-formed varint (then fails as a LENGTH the input cannot hold)
    with pytest.raises(lm.LmError, match="length"):
        lm.DecodedProof(ok_max)
    frame = (1 << 29).to_bytes(4, "little") + bytes(16)
    with pytest.raises(lm.LmError, match="size prefix"):
        lm.DecodedProof(frame, compressed=True)
