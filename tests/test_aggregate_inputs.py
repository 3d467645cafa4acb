"""aggregate_type_1's input assembly in C (lmh_aggregate_type_1_witness, csrc/host/lm_aggregate.cpp; the reference:
rec_aggregation/src/type_1_aggregation.rs:206-377 up to its prove_execution call) against the Python restatement that rounds 3-4 used
(leanmultisig_amd/programs/xmss_aggregate.py::build_witness): sorted public keys, the public-input buffer and its digest, every hint
stream — word for word —, plus the reference's own behaviours: the (public key, signature) pairs arrive in ANY order and duplicates of a
public key are dropped (raw_xmss.sort_by / dedup_by, :232-233), and the witness runs through the VM."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from leanmultisig_amd import vm
from leanmultisig_amd.programs import xmss_aggregate as xa


@pytest.fixture(scope="module")
def program():
    return xa.build_program()


def signatures(program, n, seed, slot=0x00C0FFEE):
    rng = np.random.default_rng(seed)
    pi, wit, info = xa.build_witness(program, n, rng, slot=slot)
    return pi, wit, info


@pytest.mark.parametrize("n", [1, 2, 37, 70])
def test_c_hint_assembly_equals_python_builder(program, n):
    pi, wit, info = signatures(program, n, 100 + n)
    raw = vm.pack_xmss_signatures(info["sig"])                 # (build_witness returns the signatures SORTED by public key)
    perm = np.random.default_rng(n).permutation(n)
    w = vm.Type1Witness(program, raw[perm], info["message"], info["slot"])   # ... the C side gets them shuffled
    assert w.n_sigs == n
    assert np.array_equal(w.public_input, np.asarray(pi, dtype=np.uint32))
    assert np.array_equal(w.input_data, info["input_data"])
    assert np.array_equal(w.pubkeys, np.concatenate([info["sig"]["root"], info["sig"]["pp"]], axis=1))
    neb, eo, data = w.streams()
    assert int(w.c.preamble_memory_len) == wit.preamble_memory_len == xa.PREAMBLE_MEMORY_LEN
    assert np.array_equal(neb, wit.name_entry_begin) and np.array_equal(eo, wit.entry_offset)
    assert np.array_equal(data, wit.data[:data.size]) and data.size == int(wit.entry_offset[-1])


def test_duplicate_public_keys_are_dropped_and_the_witness_runs(program):
    n = 5
    pi, wit, info = signatures(program, n, 7)
    raw = vm.pack_xmss_signatures(info["sig"])
    dup = np.concatenate([raw[[3, 1]], raw[::-1], raw[[3]]])   # 8 pairs, 5 distinct keys
    w = vm.Type1Witness(program, dup, info["message"], info["slot"])
    assert w.n_sigs == n and np.array_equal(w.public_input, np.asarray(pi, dtype=np.uint32))
    lib = lm.capi.load()
    out = lm.capi.C.c_void_p()
    rc = lib.lmh_execute_bytecode(program.handle(), w.public_input.ctypes.data, 8, lm.capi.C.byref(w.c), 2, lm.capi.C.byref(out))
    assert rc == 0, lib.lm_last_error().decode()
    ex = vm.Execution(lib, out.value)
    ref = vm.execute(program, pi, wit, n_threads=2)
    assert np.array_equal(ex.pcs(), ref.pcs()) and np.array_equal(ex.memory(), ref.memory()) and ex.n_poseidon_calls == n * 166 + ref.n_poseidon_calls - n * 166
    info_c = vm.VmRunInfo()
    lib.lmh_execution_info(ex.h, lm.capi.C.byref(info_c))
    d = info_c.to_dict()
    assert not d["vm_on_device"] and d["host_batches"] == 1 and "no device context" in d["fallback_reason"]


def test_bad_arguments(program):
    pi, wit, info = signatures(program, 2, 9)
    raw = vm.pack_xmss_signatures(info["sig"])
    with pytest.raises(lm.LmError, match="at least one signature"):
        vm.Type1Witness(program, raw[:0], info["message"], info["slot"])
    bare = vm.Bytecode(program.multilinear, program.ending_pc, program.starting_frame_memory, program.hints, {})
    bare.names = {}
    # (a bytecode object without names: the C builder cannot place its streams)
    lib = lm.capi.load()
    arr = bare.hint_array()
    h = lib.lmh_bytecode_new(bare.multilinear.ctypes.data, bare.log_size, bare.size, bare.ending_pc, bare.starting_frame_memory,
                             lm.capi.C.cast(arr, lm.capi.C.c_void_p), len(bare.hints), len(program.names))
    assert h
    out = lm.capi.C.c_void_p()
    msg = np.ascontiguousarray(info["message"], dtype=np.uint32)
    assert lib.lmh_aggregate_type_1_witness(h, raw.ctypes.data, 2, msg.ctypes.data, 1, lm.capi.C.byref(out)) != 0
    assert "hint names" in lib.lm_last_error().decode()
    lib.lmh_bytecode_free(h)
