"""SURVEY.md §8(f) rank 4 on the device: the tables lmh_get_execution_trace builds from the runner's log equal the oracle's
get_execution_trace column for column, and lmh_prove_execution_vm (VM run + trace + proof) equals the oracle's proof of the
oracle's trace word for word — on the hand-assembled XMSS aggregation program with real signatures."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from leanmultisig_amd import vm
from leanmultisig_amd.programs import xmss_aggregate as xa
from tests import oracle_binding as ob
from tests import synth_witness

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def program():
    return xa.build_program()


def test_device_trace_equals_oracle_trace(ctx, orc, program):
    pi, w, _ = xa.build_witness(program, 7, np.random.default_rng(11), slot=0x0BADCAFE)
    ex = vm.execute(program, pi, w)
    run = ob.VmRun(orc, program, pi, w)
    ref = run.trace()
    dt = vm.DeviceTrace(ctx, program, ex, pi)
    assert dt.view.log_memory == ref["log_memory"] and dt.view.ending_pc == program.ending_pc
    assert np.array_equal(dt.memory(), ref["memory"])
    for t in range(3):
        assert dt.view.tables[t].log_rows == ref["log_rows"][t] and dt.view.tables[t].non_padded_n_rows == ref["non_padded"][t]
        got, want = dt.table(t), ref["tables"][t]
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, f"table {t}: columns {bad[:10]} differ"
    dt.close()


@pytest.mark.parametrize("n_sigs", [3, 33])
def test_prove_execution_vm_equals_oracle(ctx, orc, program, n_sigs):
    pi, w, _ = xa.build_witness(program, n_sigs, np.random.default_rng(5 + n_sigs))
    builder = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    lm_builder = lm.WhirBuilder.default(1, security_level=60, pow_bits=6)
    pr = lm.Prover(ctx)
    times = vm.prove_execution_vm(ctx, pr, program, pi, w, lm_builder)
    assert len(times) == 3
    proof = pr.proof()
    ww = ob.VmRun(orc, program, pi, w).trace()
    ok, err = lm.verify_execution(ww, pr.proof_bytes(compressed=True), lm_builder, compressed=True)
    assert ok, err
    ok, err = ob.verify_execution(orc, ww, proof, builder)
    assert ok, err
    ob.set_threads(orc, 8)
    ref = ob.prove_execution(orc, ww, synth_witness.header(ww), builder)
    assert np.array_equal(proof, ref)


@pytest.mark.parametrize("n_sigs", [5, 64])
def test_aggregate_type_1_whole_function(ctx, program, n_sigs):
    """lmh_aggregate_type_1 (type_1_aggregation.rs:206-377: sort + dedup, input hashes, hint map, prove_execution) on shuffled pairs with a
    duplicate = lmh_prove_execution_vm on the Python builder's witness, word for word; and it reports where the VM's batch ran"""
    pi, w, info = xa.build_witness(program, n_sigs, np.random.default_rng(40 + n_sigs))
    lm_builder = lm.WhirBuilder.default(1, security_level=60, pow_bits=6)
    ref = lm.Prover(ctx)
    vm.prove_execution_vm(ctx, ref, program, pi, w, lm_builder)
    raw = vm.pack_xmss_signatures(info["sig"])
    raw = np.concatenate([raw[np.random.default_rng(1).permutation(n_sigs)], raw[[2]]])
    pr = lm.Prover(ctx)
    times, run = vm.aggregate_type_1(ctx, pr, program, raw, info["message"], info["slot"], lm_builder)
    assert len(times) == 4 and times[0] > 0 and np.array_equal(pr.proof(), ref.proof())
    d = run.to_dict()
    if n_sigs > 33:
        assert d["vm_on_device"] and d["device_batches"] == 1 and d["host_batches"] == 0 and d["fallback_reason"] is None
    else:
        assert not d["vm_on_device"] and d["host_batches"] == 1 and "segments" in d["fallback_reason"]


def test_aggregate_type_1_late_inputs_error_paths(ctx, program):
    """lmh_aggregate_type_1 starts the VM while hash_pubkeys is still running (the digest inside `input_data` and the public input are
    late words of the run, lmh::VmLate).  A forged signature makes a segment fail, the batch moves to the host pool and the run is
    repeated with everything executed at once — all of which read the late words: the error must be the runner's, and the same
    context must prove the honest set afterwards (proof == the witness-first path's)."""
    n = 70
    pi, w, info = xa.build_witness(program, n, np.random.default_rng(4321))
    lm_builder = lm.WhirBuilder.default(1, security_level=60, pow_bits=6)
    raw = vm.pack_xmss_signatures(info["sig"])
    bad = raw.copy()
    bad[17, 8 + 6 + 3] ^= 1  # a chain tip of signature 17 (behind the public key and the randomness)
    pr = lm.Prover(ctx)
    with pytest.raises(lm.LmError, match="lmh_execute_bytecode"):
        vm.aggregate_type_1(ctx, pr, program, bad, info["message"], info["slot"], lm_builder)
    ref = lm.Prover(ctx)
    vm.prove_execution_vm(ctx, ref, program, pi, w, lm_builder)
    pr = lm.Prover(ctx)
    times, run = vm.aggregate_type_1(ctx, pr, program, raw[::-1].copy(), info["message"], info["slot"], lm_builder)
    assert np.array_equal(pr.proof(), ref.proof()) and run.to_dict()["vm_on_device"]


def test_runner_error_surfaces(ctx, program):
    pi, w, info = xa.build_witness(program, 3, np.random.default_rng(2))
    bad = pi.copy()
    bad[0] ^= 1     # the hash of the input buffer no longer lands on the public input
    pr = lm.Prover(ctx)
    with pytest.raises(lm.LmError, match="MemoryAlreadySet"):
        vm.prove_execution_vm(ctx, pr, program, bad, w, lm.WhirBuilder.default(1, security_level=60, pow_bits=6))


def test_whole_function_from_several_threads_at_once(ctx, program):
    """Leaves in flight: four host threads, one context each, prove different leaves with lmh_prove_execution_vm at the same time —
    every VM run on its own leased pool, uploads from registered runner buffers, the GKR tails resident side by side — and every
    proof must equal the one the same leaf gives alone."""
    import threading
    lm_builder = lm.WhirBuilder.default(1, security_level=60, pow_bits=6)
    leaves = [xa.build_witness(program, 5 + 4 * t, np.random.default_rng(700 + t))[:2] for t in range(4)]
    alone = []
    for pi, w in leaves:
        pr = lm.Prover(ctx)
        vm.prove_execution_vm(ctx, pr, program, pi, w, lm_builder, n_threads=4)
        alone.append(pr.proof().copy())
    ctxs = [lm.Context(0) for _ in range(4)]
    bad = []

    def worker(t):
        try:
            ctxs[t]._check(ctxs[t].lib.lm_bind_thread(ctxs[t].h))  # the HIP device is per host thread
            pi, w = leaves[t]
            for k in range(6):
                pr = lm.Prover(ctxs[t])
                vm.prove_execution_vm(ctxs[t], pr, program, pi, w, lm_builder, n_threads=4)
                if not np.array_equal(pr.proof(), alone[t]):
                    bad.append((t, k))
        except Exception as e:  # noqa: BLE001
            bad.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not bad, bad
