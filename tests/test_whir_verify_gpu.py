"""The recursion program's PCS opening (programs/whir_verify.py) on the device, on GENUINE child proofs: two leaves of real XMSS
signatures proved by this library (lmh_prove_execution_vm, rate 1/4, the reference's 124-bit parameters = `recursion --log-inv-rate 2`),
their raw transcripts / opening claims / un-pruned Merkle openings fed to the in-VM verifier, whose (child, query) loops run as
device batches (csrc/lm_vm_device.hip).  The run must equal the oracle VM's cell for cell, every parallel loop must have run on the
device, the proof of THAT execution must equal the oracle prover's word for word and be accepted by both verifiers, and a flipped
sibling must be rejected with the host runner's error.  The program is the reference's `recursion()` whole (recursion.py:48-787 + whir.py): GKR quotient,
logup statement, batched AIR sumcheck with evaluate_air_constraints in the VM, public-memory point, PCS statement, whir_open."""
import ctypes

import numpy as np
import pytest

import leanmultisig_amd as lm
from leanmultisig_amd import capi, vm
from leanmultisig_amd.programs import whir_verify as wv
from leanmultisig_amd.programs import xmss_aggregate as xa
from tests import oracle_binding as ob
from tests import synth_witness

pytestmark = pytest.mark.gpu
N_CHILDREN, CHILD_SIGS, RATE = 2, 40, 2


@pytest.fixture(scope="module")
def recursion(ctx):
    leaf = xa.build_program()
    builder = lm.WhirBuilder.default(RATE)       # default_whir_config: 124 bits, JohnsonBound
    inst = dict(log_bytecode=leaf.log_size, ending_pc=leaf.ending_pc, bytecode_hash=leaf.hash(), bytecode=leaf.multilinear)
    signer = xa.Xmss(compress=lambda x: ctx.poseidon16(x, compress=True))
    children, n_vars = [], None
    for c in range(N_CHILDREN):
        pi, wit, _ = xa.build_witness(leaf, CHILD_SIGS, np.random.default_rng(900 + c), xmss=signer)
        pr = lm.Prover(ctx)
        vm.prove_execution_vm(ctx, pr, leaf, pi, wit, builder)
        raw, claim, stmt = capi.verify_execution_raw(dict(inst, public_input=pi), pr, builder, with_statement=True)
        children.append((raw, claim, wv.parse_raw_proof(pr.proof())[1], stmt, pi))
        assert n_vars in (None, claim.num_variables)
        n_vars = claim.num_variables
    cfg = lm.WhirConfig.new(builder, n_vars).to_dict()
    # `recursion()` of the reference whole: the transcript replayed from its first word (GKR quotient, logup statement, batched AIR
    # sumcheck with the three constraint polynomials evaluated in the VM, public-memory point, PCS statement, whir_open); what is
    # taken as given: the bytecode value (a hint in the reference too) and the domain-separator digest
    bc = wv.build_program(cfg, N_CHILDREN, statement=wv.Statement(children[0][3], children[0][1], public_input_len=8), air=True, head=True, evaluators=True)
    pi, wit, _ = wv.build_witness(bc, children)
    return bc, children, pi, wit


def run_info(ex):
    info = vm.VmRunInfo()
    capi.load().lmh_execution_info(ex.h, ctypes.byref(info))
    return info.to_dict()


def test_device_run_equals_oracle_vm(ctx, orc, recursion):
    bc, children, pi, wit = recursion
    S = bc.info["shape"]
    ex = vm.execute(bc, pi, wit, n_threads=4, ctx=ctx)
    d = run_info(ex)
    assert d["vm_on_device"] and d["device_batches"] == 2 * S.n_rounds + 1 and d["host_batches"] == 0, d
    run = ob.VmRun(orc, bc, pi, wit)
    assert ex.n_cycles == run.pcs.size and ex.memory_len == run.memory.size
    assert np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)
    assert np.array_equal(ex.memory_defined(), run.defined)
    bad = np.nonzero(ex.memory() != run.memory)[0]
    assert bad.size == 0, f"memory differs at {bad[:8]}"
    assert ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows
    assert ex.n_poseidon_calls >= wv.expected_counts(S)["poseidon_calls"]  # (the opening's own calls; the statement adds none)
    ex_h = vm.execute(bc, pi, wit, n_threads=4)
    assert np.array_equal(ex.poseidon_calls(), ex_h.poseidon_calls()) and np.array_equal(ex.extension_rows(), ex_h.extension_rows())


def test_proof_of_the_verifier_run_equals_oracle_prover(ctx, orc, recursion):
    """prove_execution of the root step: VM (device batches) -> trace on the device -> proof, against the oracle's VM + prover"""
    bc, children, pi, wit = recursion
    builder = ob.whir_builder(log_inv_rate=RATE, pow_bits=6, security=60)
    lm_builder = lm.WhirBuilder.default(RATE, security_level=60, pow_bits=6)
    pr = lm.Prover(ctx)
    vm.prove_execution_vm(ctx, pr, bc, pi, wit, lm_builder)
    ww = ob.VmRun(orc, bc, pi, wit).trace(RATE)
    ok, err = lm.verify_execution(ww, pr.proof_bytes(compressed=True), lm_builder, compressed=True)
    assert ok, err
    ok, err = ob.verify_execution(orc, ww, pr.proof(), builder)
    assert ok, err
    ob.set_threads(orc, 8)
    assert np.array_equal(pr.proof(), ob.prove_execution(orc, ww, synth_witness.header(ww), builder))


def test_flipped_sibling_is_rejected_with_the_host_error(ctx, recursion):
    bc, children, pi, wit = recursion
    raw, claim, ops, stmt, pub = children[1]
    ops2 = [(i, leaf.copy(), path.copy()) for i, leaf, path in ops]
    ops2[7][2][19] ^= 1
    pi2, wit2, _ = wv.build_witness(bc, [children[0], (raw, claim, ops2, stmt, pub)])
    with pytest.raises(lm.LmError) as dev:
        vm.execute(bc, pi2, wit2, n_threads=4, ctx=ctx)
    with pytest.raises(lm.LmError) as host:
        vm.execute(bc, pi2, wit2, n_threads=4)
    assert str(dev.value) == str(host.value) and "MemoryAlreadySet" in str(host.value)


def test_recursion_n4_full_size_equals_oracle(ctx, orc):
    """BASELINE configs[3] at FULL size (`recursion --n 4 --log-inv-rate 2`, src/main.rs:91-115): four leaves of 775 REAL signatures proved
    by this library at rate 1/4 with the reference's production parameters, then the root step — `recursion()` of the in-VM verifier on
    the four genuine child proofs — with production parameters as well: the device VM run must equal the oracle VM cell for cell with
    every (child, query) loop on the device, and the root proof must equal the oracle PROVER's word for word and be accepted by both
    verifiers (bench.py --shape whir-recursion is the timed twin of this test)."""
    n_children, child_sigs, rate = 4, 775, 2
    leaf = xa.build_program(19)
    builder = lm.WhirBuilder.default(rate)
    inst = dict(log_bytecode=leaf.log_size, ending_pc=leaf.ending_pc, bytecode_hash=leaf.hash(), bytecode=leaf.multilinear)
    signer = xa.Xmss(compress=lambda x: ctx.poseidon16(x, compress=True))
    children, n_vars = [], None
    for c in range(n_children):
        pi, wit, _ = xa.build_witness(leaf, child_sigs, np.random.default_rng(7000 + c), xmss=signer)
        pr = lm.Prover(ctx)
        vm.prove_execution_vm(ctx, pr, leaf, pi, wit, builder)
        raw, claim, stmt = capi.verify_execution_raw(dict(inst, public_input=pi), pr, builder, with_statement=True)  # (the library's verifier accepts the child)
        children.append((raw, claim, wv.parse_raw_proof(pr.proof())[1], stmt, pi))
        assert n_vars in (None, claim.num_variables)
        n_vars = claim.num_variables
    assert n_vars == 25   # a 775-signature leaf at rate 1/4: stacked 2^25
    cfg = lm.WhirConfig.new(builder, n_vars).to_dict()
    bc = wv.build_program(cfg, n_children, log_size=19, statement=wv.Statement(children[0][3], children[0][1], public_input_len=8), air=True, head=True,
                          evaluators=True)
    S = bc.info["shape"]
    pi, wit, _ = wv.build_witness(bc, children)
    # the VM run: device == oracle
    ex = vm.execute(bc, pi, wit, ctx=ctx)
    d = run_info(ex)
    assert d["vm_on_device"] and d["device_batches"] == 2 * S.n_rounds + 1 and d["host_batches"] == 0, d
    run = ob.VmRun(orc, bc, pi, wit)
    assert ex.n_cycles == run.pcs.size and ex.memory_len == run.memory.size
    assert np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)
    assert np.array_equal(ex.memory_defined(), run.defined) and np.array_equal(ex.memory(), run.memory)
    assert ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows
    assert ex.n_cycles > 200_000 and ex.n_poseidon_calls > 30_000 and ex.n_extension_rows > 150_000   # (the shape profiles/r0x_bench_whir_recursion.json reports)
    # the root proof: device == oracle prover, production parameters
    pr = lm.Prover(ctx)
    vm.prove_execution_vm(ctx, pr, bc, pi, wit, builder)
    ww = run.trace(rate)
    ok, err = lm.verify_execution(ww, pr.proof_bytes(compressed=True), builder, compressed=True)
    assert ok, err
    ob_builder = ob.whir_builder(log_inv_rate=rate)
    ok, err = ob.verify_execution(orc, ww, pr.proof(), ob_builder)
    assert ok, err
    ob.set_threads(orc, 16)
    assert np.array_equal(pr.proof(), ob.prove_execution(orc, ww, synth_witness.header(ww), ob_builder))
