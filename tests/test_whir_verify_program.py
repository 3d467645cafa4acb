"""The recursion program's PCS opening (leanmultisig_amd/programs/whir_verify.py = zkdsl_implem/whir.py `whir_open`, hand-assembled)
on the host runner, CPU: it accepts genuine proofs (here: the ORACLE prover's, which the device prover's equal word for word), its
run equals the oracle VM's (oracle/vm_oracle.hpp) cell for cell, its Poseidon / ExtensionOp / hint consumption equals what the
protocol parameters predict, and it rejects what lmh_verify_execution rejects: a flipped sibling, a changed leaf, a changed
transcript word, a wrong claim (root, sums, folding randomness), a wrong public input."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from leanmultisig_amd import capi, vm
from leanmultisig_amd.programs import whir_verify as wv
from tests import oracle_binding as ob
from tests import synth_witness

N_CHILDREN = 2


def _children(orc):
    ob_b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    lm_b = lm.WhirBuilder.default(1, security_level=60, pow_bits=6)
    children, cfg = [], None
    for c in range(N_CHILDREN):
        w = synth_witness.build(orc, np.random.default_rng(131 + c), n_calls=40)
        cfg = lm.WhirConfig.new(lm_b, synth_witness.stacked_n_vars(w)).to_dict()
        sizes = [r["num_queries"] for r in cfg["rounds"]] + [cfg["final_queries"]]
        pr = lm.Prover.from_raw(ob.prove_execution(orc, w, synth_witness.header(w), ob_b), sizes)
        raw, claim, stmt = capi.verify_execution_raw(w, pr, lm_b, with_statement=True)
        children.append((raw, claim, wv.parse_raw_proof(pr.proof())[1], stmt, w["public_input"]))
    return cfg, children


@pytest.fixture(scope="module")
def setup(orc):
    """the opening alone: the statement's two sums are claims"""
    cfg, children = _children(orc)
    return wv.build_program(cfg, N_CHILDREN), [ch[:3] for ch in children]


@pytest.fixture(scope="module")
def setup_stmt(orc):
    """the opening WITH its statement (recursion.py:469-518, 534-652): points in the claims, values read from the raw transcript"""
    cfg, children = _children(orc)
    T = wv.Statement(children[0][3], children[0][1])
    assert T.n_values == children[0][1].n_statement_values == 252
    return wv.build_program(cfg, N_CHILDREN, statement=T), children


@pytest.fixture(scope="module")
def setup_air(orc):
    """from the batched AIR sumcheck on (recursion.py:383-654): the sumcheck's challenges and the public-memory point are sampled in the VM"""
    cfg, children = _children(orc)
    return wv.build_program(cfg, N_CHILDREN, statement=wv.Statement(children[0][3], children[0][1]), air=True), children


@pytest.fixture(scope="module")
def setup_head(orc):
    """the whole verifier of recursion.py except evaluate_air_constraints: the transcript is replayed from its first word"""
    cfg, children = _children(orc)
    T = wv.Statement(children[0][3], children[0][1], public_input_len=len(children[0][4]))
    return wv.build_program(cfg, N_CHILDREN, statement=T, air=True, head=True), children


@pytest.fixture(scope="module")
def setup_full(orc):
    """the reference's `recursion()` whole: as setup_head, and the three AIR constraint polynomials evaluated in the VM (programs/air_eval.py)"""
    cfg, children = _children(orc)
    T = wv.Statement(children[0][3], children[0][1], public_input_len=len(children[0][4]))
    return wv.build_program(cfg, N_CHILDREN, statement=T, air=True, head=True, evaluators=True), children


def test_raw_transcript_layout(setup):
    """RawProof::transcript (fiat-shamir/src/verifier.rs:54-60,126-195): whole rate blocks, the part whir_open reads has the length the
    configuration implies, and the claim's sponge state is reproducible from nothing but the raw transcript's words"""
    bc, children = setup
    S = bc.info["shape"]
    for raw, claim, openings in children:
        assert raw.size % 8 == 0 and raw.size - int(claim.transcript_offset) == S.transcript_words
        assert len(openings) == sum(S.queries) and claim.n_ood == S.oods[0] and claim.num_variables == S.n


def test_accepts_and_equals_oracle_vm(orc, setup):
    bc, children = setup
    S = bc.info["shape"]
    pi, wit, _ = wv.build_witness(bc, children)
    ex = vm.execute(bc, pi, wit, n_threads=4)
    run = ob.VmRun(orc, bc, pi, wit)
    assert ex.n_cycles == run.pcs.size and ex.memory_len == run.memory.size
    assert np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)
    assert np.array_equal(ex.memory_defined(), run.defined) and np.array_equal(ex.memory(), run.memory)
    assert ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows
    # what the protocol parameters predict (tools/recursion_shape.py counts the same terms for the reference's configuration)
    assert ex.n_poseidon_calls == wv.expected_counts(S)["poseidon_calls"]
    assert ex.n_extension_rows == wv.expected_counts(S)["extension_rows"]
    info = vm.VmRunInfo()
    import ctypes
    capi.load().lmh_execution_info(ex.h, ctypes.byref(info))
    # every (child, query) loop is a parallel batch of its own: n_rounds + 1 Merkle loops and n_rounds eq-factor loops
    assert info.n_host_batches == 2 * S.n_rounds + 1


def _rejected(bc, children, match=None):
    pi, wit, _ = wv.build_witness(bc, children)
    with pytest.raises(lm.LmError, match=match):
        vm.execute(bc, pi, wit, n_threads=4)


def test_rejects_tampered_openings(setup):
    bc, children = setup
    raw, claim, ops = children[1]
    n0 = bc.info["shape"].queries[0]
    for k, part, word in ((3, 2, 9), (n0 + 2, 2, 37), (5, 1, 17), (n0 + 1, 1, 150)):  # (opening, 1 = leaf / 2 = path, word)
        ops2 = [(i, leaf.copy(), path.copy()) for i, leaf, path in ops]
        ops2[k][part][word] = (int(ops2[k][part][word]) + 1) % wv.P
        _rejected(bc, [children[0], (raw, claim, ops2)], "MemoryAlreadySet|InvalidExtensionOp")


def test_rejects_tampered_transcript_and_claim(setup):
    bc, children = setup
    raw, claim, ops = children[0]
    off = int(claim.transcript_offset)
    for pos in (off + 5, off + 16 * 7 + 3, raw.size - 3):
        raw2 = raw.copy()
        raw2[pos] ^= 2
        _rejected(bc, [(raw2, claim, ops), children[1]])
    for field, k in (("statement_weights", 2), ("statement_sum", 0), ("folding_randomness", 41), ("root", 1), ("ood_points", 3), ("ood_answers", 4),
                     ("challenger_state", 2), ("challenger_state", 7)):
        cl2 = capi.WhirOpeningClaim.from_buffer_copy(claim)
        getattr(cl2, field)[k] ^= 1
        _rejected(bc, [(raw, cl2, ops), children[1]])
    pi, wit, _ = wv.build_witness(bc, children)
    pi = pi.copy()
    pi[3] ^= 1
    with pytest.raises(lm.LmError, match="MemoryAlreadySet"):
        vm.execute(bc, pi, wit)


def test_with_statement_accepts_and_equals_oracle_vm(orc, setup_stmt):
    bc, children = setup_stmt
    pi, wit, _ = wv.build_witness(bc, children)
    ex = vm.execute(bc, pi, wit, n_threads=4)
    run = ob.VmRun(orc, bc, pi, wit)
    assert ex.n_cycles == run.pcs.size and np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)
    assert np.array_equal(ex.memory_defined(), run.defined) and np.array_equal(ex.memory(), run.memory)
    assert ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows


def test_with_statement_rejects_a_wrong_statement(setup_stmt):
    """every ingredient of the PCS statement is bound: a claimed evaluation in the part of the transcript in front of the opening (a
    logup column value, a column evaluation behind the AIR sumcheck, value_memory), each of the three points, the child's public input"""
    bc, children = setup_stmt
    raw, claim, ops, stmt, pub = children[0]
    T = bc.info["shape"].statement
    for off in (T.off_value_memory + 1, T.off_value_bytecode_acc, T.logup[0][3][1] + 2, T.logup[2][-1][1], T.off_inner[2] + 5 * 50 + 1, T.off_inner[1] + 5 * 30,
                T.off_inner[0] + 3):
        raw2 = raw.copy()
        raw2[off] ^= 1
        _rejected(bc, [(raw2, claim, ops, stmt, pub), children[1]], "InvalidExtensionOp")
    for field, k in (("gkr_point", 7), ("gkr_point", 5 * (T.gkr_n_vars - 1)), ("air_point", 2), ("air_point", 5 * (T.n_max - 1) + 4), ("pm_point", 6)):
        st2 = capi.PcsStatementClaim.from_buffer_copy(stmt)
        getattr(st2, field)[k] ^= 1
        _rejected(bc, [(raw, claim, ops, st2, pub), children[1]], "InvalidExtensionOp")
    pub2 = np.asarray(pub).copy()
    pub2[5] ^= 1
    _rejected(bc, [(raw, claim, ops, stmt, pub2), children[1]], "InvalidExtensionOp")
    # a child of another shape than the program was assembled for is refused when the witness is built
    st2 = capi.PcsStatementClaim.from_buffer_copy(stmt)
    st2.log_rows[1] += 1
    with pytest.raises(AssertionError, match="another shape"):
        wv.build_witness(bc, [(raw, claim, ops, st2, pub), children[1]])


def test_from_the_air_sumcheck_accepts_and_equals_oracle_vm(orc, setup_air):
    bc, children = setup_air
    pi, wit, _ = wv.build_witness(bc, children)
    ex = vm.execute(bc, pi, wit, n_threads=4)
    run = ob.VmRun(orc, bc, pi, wit)
    assert ex.n_cycles == run.pcs.size and np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)
    assert np.array_equal(ex.memory_defined(), run.defined) and np.array_equal(ex.memory(), run.memory)
    assert ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows


def test_from_the_air_sumcheck_rejects(setup_air):
    """a round polynomial of the AIR sumcheck, a column evaluation behind it (absorbed: the transcript diverges), a bus evaluation, and
    every claim that is left (logup_c, the three constraint evaluations, the GKR point, the sponge state)"""
    bc, children = setup_air
    raw, claim, ops, stmt, pub = children[1]
    T = bc.info["shape"].statement
    for off in (T.air_off + 2, T.air_off + 56 * 3 + 11, T.off_inner[0] + 7, T.off_inner[2] + 5 * 100, T.off_bus_selector[1] + 1, T.off_bus_data[2] + 4,
                T.off_value_memory):
        raw2 = raw.copy()
        raw2[off] ^= 1
        _rejected(bc, [children[0], (raw2, claim, ops, stmt, pub)], "InvalidExtensionOp|NotEqual")
    for field, k in (("logup_c", 1), ("gkr_point", 11), ("air_challenger_state", 3), ("air_challenger_state", 9)):
        st2 = capi.PcsStatementClaim.from_buffer_copy(stmt)
        getattr(st2, field)[k] ^= 1
        _rejected(bc, [children[0], (raw, claim, ops, st2, pub)], "InvalidExtensionOp|NotEqual")
    for t in range(3):
        st2 = capi.PcsStatementClaim.from_buffer_copy(stmt)
        st2.air_constraint_evals[t][2] ^= 1
        _rejected(bc, [children[0], (raw, claim, ops, st2, pub)], "InvalidExtensionOp|NotEqual")


def test_whole_verifier_accepts_and_equals_oracle_vm(orc, setup_head):
    bc, children = setup_head
    pi, wit, _ = wv.build_witness(bc, children)
    ex = vm.execute(bc, pi, wit, n_threads=4)
    run = ob.VmRun(orc, bc, pi, wit)
    assert ex.n_cycles == run.pcs.size and np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)
    assert np.array_equal(ex.memory_defined(), run.defined) and np.array_equal(ex.memory(), run.memory)
    assert ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows


def test_whole_verifier_rejects_any_changed_word(setup_head):
    """recursion.py without evaluate_air_constraints: every word of the raw transcript is either absorbed by the sponge or checked to be
    zero padding, and every claim that is left (public input, domain-separator digest, the three constraint evaluations, the bytecode
    value) enters an equation: 40 random positions of the transcript and every claim field, one word each"""
    bc, children = setup_head
    raw, claim, ops, stmt, pub = children[0]
    rng = np.random.default_rng(77)
    for pos in list(rng.integers(0, raw.size, size=40)) + [0, 7, raw.size - 1]:
        raw2 = raw.copy()
        raw2[pos] ^= 1
        _rejected(bc, [(raw2, claim, ops, stmt, pub), children[1]])
    for field, k in (("bytecode_hash_domsep", 3), ("bytecode_value", 0), ("bytecode_value", 4)):
        st2 = capi.PcsStatementClaim.from_buffer_copy(stmt)
        getattr(st2, field)[k] ^= 1
        _rejected(bc, [(raw, claim, ops, st2, pub), children[1]])
    for t in range(3):
        st2 = capi.PcsStatementClaim.from_buffer_copy(stmt)
        st2.air_constraint_evals[t][t] ^= 1
        _rejected(bc, [(raw, claim, ops, st2, pub), children[1]])
    pub2 = np.asarray(pub).copy()
    pub2[2] ^= 1
    _rejected(bc, [(raw, claim, ops, stmt, pub2), children[1]])
    cl2 = capi.WhirOpeningClaim.from_buffer_copy(claim)
    cl2.folding_randomness[3] ^= 1
    _rejected(bc, [(raw, cl2, ops, stmt, pub), children[1]])


def test_recursion_whole_accepts_and_equals_oracle_vm(orc, setup_full):
    """`recursion()` of zkdsl_implem/recursion.py, every line of it: nothing about the child proof is taken on trust but the bytecode
    value (a hint in the reference too, reduced by main.py's bytecode-claim sumcheck)"""
    bc, children = setup_full
    pi, wit, _ = wv.build_witness(bc, children)
    ex = vm.execute(bc, pi, wit, n_threads=4)
    run = ob.VmRun(orc, bc, pi, wit)
    assert ex.n_cycles == run.pcs.size and np.array_equal(ex.pcs(), run.pcs) and np.array_equal(ex.fps(), run.fps)
    assert np.array_equal(ex.memory_defined(), run.defined) and np.array_equal(ex.memory(), run.memory)
    assert ex.counts == run.counts and ex.n_poseidon_calls == run.n_poseidon_calls and ex.n_extension_rows == run.n_extension_rows


def test_recursion_whole_rejects(setup_full):
    """with the constraint polynomials evaluated in the VM the three claimed evaluations are ignored (changing them changes nothing but
    the public input, which is recomputed); every transcript word and the remaining claims still decide acceptance"""
    bc, children = setup_full
    raw, claim, ops, stmt, pub = children[1]
    T = bc.info["shape"].statement
    rng = np.random.default_rng(78)
    for pos in list(rng.integers(0, raw.size, size=25)) + [T.off_inner[2] + 5 * 60, T.off_inner[1] + 9, T.off_inner[0] + 40]:
        raw2 = raw.copy()
        raw2[pos] ^= 1
        _rejected(bc, [children[0], (raw2, claim, ops, stmt, pub)])
    for field, k in (("bytecode_hash_domsep", 0), ("bytecode_value", 2)):
        st2 = capi.PcsStatementClaim.from_buffer_copy(stmt)
        getattr(st2, field)[k] ^= 1
        _rejected(bc, [children[0], (raw, claim, ops, st2, pub)])
    st2 = capi.PcsStatementClaim.from_buffer_copy(stmt)
    st2.air_constraint_evals[2][0] ^= 1          # no longer an input of any equation
    pi, wit, _ = wv.build_witness(bc, [children[0], (raw, claim, ops, st2, pub)])
    vm.execute(bc, pi, wit, n_threads=4)
