"""GPU parity: batched AIR sumcheck (3 leanVM tables) through the C ABI vs the CPU oracle.  The oracle evaluates the
constraint polynomials in the literal textbook form over EF; the device uses base-field first rounds, the sparse Poseidon
partial rounds and lazily loaded columns — transcripts must still be word-identical."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob
from tests.oracle_binding import P, rand_field

pytestmark = pytest.mark.gpu


def _challenges(rng):
    return rand_field(rng, 5), rand_field(rng, (16, 5)), rand_field(rng, 5), rand_field(rng, 5)


def _run(ctx, orc, tables, alpha, eq16, beta, eta):
    ref_proof, ref_pt, ref_ev = ob.air_prove(orc, tables, alpha, eq16, beta, eta)
    pr = lm.Prover(ctx)
    dev_tables = []
    keep = []
    for t in tables:
        bufs = [ctx.to_device(c) for c in t["cols"]]
        keep.append(bufs)
        dev_tables.append(dict(table=t["table"], log_rows=t["log_rows"], cols=bufs, eq_point=t["eq_point"], sum=t["sum"],
                               non_padded_n_rows=t.get("non_padded_n_rows", 0)))
    pt, evs = pr.prove_batched_air_sumcheck(dev_tables, alpha, eq16, beta, eta)
    proof = pr.proof()
    assert np.array_equal(pt, ref_pt)
    assert np.array_equal(np.concatenate(evs), ref_ev)
    assert np.array_equal(proof, ref_proof)
    return proof


@pytest.mark.parametrize("table,log_rows", [(0, 1), (0, 2), (1, 2), (2, 2), (0, 5), (1, 4), (2, 3), (2, 6)])
def test_single_table_random_columns(ctx, orc, table, log_rows):
    """Random (unsatisfied) columns: checks the constraint POLYNOMIALS, every alpha power and the fold schedule."""
    rng = np.random.default_rng(table * 10 + log_rows)
    alpha, eq16, beta, eta = _challenges(rng)
    cols = rand_field(rng, (ob.AIR_N_COLUMNS[table], 1 << log_rows))
    t = dict(table=table, log_rows=log_rows, cols=cols, eq_point=rand_field(rng, (log_rows, 5)), sum=rand_field(rng, 5))
    _run(ctx, orc, [t], alpha, eq16, beta, eta)


@pytest.mark.parametrize("table,log_rows,n_active", [(0, 6, 1), (0, 6, 37), (0, 6, 62), (0, 6, 63), (0, 11, 1500), (1, 5, 9), (1, 9, 300),
                                                     (2, 4, 5), (2, 7, 100), (2, 10, 513), (0, 3, 8)])
def test_active_prefix_equals_full_sum(ctx, orc, table, log_rows, n_active):
    """AirSumcheckSession's active prefix (air_sumcheck.rs:194-200,236-240): random (unsatisfied) columns whose rows behind
    n_active all equal one random padding row — the constraint value there is an arbitrary non-zero constant.  With
    non_padded_n_rows the device evaluates the active pairs and ONE padding pair weighted by the tail of the eq weights; the
    transcript must equal the oracle's, which sums every row.  Boundaries: one active row, an odd count, a single padding
    pair left, no full padding pair (falls back to the full sum), all rows active."""
    rng = np.random.default_rng(1000 * table + 10 * log_rows + n_active)
    alpha, eq16, beta, eta = _challenges(rng)
    n = 1 << log_rows
    cols = rand_field(rng, (ob.AIR_N_COLUMNS[table], n))
    cols[:, n_active:] = cols[:, -1:]
    t = dict(table=table, log_rows=log_rows, cols=cols, eq_point=rand_field(rng, (log_rows, 5)), sum=rand_field(rng, 5),
             non_padded_n_rows=n_active)
    _run(ctx, orc, [t], alpha, eq16, beta, eta)


def test_three_tables_back_loaded_and_verified(ctx, orc):
    """Three tables of different heights (back-loaded batching, air_sumcheck.rs:636-681) with TRUE sums, so that the
    oracle's restatement of the verifier side (verify_execution.rs:109-170) accepts the device transcript."""
    rng = np.random.default_rng(99)
    alpha, eq16, beta, eta = _challenges(rng)
    specs = [(0, 6), (2, 5), (1, 3)]  # descending height
    tables = []
    for table, lr in specs:
        cols = ob.poseidon_table(orc, rng, lr, n_active=(1 << lr) - 3) if table == 2 else rand_field(rng, (ob.AIR_N_COLUMNS[table], 1 << lr))
        eqp = rand_field(rng, (lr, 5))
        vals = ob.air_eval_rows(orc, table, cols, alpha, eq16, beta)
        eq = orc.eq_table(eqp)
        s = np.zeros(5, dtype=np.uint32)
        for r in range(1 << lr):
            s = ((s.astype(np.uint64) + orc.ef_mul(eq[r], vals[r])) % P).astype(np.uint32)
        tables.append(dict(table=table, log_rows=lr, cols=cols, eq_point=eqp, sum=s))
    proof = _run(ctx, orc, tables, alpha, eq16, beta, eta)
    ok, err = ob.air_verify(orc, tables, alpha, eq16, beta, eta, proof)
    assert ok, err
    bad = proof.copy()
    bad[7] ^= 1
    assert not ob.air_verify(orc, tables, alpha, eq16, beta, eta, bad)[0]


def test_poseidon_trace_satisfies_all_but_bus(ctx, orc):
    """On a generated Poseidon trace every constraint except the bus column vanishes: with alpha^0 = 1 on the bus the
    row value equals the bus expression alone (sanity of the trace generator against the AIR, trace_gen.rs:44-112)."""
    rng = np.random.default_rng(5)
    alpha, eq16, beta, _ = _challenges(rng)
    cols = ob.poseidon_table(orc, rng, 3)
    vals = ob.air_eval_rows(orc, 2, cols, alpha, eq16, beta)
    alpha2 = rand_field(rng, 5)
    vals2 = ob.air_eval_rows(orc, 2, cols, alpha2, eq16, beta)
    assert np.array_equal(vals, vals2)  # independent of alpha => constraints 1..99 are all zero


def test_poseidon_trace_rows_match_oracle(ctx, orc):
    """a21: device trace generation == oracle generate_trace_rows_for_perm, including permute-mode rows."""
    rng = np.random.default_rng(21)
    n = 1000
    ref = ob.poseidon_table(orc, rng, 10, n_active=n)  # (109, 1024) column major, padding rows included
    # make some rows permute-mode (flag_permute = 1): regenerate those rows with the oracle
    rows = np.ascontiguousarray(ref.T)
    rows[5:50, 8] = 0x01FFFFFE
    import ctypes
    orc.lib.orc_poseidon16_fill_rows(rows.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(rows.shape[0]))
    ref = np.ascontiguousarray(rows.T)
    bufs = [ctx.to_device(ref[c] if c < 25 else np.zeros(1024, dtype=np.uint32)) for c in range(109)]
    ctx.poseidon_trace(bufs, 1024)
    got = np.stack([b.download() for b in bufs])
    assert np.array_equal(got, ref)


def test_poseidon_table_large_paths(ctx, orc):
    """2^14 rows: the base-field round takes the one-launch-per-segment path (2^13 pairs), the extension-field rounds the
    multi-workgroup combined kernel with the XCD-aware id mapping, virtual columns are folded over 13 rounds — the same
    launch shapes as the config-2 table, still word-identical to the textbook oracle."""
    rng = np.random.default_rng(314)
    alpha, eq16, beta, eta = _challenges(rng)
    lr = 14
    cols = rand_field(rng, (ob.AIR_N_COLUMNS[2], 1 << lr))
    t = dict(table=2, log_rows=lr, cols=cols, eq_point=rand_field(rng, (lr, 5)), sum=rand_field(rng, 5))
    _run(ctx, orc, [t], alpha, eq16, beta, eta)


def test_small_rounds_without_the_cooperative_kernels():
    """LM_AIR_NO_COOP=1: the small extension-field rounds run on the one-lane-per-evaluation kernel (the path every round took
    before the 16-lane Poseidon evaluation and the four-part ExtensionOp list) — same round polynomials.  The switch is read once
    per process, so the parity tests above are re-run in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LM_AIR_NO_COOP="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_air_gpu.py", "-k",
                        "test_single_table_random_columns or test_active_prefix_equals_full_sum or test_three_tables_back_loaded_and_verified"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]


def test_rounds_with_every_fold_materialised():
    """LM_AIR_NO_LAZY_FOLD=1: the first fold of a session writes the folded extension-field columns (k_air_fold_base) and round 1
    reads them, as every round did before round 1 read the base columns through the first challenge (FoldCols, k_air_fold2_base) —
    same round polynomials.  The switch is read once per process: the parity tests above are re-run in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LM_AIR_NO_LAZY_FOLD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_air_gpu.py", "-k",
                        "test_single_table_random_columns or test_active_prefix_equals_full_sum or test_three_tables_back_loaded_and_verified"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]
