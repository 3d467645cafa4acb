"""GPU end-to-end: lmh_prove_execution (commit -> logup/GKR -> AIR -> WHIR open on the device) on a consistent synthetic
witness.  The proof must equal the oracle's proof word for word and be accepted by the oracle's verify_execution."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob
from tests import synth_witness

pytestmark = pytest.mark.gpu


def _lm_builder(b):
    """the oracle's 8-integer builder as the library's lm_whir_builder"""
    return lm.WhirBuilder.default(int(b[0]), max_num_variables_to_send_coeffs=int(b[1]), rs_domain_initial_reduction_factor=int(b[2]),
                                  folding_factor_first=int(b[3]), folding_factor_subsequent=int(b[4]), soundness_type=int(b[5]),
                                  security_level=int(b[6]), pow_bits=int(b[7]))


def _device_proof(ctx, orc, w, builder, device_counters=True):
    """prove on the device with the WHIR schedule of the library's own WhirConfig::new; every proof is also put through the
    library's verifier in its wire form (postcard bytes)"""
    tr, keep = lm.make_execution_trace(ctx, w, device_counters=device_counters)
    n = ctx.lib.lmh_stacked_n_vars(lm.capi.C.byref(tr))
    cfg = lm.WhirConfig.new(_lm_builder(builder), n)
    assert cfg.to_dict() == {k: v for k, v in ob.whir_config(orc, builder, n).items() if k != "final_log_inv_rate"} | {
        "rounds": [{k: r[k] for k in ("query_pow_bits", "folding_pow_bits", "num_queries", "ood_samples")} for r in ob.whir_config(orc, builder, n)["rounds"]]}
    pr = lm.Prover(ctx)
    pr.prove_execution(tr, cfg)
    ok, err = lm.verify_execution(w, pr.proof_bytes(), _lm_builder(builder))
    assert ok, err
    return pr.proof()


def test_prove_execution_matches_oracle_and_verifies(ctx, orc):
    rng = np.random.default_rng(0)
    w = synth_witness.build(orc, rng, n_calls=40)
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    ref = ob.prove_execution(orc, w, synth_witness.header(w), b)
    proof = _device_proof(ctx, orc, w, b)
    ok, err = ob.verify_execution(orc, w, proof, b)
    assert ok, err
    assert proof.size == ref.size and np.array_equal(proof, ref)
    # the access counters computed on the device (prove_execution.rs:90-110) equal the witness generator's
    assert np.array_equal(_device_proof(ctx, orc, w, b, device_counters=False), ref)


def test_access_counts_match_histogram(ctx):
    """lm_access_counts vs numpy: uniform addresses, heavily repeated addresses (whole waves on one address, as padding rows
    are), multi-word lookups, and out-of-range rows that must be ignored."""
    rng = np.random.default_rng(4)
    P = 0x7F000001
    length = 1 << 12
    to_m = lambda a: ((a.astype(np.uint64) << np.uint64(32)) % np.uint64(P)).astype(np.uint32)  # noqa: E731
    a1 = rng.integers(0, length - 16, size=5000)
    a2 = np.concatenate([np.full(3000, 7), rng.integers(0, 64, size=1000), np.full(777, length - 4)])
    a3 = np.concatenate([rng.integers(0, length, size=300), [length - 1, length - 2, length + 5]])  # last rows overflow with n_values 4
    jobs, want, bad = [], np.zeros(length, dtype=np.int64), 0
    for addr, nv in ((a1, 1), (a2, 4), (a3, 4), (a1, 16)):
        jobs.append((ctx.to_device(to_m(addr)), addr.size, nv))
        ok = addr + nv <= length
        bad += int(np.sum(~ok))
        for j in range(nv):
            want += np.bincount(addr[ok] + j, minlength=length)
    ctx.access_errors(reset=True)
    got = ctx.access_counts(length, jobs).download()
    assert np.array_equal(got, to_m(want))
    assert bad >= 3 and ctx.access_errors() == bad   # the skipped rows are counted (the reference would panic on them)
    assert ctx.access_errors() == 0                  # (reset by the previous call)
    assert np.array_equal(ctx.access_counts(64, []).download(), np.zeros(64, dtype=np.uint32))


def test_access_counts_hot_window(ctx):
    """the shape of a real trace: most lookups scattered, but one window (public input, constants, the zero vector) receives
    a large share — hot addresses interleaved with real ones inside a wave, runs that straddle the window's end — so that
    window's list is split between several workgroups."""
    rng = np.random.default_rng(5)
    P = 0x7F000001
    length = (1 << 16) + 100
    to_m = lambda a: ((a.astype(np.uint64) << np.uint64(32)) % np.uint64(P)).astype(np.uint32)  # noqa: E731
    n = 200000
    scattered = rng.integers(0, length - 16, size=n)
    low = rng.integers(8192 - 40, 2 * 8192, size=n)           # window 1, some runs end in window 2
    hot = np.where(rng.random(n) < 0.3, 8192 + 64, np.where(rng.random(n) < 0.2, 8192 + 65, scattered))
    jobs, want = [], np.zeros(length, dtype=np.int64)
    for addr, nv in ((scattered, 1), (low, 16), (hot, 5), (low, 1), (hot, 1)):
        jobs.append((ctx.to_device(to_m(addr)), addr.size, nv))
        for j in range(nv):
            want += np.bincount(addr + j, minlength=length)
    got = ctx.access_counts(length, jobs).download()
    assert np.array_equal(got, to_m(want))


def test_prove_execution_production_parameters_verifies(ctx, orc):
    """default_whir_config (124-bit, 16 PoW bits, lean_prover/src/lib.rs:22-50), different table heights, bigger program:
    too slow for the oracle PROVER, checked by the oracle VERIFIER."""
    rng = np.random.default_rng(1)
    w = synth_witness.build(orc, rng, n_calls=900, n_blocks=32, log_exec=11, log_pos=10, log_ext=8, log_memory=16, log_bytecode=10)
    proof = _device_proof(ctx, orc, w, ob.whir_builder(log_inv_rate=1))
    ok, err = ob.verify_execution(orc, w, proof, None)
    assert ok, err


def test_inconsistent_witness_is_rejected(ctx, orc):
    rng = np.random.default_rng(2)
    w = synth_witness.build(orc, rng, n_calls=40)
    w["memory_acc"] = w["memory_acc"].copy()
    w["memory_acc"][200] = int(orc.to_monty(77))  # wrong access count -> logup sum != 0
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    with pytest.raises(lm.LmError):
        _device_proof(ctx, orc, w, b, device_counters=False)
    # with the counters recomputed on the device the same witness is fine again ...
    _device_proof(ctx, orc, w, b, device_counters=True)
    # ... but a memory word that no longer matches what a table looked up is not (logup sum != 0)
    w["memory"] = w["memory"].copy()
    hot = int(np.argmax(orc.from_monty_fast(w["memory_acc"]) > 0))
    w["memory"][hot] = (int(w["memory"][hot]) + 1) % 0x7F000001
    with pytest.raises(lm.LmError):
        _device_proof(ctx, orc, w, b, device_counters=True)
    # a lookup that leaves the memory image is an error of its own (the reference panics in its counting loop)
    w2 = synth_witness.build(orc, np.random.default_rng(2), n_calls=40)
    t0 = w2["tables"][0] = w2["tables"][0].copy()
    t0[2][5] = int(orc.to_monty(np.array([w2["memory"].size + 3]))[0])
    with pytest.raises(lm.LmError, match="outside the memory"):
        _device_proof(ctx, orc, w2, b, device_counters=True)


def test_concurrent_provers_are_independent(orc):
    """Three lm_ctx (one stream each) driven from three host threads on the same GPU — the bench's proofs-in-flight mode.
    Every proof must equal the one the same prover produces alone."""
    import threading
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    ctxs, traces, cfgs, alone = [], [], [], []
    for c in range(3):
        ctx = lm.Context(0)
        w = synth_witness.build(orc, np.random.default_rng(50 + c), n_calls=30 + 7 * c)
        tr, keep = lm.make_execution_trace(ctx, w)
        cfg = lm.WhirConfig.from_dict(ob.whir_config(orc, b, ctx.lib.lmh_stacked_n_vars(lm.capi.C.byref(tr))))
        pr = lm.Prover(ctx)
        pr.prove_execution(tr, cfg)
        ctxs.append(ctx), traces.append((tr, keep)), cfgs.append(cfg), alone.append(pr.proof())
    got = [[] for _ in range(3)]
    start = threading.Barrier(3)

    def worker(c):
        start.wait()
        for _ in range(4):
            pr = lm.Prover(ctxs[c])
            pr.prove_execution(traces[c][0], cfgs[c])
            got[c].append(pr.proof())

    th = [threading.Thread(target=worker, args=(c,)) for c in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in range(3):
        assert len(got[c]) == 4
        for pf in got[c]:
            assert np.array_equal(pf, alone[c])


def test_pruned_proof_matches_oracle_and_restores(ctx, orc):
    """Merkle-path pruning of the device proof (lmh_proof_pruned_*, merkle_pruning.rs:18-86): word-identical to the oracle's
    restatement, restores to the un-pruned proof, which the oracle verifier accepts; Proof::proof_size_fe agrees."""
    rng = np.random.default_rng(3)
    w = synth_witness.build(orc, rng, n_calls=120, n_blocks=16, log_exec=9, log_pos=8)
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    tr, keep = lm.make_execution_trace(ctx, w)
    cfg = lm.WhirConfig.from_dict(ob.whir_config(orc, b, ctx.lib.lmh_stacked_n_vars(lm.capi.C.byref(tr))))
    pr = lm.Prover(ctx)
    pr.prove_execution(tr, cfg)
    full, pruned, sizes = pr.proof(), pr.proof_pruned(), pr.batch_sizes()
    assert sizes.sum() == full[1 + full[0]] and len(sizes) >= 2  # one batch per queried commitment
    assert np.array_equal(pruned, ob.prune_proof(orc, full, sizes))
    restored = ob.restore_proof(orc, pruned)
    assert np.array_equal(restored, full)
    ok, err = ob.verify_execution(orc, w, restored, b)
    assert ok, err
    fe = pr.proof_size_fe()
    assert fe == ob.pruned_size_fe(orc, pruned) and fe < full.size


@pytest.mark.parametrize("log_inv_rate", [2, 3])
def test_prove_execution_other_rates_verify(ctx, orc, log_inv_rate):
    """BASELINE config 3 (rate 1/4) and rate 1/8: the WHIR schedule (queries, folding, PoW) changes with the rate; the device
    proof equals the oracle's at reduced security and is accepted at the production parameters."""
    rng = np.random.default_rng(20 + log_inv_rate)
    w = synth_witness.build(orc, rng, n_calls=60)
    w["log_inv_rate"] = log_inv_rate
    small = ob.whir_builder(log_inv_rate=log_inv_rate, pow_bits=6, security=60)
    ref = ob.prove_execution(orc, w, synth_witness.header(w), small)
    proof = _device_proof(ctx, orc, w, small)
    assert np.array_equal(proof, ref)
    prod = _device_proof(ctx, orc, w, ob.whir_builder(log_inv_rate=log_inv_rate))
    ok, err = ob.verify_execution(orc, w, prod, None)
    assert ok, err


def test_mixed_program_matches_oracle(ctx, orc):
    """ADD / MUL / DEREF instructions and every ExtensionOp mode (active rows in the 42-column degree-6 AIR, non-trivial
    bus and memory lookups for all three tables): device proof == oracle proof, word for word."""
    rng = np.random.default_rng(5)
    w = synth_witness.build_mixed(orc, rng)
    b = ob.whir_builder(log_inv_rate=1, pow_bits=6, security=60)
    ref = ob.prove_execution(orc, w, synth_witness.header(w), b)
    proof = _device_proof(ctx, orc, w, b)
    ok, err = ob.verify_execution(orc, w, proof, b)
    assert ok, err
    assert proof.size == ref.size and np.array_equal(proof, ref)


def test_recursion_shaped_tables_verify(ctx, orc):
    """BASELINE configs[3] stand-in, reduced (the full derived size: test_recursion_derived_shape_equals_oracle_prover): the ExtensionOp table is as tall as the execution table (the reference requires execution >= every table, stacked_pcs.rs:111; ~2^15 active rows of long
    dot products / poly_eq chains), the Poseidon table 32x shorter — the batched AIR sumcheck starts on the execution and
    ExtensionOp AIRs and the Poseidon AIR joins late.  Production parameters; checked by the oracle verifier."""
    rng = np.random.default_rng(6)
    ext = [("mul", False, 64, 200), ("mul", True, 128, 60), ("poly_eq", False, 20, 300), ("add", False, 1, 500), ("poly_eq", True, 9, 100)]
    w = synth_witness.build(orc, rng, n_calls=500, n_blocks=32, log_exec=15, log_pos=10, log_ext=15, log_memory=18, log_bytecode=12,
                            n_arith=600, ext_calls=ext)
    assert sum(s * c for _, _, s, c in ext) > (1 << 14)
    w["log_inv_rate"] = 2
    proof = _device_proof(ctx, orc, w, ob.whir_builder(log_inv_rate=2))
    ok, err = ob.verify_execution(orc, w, proof, None)
    assert ok, err


def test_recursion_derived_shape_equals_oracle_prover(ctx, orc):
    """BASELINE configs[3] (`recursion --n 4 --log-inv-rate 2`) stand-in at the FULL size tools/recursion_shape.py derives by counting
    the in-VM verifier's work for four children of 775 signatures (execution 2^19, ExtensionOp 2^18 with the derived mix of leaf folds /
    eq chains / single products, Poseidon16 2^16, memory 2^21; rate 1/4, production parameters): the device proof equals the oracle
    PROVER's proof word for word (a few minutes of oracle time on 16 threads)."""
    import bench
    d = bench.recursion_shape()
    assert d["shape"] == dict(log_exec=19, log_pos=16, log_ext=18, log_memory=21, log_bytecode=19) and d["child"]["stacked_n_vars"] == 25
    w = bench.build_workload(ctx, orc, ob, np.random.default_rng(11), 0, 2, "recursion", False, "synthetic")
    assert w["w"]["log_rows"] == {0: 19, 1: 18, 2: 16}
    proof = bench.run_step(ctx, lm, w).proof()
    ob.set_threads(orc, 16)
    ref = ob.prove_execution(orc, w["w"], synth_witness.header(w["w"]), bench.oracle_builder(ob, w))
    assert proof.size == ref.size and np.array_equal(proof, ref)


def test_extension_op_trace_gather(ctx, orc):
    """lm_extension_op_trace == fill_trace_extension_op (extension_op/exec.rs:192-203) on the mixed witness and on random
    addresses including the last valid one and an overflowing one (zeros)."""
    rng = np.random.default_rng(7)
    w = synth_witness.build_mixed(orc, rng)
    ext, mem = w["tables"][1], w["memory"]
    n = ext.shape[1]
    d_mem, d_idx = ctx.to_device(mem), ctx.to_device(ext[6])
    va = [ctx.alloc(n) for _ in range(5)]
    ctx.extension_op_trace(d_mem, mem.size, d_idx, va, n)
    for k in range(5):
        assert np.array_equal(va[k].download(), ext[14 + k])
    addr = np.concatenate([rng.integers(0, mem.size - 5, size=3000), [mem.size - 5, mem.size - 4, mem.size + 9]])
    d_idx = ctx.to_device(orc.to_monty(addr))
    va = [ctx.alloc(addr.size) for _ in range(5)]
    ctx.extension_op_trace(d_mem, mem.size, d_idx, va, addr.size)
    padded = np.concatenate([mem, np.zeros(32, dtype=np.uint32)])
    for k in range(5):
        assert np.array_equal(va[k].download(), padded[np.minimum(addr + k, mem.size + 16)])
    ctx.extension_op_trace(d_mem, mem.size, d_idx, va, 0)


@pytest.mark.parametrize("log_inv_rate", [1, 2])
def test_full_size_proof_verifies_and_is_deterministic(ctx, orc, log_inv_rate):
    """BASELINE configs[1] (rate 1/2) and configs[2] (rate 1/4) at their FULL size on the default workload of bench.py (the
    aggregation program run by lmh_execute_bytecode on 1550 real signatures: Poseidon table 2^18, execution 2^20, memory 2^22,
    stacked polynomial 2^26, logup 2^25, production WHIR parameters): size-independent properties — the oracle's VERIFIER accepts
    the proof (every sumcheck, GKR layer, Merkle path and PoW witness of the real size), proving twice gives the same words, and
    the pruned wire form restores to the proof."""
    import bench
    w = bench.build_vm_workload(ctx, np.random.default_rng(77), bench.N_SIGS, log_inv_rate, False)
    assert w["n_vars"] == 26 and w["w"]["log_bytecode"] == 19 and w["w"]["log_rows"] == {0: 20, 1: 15, 2: 18}
    p1 = bench.run_step(ctx, lm, w)
    proof = p1.proof()
    ok, err = ob.verify_execution(orc, w["w"], proof, None)
    assert ok, err
    p2 = bench.run_step(ctx, lm, w)
    assert np.array_equal(p2.proof(), proof)
    # the wire form: lz4-framed postcard bytes through the library's own verifier (default_whir_config read off the proof)
    wire = p1.proof_bytes(compressed=True)
    ok, err = lm.verify_execution(w["w"], wire, compressed=True)
    assert ok, err
    assert 4.0 < len(p1.proof_bytes()) / p1.proof_size_fe() < 4.9  # varints: 87.5 % of random Montgomery words need 5 bytes, flags / zeros 1
    bad = dict(w["w"], public_input=np.roll(w["w"]["public_input"], 1))
    assert not lm.verify_execution(bad, wire, compressed=True)[0]
    pruned = p1.proof_pruned()
    assert np.array_equal(ob.restore_proof(orc, pruned), proof)
    assert p1.proof_size_fe() == ob.pruned_size_fe(orc, pruned) < proof.size
    # the whole node (VM run + trace + proof in one call) produces the same proof
    from leanmultisig_amd import vm
    p3 = lm.Prover(ctx)
    vm.prove_execution_vm(ctx, p3, w["vm"]["bc"], w["vm"]["pi"], w["vm"]["wit"], w["lm_builder"])
    assert np.array_equal(p3.proof(), proof)


@pytest.mark.parametrize("log_inv_rate,capacity", [(1, False), (2, False), (1, True)])
def test_full_size_proof_equals_oracle_prover(ctx, orc, log_inv_rate, capacity):
    """BASELINE configs[1] (rate 1/2), configs[2] (rate 1/4) and the `prox-gaps-conjecture` regime (CapacityBound, the README's
    176 KiB row) at FULL size on the default workload of bench.py (1550 real XMSS signatures, stacked 2^26, logup domain 2^25,
    memory 2^22, production WHIR parameters): the device proof — of the trace the library's own runner and device trace builder
    produced — equals, word for word, the proof of the oracle PROVER on the trace of the oracle's runner + get_execution_trace:
    every root, round polynomial, PoW witness, query answer and sibling of the full-size schedule (~1-2 minutes of oracle time
    on 16 OpenMP threads each)."""
    import bench
    w = bench.build_vm_workload(ctx, np.random.default_rng(5), bench.N_SIGS, log_inv_rate, capacity)
    assert w["n_vars"] == 26
    proof = bench.run_step(ctx, lm, w).proof()
    ob.set_threads(orc, 16)
    full = bench.oracle_witness(orc, ob, w)
    ref = ob.prove_execution(orc, full, synth_witness.header(full), bench.oracle_builder(ob, w))
    assert proof.size == ref.size and np.array_equal(proof, ref)


def test_execution_table_trace_matches_oracle(ctx, orc):
    """lm_execution_table_trace == the oracle's restatement of get_execution_trace's main loop (trace_gen.rs:27-100): on the
    mixed program's VM log, and on random instruction rows that reach every branch (fp-relative operands, DEREF through
    value_a, immediates, out-of-range addresses and pcs -> 0)."""
    rng = np.random.default_rng(11)
    w = synth_witness.build_mixed(orc, rng)
    pcs, fps = synth_witness.vm_log(w)

    def device(pcs, fps, bytecode, memory):
        bufs = [ctx.alloc(pcs.size) for _ in range(24)]
        ctx.execution_table_trace(ctx.to_device(pcs), ctx.to_device(fps), pcs.size, ctx.to_device(bytecode.reshape(-1)), bytecode.shape[0],
                                  ctx.to_device(memory), memory.size, bufs)
        return np.stack([b.download() for b in bufs])

    assert np.array_equal(device(pcs, fps, w["bytecode"], w["memory"]), ob.execution_table_fill(orc, pcs, fps, w["bytecode"], w["memory"]))
    # random rows: flags in {0, 1}, aux in {0, 1, 2}, operands small, fp up to the memory size
    n_rows, n, mem_len = 512, 20000, 1 << 12
    M = lambda x: orc.to_monty(np.asarray(x, dtype=np.uint64))  # noqa: E731
    bc = np.zeros((n_rows, 16), dtype=np.uint32)
    bc[:, 0:3] = M(rng.integers(0, 64, size=(n_rows, 3)))
    bc[:, 3:8] = M(rng.integers(0, 2, size=(n_rows, 5)))
    bc[:, 8:10] = M(rng.integers(0, 2, size=(n_rows, 2)))
    bc[:, 10] = M(rng.integers(0, 3, size=n_rows))
    bc[:, 11] = ob.rand_field(rng, n_rows)
    memory = ob.rand_field(rng, mem_len)
    memory[: mem_len // 2] = M(rng.integers(0, mem_len + 50, size=mem_len // 2))   # pointers for DEREF, some out of range
    pcs = rng.integers(0, n_rows + 3, size=n).astype(np.uint32)                    # a few pcs past the table
    fps = rng.integers(0, mem_len, size=n).astype(np.uint32)
    got, want = device(pcs, fps, bc, memory), ob.execution_table_fill(orc, pcs, fps, bc, memory)
    # is_precompile: the device evaluates the AIR's polynomial, the oracle the instruction kind; they agree on every VALID
    # encoding (at most one of add / mul / deref / jump), random rows are not all valid
    valid = ((orc.from_monty_fast(want[16]) + orc.from_monty_fast(want[17]) + (orc.from_monty_fast(want[18]) > 0)) <= 1)
    assert valid.sum() > n // 4
    for c in range(24):
        sel = valid if c == 20 else slice(None)
        assert np.array_equal(got[c][sel], want[c][sel]), c


def test_poseidon_outputs_from_memory(ctx, orc):
    """lm_poseidon_trace_outputs_from_memory (trace_gen.rs:118-147) vs the oracle's restatement: random flags (permute / half
    output), result pointers anywhere in memory including the last words (out-of-range reads give 0)."""
    import ctypes
    rng = np.random.default_rng(12)
    n, mem_len = 3000, 1 << 12
    memory = ob.rand_field(rng, mem_len)
    rows = ob.rand_field(rng, (n, 109))
    rows[:, 8] = orc.to_monty(rng.integers(0, 2, size=n))
    rows[:, 3] = orc.to_monty(rng.integers(0, 2, size=n))
    res = rng.integers(0, mem_len, size=n)
    res[:4] = [mem_len - 1, mem_len - 9, mem_len - 16, 0]
    rows[:, 2] = orc.to_monty(res)
    want = np.ascontiguousarray(rows.copy())
    orc.lib.orc_poseidon16_outputs_from_memory(want.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n),
                                               memory.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(mem_len))
    assert not np.array_equal(want, rows)
    cols = [ctx.to_device(np.ascontiguousarray(rows[:, c])) for c in range(109)]
    ctx.poseidon_trace_outputs_from_memory(cols, n, ctx.to_device(memory), mem_len)
    got = np.stack([c.download() for c in cols], axis=1)
    assert np.array_equal(got, want)


def test_config0_full_shape_production_parameters_equals_oracle(ctx, orc):
    """BASELINE configs[0] (`xmss --n-signatures 128 --log-inv-rate 1`, the top size of the reference's test_aggregation,
    tests/test_multisignatures.rs:31-41) at its FULL shape and at the PRODUCTION WHIR parameters (default_whir_config:
    124-bit JohnsonBound, 16 PoW bits, fold 7/5, lean_prover/src/lib.rs:22-50): 128 x 167 Poseidon calls -> Poseidon table
    2^15, execution 2^17, bytecode 2^19 (rec_aggregation/TYPE1_TYPE2_LAYOUT.md:9), memory 2^19 (>= bytecode,
    prove_execution.rs:41-45), stacked polynomial 2^23.  The device proof must equal the oracle prover's word for word —
    every root, sumcheck coefficient, PoW witness, query leaf and sibling of the full-security schedule."""
    rng = np.random.default_rng(128)
    w = synth_witness.build(orc, rng, n_calls=128 * 167, n_blocks=1024, log_exec=17, log_pos=15, log_ext=8, log_memory=19,
                            log_bytecode=19)
    b = ob.whir_builder(log_inv_rate=1)
    proof = _device_proof(ctx, orc, w, b)
    ref = ob.prove_execution(orc, w, synth_witness.header(w), b)
    assert proof.size == ref.size and np.array_equal(proof, ref)
    ok, err = ob.verify_execution(orc, w, proof, None)
    assert ok, err


def test_mixed_program_2p15_production_parameters_equals_oracle(ctx, orc):
    """The mixed program (Poseidon calls, ADD / MUL / DEREF, every ExtensionOp mode with ~2^14 active rows) on 2^15-row
    tables at the production WHIR parameters, rate 1/4 (the recursion configs' rate): device proof == oracle proof."""
    rng = np.random.default_rng(215)
    ext = [("mul", False, 64, 100), ("mul", True, 128, 30), ("poly_eq", False, 20, 150), ("add", False, 1, 250),
           ("poly_eq", True, 9, 50), ("add", True, 2, 40), ("mul", False, 1, 60)]
    w = synth_witness.build(orc, rng, n_calls=20000, n_blocks=256, log_exec=15, log_pos=15, log_ext=15, log_memory=18,
                            log_bytecode=15, n_arith=6000, ext_calls=ext)
    w["log_inv_rate"] = 2
    b = ob.whir_builder(log_inv_rate=2)
    proof = _device_proof(ctx, orc, w, b)
    ref = ob.prove_execution(orc, w, synth_witness.header(w), b)
    assert proof.size == ref.size and np.array_equal(proof, ref)
    ok, err = ob.verify_execution(orc, w, proof, None)
    assert ok, err


def test_pad_table_matches_generator(ctx, orc):
    """lmh_pad_table (pad_table, trace_gen.rs:170-191) against the padding rows the witness generator wrote from the reference's
    executor semantics (execution/mod.rs:59-74, extension_op/mod.rs:125-134, poseidon_16/mod.rs:182-205): the padded region of
    every committed column is overwritten with garbage on the device, padded again by the library, the Poseidon rows
    re-derived by lm_poseidon_trace + lm_poseidon_trace_outputs_from_memory, and must equal the generator's table."""
    rng = np.random.default_rng(21)
    w = synth_witness.build_mixed(orc, rng)
    Z, NULL = 64, 96   # synth_witness: zero vector, null hash
    assert not w["memory"][Z:Z + 16].any() and w["memory"][NULL:NULL + 8].any()
    d_mem = ctx.to_device(w["memory"])
    for t in (0, 1, 2):
        tab = np.ascontiguousarray(w["tables"][t])
        n_cols, n = (20, 29, 109)[t], tab.shape[1]
        tab = tab[:n_cols]
        same = np.all(tab == tab[:, -1:], axis=0)               # rows equal to the last (padding) row
        n_active = n - int(np.argmin(same[::-1])) if not same.all() else 0
        assert 0 < n_active < n and ctx.lib.lmh_table_log_rows(n_active) <= int(np.log2(n))
        dirty = tab.copy()
        dirty[:, n_active:] = ob.rand_field(rng, (n_cols, n - n_active))
        cols = [ctx.to_device(np.ascontiguousarray(dirty[c])) for c in range(n_cols)]
        ctx.pad_table(t, cols, n_active, int(np.log2(n)), Z, NULL, w["ending_pc"])
        if t == 2:
            ctx.poseidon_trace(cols, n)
            ctx.poseidon_trace_outputs_from_memory(cols, n, d_mem, w["memory"].size)
        got = np.stack([c.download() for c in cols])
        assert np.array_equal(got, tab), f"table {t}: columns {sorted(set(np.nonzero(got != tab)[0]))}"
    with pytest.raises(lm.LmError):
        ctx.pad_table(0, cols[:20], 1 << 8, 8, Z, NULL, 0)      # no room for a padding row
