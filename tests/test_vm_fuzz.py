"""Differential fuzzing of the leanVM runners (tools/vm_fuzz.py): seeded random programs — every instruction kind and operand mode,
every hint, the Poseidon16 / ExtensionOp variants, deferred writes, writes into other frames, deref chains, digests read while still
pending, several parallel loops, injected faults — must give the same result (cycle log, memory, defined mask, counts, precompile
records) and the same RunnerError on the host runner (calls executed at once / deferred), on the device runner and on the oracle's
sequential restatement of runner.rs (oracle/vm_oracle.hpp).  10^4 programs on the CPU (LM_FUZZ_N overrides), 10^3 with the device."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import vm_fuzz  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402


def test_host_runners_equal_oracle_on_random_programs(orc):
    n = int(os.environ.get("LM_FUZZ_N", 10000))
    bad = [r for r in (vm_fuzz.run_all(s, orc, ob) for s in range(n)) if r]
    assert not bad, f"{len(bad)} of {n} programs disagree:\n" + "\n".join(bad[:10])


def test_generator_reaches_what_it_claims(orc):
    """the corpus is not degenerate: valid programs succeed, every fault kind fails somewhere behind iteration 0, second batches, deferred
    calls and the literal re-run are all exercised"""
    import ctypes

    import leanmultisig_amd as lm
    from leanmultisig_amd import capi, vm
    seen = dict(ok=0, two_batches=0, faults=set(), in_segment=0)
    for s in range(600):
        bc, pi, w, meta = vm_fuzz.gen(s)
        try:
            ex = vm.execute(bc, pi, w, n_threads=2)
        except lm.LmError as e:
            assert meta["fault"], f"seed {s}: a program without a fault failed: {e}"
            seen["faults"].add(meta["fault"])
            seen["in_segment"] += "ParallelSegmentFailed" in str(e)
            continue
        assert not meta["fault"] or meta["fault"] == "assert_in_one_iteration" and meta["n1"] < 2, f"seed {s}: fault {meta['fault']} went unnoticed"
        info = vm.VmRunInfo()
        capi.load().lmh_execution_info(ex.h, ctypes.byref(info))
        seen["ok"] += 1
        seen["two_batches"] += info.n_host_batches >= 2
    assert seen["ok"] > 400 and seen["two_batches"] > 40 and seen["faults"] == set(vm_fuzz.FAULTS) and seen["in_segment"] > 5, seen


@pytest.mark.gpu
def test_device_runner_equals_host_and_oracle_on_random_programs(ctx, orc):
    n = int(os.environ.get("LM_FUZZ_N_GPU", 1000))
    bad = [r for r in (vm_fuzz.run_all(s, orc, ob, ctx, device=True) for s in range(n)) if r]
    assert not bad, f"{len(bad)} of {n} programs disagree:\n" + "\n".join(bad[:10])
