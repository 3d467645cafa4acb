"""The conservative build — release fences before every published flag and every cross-workgroup ticket (-DLM_PUBLISH_FENCES=1,
csrc/lm_common.h) instead of the fence-free hand-over the product library uses on gfx950 — must give the same results: the GKR and
AIR parity tests (resident tail with its mailbox, multi-workgroup round kernels, grid sums) and the whole-function tests of
tests/test_vm_gpu.py (VM run on the device, trace, proof == oracle proof) run on libleanmultisig_hip_fences.so in a child process (the library is loaded once per process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_parity_tests_on_the_release_fence_build():
    lib = os.path.join(ROOT, "leanmultisig_amd", "libleanmultisig_hip_fences.so")
    assert os.path.exists(lib), "the variant is built by __graft_entry__.build()"
    env = dict(os.environ, LM_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gkr_gpu.py", "tests/test_air_gpu.py",
                        "tests/test_vm_gpu.py", "-k", "not child and not no_tail and not no_coop and not several_threads and not n_sigs1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
