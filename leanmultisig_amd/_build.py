"""Build the HIP shared library in-tree (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libleanmultisig_hip.so")
SOURCES = ["lm_core.hip", "lm_commit.hip", "lm_whir_ops.hip", "lm_gkr.hip", "lm_air.hip", "lm_logup.hip", "host/lm_host.cpp", "host/lm_whir_config.cpp", "host/lm_wire.cpp", "host/lm_verify.cpp", "host/lm_poseidon_x86.cpp"]


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES]


def _deps():
    deps = _sources()
    for root, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".h", ".inc", ".hpp")):
                deps.append(os.path.join(root, f))
    deps.append(os.path.join(os.path.dirname(HERE), "include", "leanmultisig.h"))
    deps.append(os.path.join(os.path.dirname(HERE), "include", "leanmultisig_host.h"))
    return deps


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -> leanmultisig_amd/libleanmultisig_hip.so"""
    consts = os.path.join(CSRC, "poseidon16_consts.inc")
    gen = os.path.join(CSRC, "gen_poseidon_consts.py")
    if not os.path.exists(consts) or os.path.getmtime(gen) > os.path.getmtime(consts):
        subprocess.check_call([sys.executable, gen])
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result", *_sources(), "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
