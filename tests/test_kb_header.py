"""CPU: the product's arithmetic header (csrc/kb.h, poseidon16.h) compiled for the host and checked against 128-bit integer
arithmetic — delayed-reduction dot products at the overflow boundary, the reference's NEON regression operands
(aarch64_neon/packing.rs:44-50), the quintic product vs schoolbook, ef_inv, fold32, the Poseidon KAT (tests/cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kb_header_against_int128(tmp_path):
    exe = tmp_path / "kb_header_check"
    src = os.path.join(ROOT, "tests", "cpp", "kb_header_check.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", src, "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "kb header ok" in r.stdout, r.stdout[-3000:]
