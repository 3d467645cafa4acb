"""CPU test: the C-ABI library loads and exports every symbol include/leanmultisig.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="leanmultisig.h", prefix="lm_"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from leanmultisig_amd import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    syms = declared_symbols() + declared_symbols("leanmultisig_host.h", "lmh_")
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_covers_header():
    from leanmultisig_amd import capi
    assert sorted(capi._SIGS) == declared_symbols()
    assert sorted(capi._HOST_SIGS) == declared_symbols("leanmultisig_host.h", "lmh_")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from leanmultisig_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(capi.LmError):
        capi.load()


def test_no_undeclared_c_symbols_exported():
    """Every C-linkage function the library exports is declared in include/*.h (no stray helpers in the ABI)."""
    import re
    import subprocess
    lib = os.path.join(ROOT, "leanmultisig_amd", "libleanmultisig_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T"}
    exported = {s for s in exported if not s.startswith(("_Z", "__hip", "_init", "_fini"))}
    declared = set()
    for h in ("leanmultisig.h", "leanmultisig_host.h"):
        declared |= set(re.findall(r"\b(lmh?_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", h)).read()))
    assert exported <= declared, sorted(exported - declared)
