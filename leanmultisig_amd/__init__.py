"""leanmultisig_amd — MI355X-native proving hot path of leanMultisig (WHIR commitment, sumchecks, logup/GKR).

The product is the HIP shared library `libleanmultisig_hip.so` (C ABI: include/leanmultisig.h).  This package is the
ctypes binding used by the tests and bench; it never falls back to a CPU implementation: if the library is missing
`load()` raises.
"""
from .capi import Context, Prover, WhirConfig, WhirBuilder, DecodedProof, lz4_compress, lz4_decompress, verify_execution, host_poseidon_backend, host_poseidon16_permute, load, LIB_PATH, LmError, make_execution_trace  # noqa: F401

__all__ = ["Context", "Prover", "WhirConfig", "WhirBuilder", "load", "LIB_PATH", "LmError"]
