# Convenience targets.  The product is built by `python __graft_entry__.py` (hipcc, gfx950); nothing here is needed for it.
#
# make pin REFERENCE=/path/to/leanMultisig
#     The external parity pin, for whoever has cargo + nightly Rust (this repository's image has neither): the reference's OWN
#     verify_execution (crates/lean_prover/src/verify_execution.rs:14) checks two proofs produced by this library —
#       tests/golden/external_pin/       a device proof of the golden synthetic instance (default_whir_config)
#       tests/golden/external_pin_xmss/  a device proof of the hand-assembled aggregation program on 40 REAL XMSS signatures, VM run
#                                        with the parallel batch on the device (default_whir_config)
#     through rust_shim/ (a crate added to the reference's workspace for the duration of the test).  Needs no GPU: the fixtures are
#     committed; the shim links libleanmultisig_hip.so only for its prover-side bindings (LEANMULTISIG_HIP_DIR).
REFERENCE ?=
ROOT := $(abspath $(dir $(lastword $(MAKEFILE_LIST))))

.PHONY: pin build test
pin:
	@test -n "$(REFERENCE)" -a -f "$(REFERENCE)/Cargo.toml" || (echo "usage: make pin REFERENCE=/path/to/leanMultisig (a checkout of the reference)"; exit 2)
	rm -rf "$(REFERENCE)/rust_shim" && cp -r "$(ROOT)/rust_shim" "$(REFERENCE)/rust_shim"
	grep -q '"rust_shim"' "$(REFERENCE)/Cargo.toml" || sed -i 's/^members = \[/members = ["rust_shim", /' "$(REFERENCE)/Cargo.toml"
	cd "$(REFERENCE)" && LEANMULTISIG_HIP_DIR="$(ROOT)/leanmultisig_amd" LD_LIBRARY_PATH="$(ROOT)/leanmultisig_amd:$$LD_LIBRARY_PATH" \
	  LM_PROOF_DIR="$(ROOT)/tests/golden/external_pin" LM_PROOF_DIR_XMSS="$(ROOT)/tests/golden/external_pin_xmss" \
	  cargo test --release -p leanmultisig-hip-shim reference_verifier_accepts -- --nocapture

build:
	python $(ROOT)/__graft_entry__.py

test:
	cd $(ROOT) && python -m pytest tests/ -x -q -m "not gpu"
