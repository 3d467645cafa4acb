"""ctypes binding of include/leanmultisig.h (one-to-one), plus a thin Context helper that moves numpy arrays.

All field data are numpy uint32 arrays in Montgomery form.  Host EF arrays have a trailing axis of 5.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# LM_LIB selects another build of the same sources (the release-fence variant libleanmultisig_hip_fences.so, tests/test_fences_gpu.py)
LIB_PATH = os.environ.get("LM_LIB") or os.path.join(HERE, "libleanmultisig_hip.so")

P = 0x7F000001
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class LmError(RuntimeError):
    pass


_SIGS = {
    "lm_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "lm_ctx_destroy": (None, [vp]),
    "lm_last_error": (C.c_char_p, []),
    "lm_sync": (C.c_int, [vp]),
    "lm_bind_thread": (C.c_int, [vp]),
    "lm_ctx_stream": (vp, [vp]),
    "lm_profile_select": (C.c_int, [vp, C.c_char_p]),
    "lm_profile_names": (C.c_uint64, [vp, vp, C.c_uint64]),
    "lm_wait_log": (C.c_int, [vp, C.c_int]),
    "lm_wait_log_read": (C.c_uint64, [vp, vp, C.c_uint64]),
    "lm_soft_fallbacks": (C.c_uint32, [vp]),
    "lm_profile_read": (C.c_int, [vp, C.c_char_p, u64p, C.POINTER(C.c_double)]),
    "lm_profile_busy_ms": (C.c_double, [vp]),
    "lm_profile_read_bytes": (C.c_uint64, [vp, C.c_char_p]),
    "lm_malloc": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "lm_free": (C.c_int, [vp, vp]),
    "lm_upload": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_upload_async": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_download": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_memset_zero": (C.c_int, [vp, vp, C.c_uint64]),
    "lm_fill_columns": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint64]),
    "lm_ef_aos_to_soa": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_ef_soa_to_aos": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_poseidon16_permute": (C.c_int, [vp, vp, C.c_uint64]),
    "lm_poseidon16_compress": (C.c_int, [vp, vp, C.c_uint64]),
    "lm_poseidon16_permute_quad": (C.c_int, [vp, vp, C.c_uint64, C.c_int]),
    "lm_poseidon_trace": (C.c_int, [vp, vp, C.c_uint64]),
    "lm_extension_op_trace": (C.c_int, [vp, vp, C.c_uint64, vp, vp, C.c_uint64]),
    "lm_poseidon_trace_outputs_from_memory": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64]),
    "lm_poseidon_table_from_calls": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, vp]),
    "lm_extension_table_from_rows": (C.c_int, [vp, vp, C.c_uint64, vp]),
    "lm_execution_table_trace": (C.c_int, [vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, vp]),
    "lm_commit": (C.c_int, [vp, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(vp), vp]),
    "lm_tree_free": (None, [vp, vp]),
    "lm_tree_log_height": (C.c_uint32, [vp]),
    "lm_tree_leaf_words": (C.c_uint32, [vp]),
    "lm_tree_open": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp]),
    "lm_tree_open_begin": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(vp)]),
    "lm_tree_open_end": (C.c_int, [vp, vp, vp, vp]),
    "lm_mle_eval_points": (C.c_int, [vp, vp, C.c_int, C.c_uint32, C.c_uint32, vp, vp]),
    "lm_results_defer_begin": (C.c_int, [vp]),
    "lm_results_defer_end": (C.c_int, [vp]),
    "lm_tree_download_matrix": (C.c_int, [vp, vp, vp]),
    "lm_tree_download_digests": (C.c_int, [vp, vp, vp]),
    "lm_mle_eval_cols": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, vp, vp]),
    "lm_copy_d2d": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lm_mle_eval": (C.c_int, [vp, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp]),
    "lm_weights_accumulate": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint64, vp]),
    "lm_prod_round2": (C.c_int, [vp, vp, C.c_int, vp, C.c_uint32, vp]),
    "lm_fold2_round": (C.c_int, [vp, vp, C.c_int, vp, C.c_uint32, vp, vp, vp, vp, C.c_int, vp]),
    "lm_fold_round": (C.c_int, [vp, vp, C.c_int, vp, C.c_uint32, vp, vp, vp, vp]),
    "lm_access_counts": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp]),
    "lm_access_errors": (C.c_uint32, [vp, C.c_int]),
    "lm_stack_columns": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp]),
    "lm_weights_init": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint64, vp]),
    "lm_prod_round": (C.c_int, [vp, vp, C.c_int, vp, C.c_uint32, vp]),
    "lm_fold": (C.c_int, [vp, vp, C.c_int, C.c_uint32, vp, vp]),
    "lm_pow_grind": (C.c_int, [vp, vp, C.c_uint32, u32p]),
    "lm_logup_build": (C.c_int, [vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp]),
    "lm_gkr_build": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(vp)]),
    "lm_logup_build_active": (C.c_int, [vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp]),
    "lm_gkr_build_active": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint64, C.POINTER(vp)]),
    "lm_gkr_free": (None, [vp, vp]),
    "lm_gkr_top": (C.c_int, [vp, vp, vp, vp]),
    "lm_gkr_layer_begin": (C.c_int, [vp, vp, C.c_uint32, vp, vp]),
    "lm_gkr_round": (C.c_int, [vp, vp, vp, vp]),
    "lm_gkr_layer_end": (C.c_int, [vp, vp, vp, vp]),
    "lm_air_new": (C.c_int, [vp, C.c_uint32, vp, C.c_uint32, vp, vp, vp, vp, C.POINTER(vp)]),
    "lm_air_free": (None, [vp, vp]),
    "lm_air_set_active_rows": (C.c_int, [vp, C.c_uint64]),
    "lm_air_degree": (C.c_uint32, [vp]),
    "lm_air_n_evals": (C.c_uint32, [vp]),
    "lm_air_round": (C.c_int, [vp, vp, vp]),
    "lm_air_round_launch": (C.c_int, [vp, vp]),
    "lm_air_round_wait": (C.c_int, [vp, vp, vp]),
    "lm_air_bind": (C.c_int, [vp, vp, vp]),
    "lm_air_final_evals": (C.c_int, [vp, vp, vp]),
    "lm_air_final_evals_begin": (C.c_int, [vp, vp]),
    "lm_air_final_evals_end": (C.c_int, [vp, vp, vp]),
}

# include/leanmultisig_host.h
_HOST_SIGS = {
    "lmh_table_log_rows": (C.c_uint32, [C.c_uint64]),
    "lmh_pad_table": (C.c_int, [vp, C.c_uint32, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "lmh_poseidon_backend": (C.c_char_p, []),
    "lmh_poseidon16_permute": (None, [vp]),
    "lmh_poseidon16_permute_scalar": (None, [vp]),
    "lmh_prover_new": (vp, []),
    "lmh_prover_free": (None, [vp]),
    "lmh_add_base_scalars": (None, [vp, vp, C.c_uint64]),
    "lmh_observe_scalars": (None, [vp, vp, C.c_uint64]),
    "lmh_add_extension_scalars": (None, [vp, vp, C.c_uint64]),
    "lmh_duplex": (None, [vp]),
    "lmh_sample_vec": (C.c_int, [vp, C.c_uint64, vp]),
    "lmh_sample_in_range": (C.c_int, [vp, C.c_uint32, C.c_uint64, vp]),
    "lmh_add_sumcheck_polynomial": (None, [vp, vp, C.c_uint32, vp]),
    "lmh_pow_grinding": (C.c_int, [vp, vp, C.c_uint32]),
    "lmh_challenger_state": (None, [vp, vp]),
    "lmh_proof_words": (C.c_uint64, [vp]),
    "lmh_proof_pruned_words": (C.c_uint64, [vp]),
    "lmh_proof_pruned_copy": (None, [vp, vp]),
    "lmh_proof_size_fe": (C.c_uint64, [vp]),
    "lmh_proof_n_batches": (C.c_uint32, [vp]),
    "lmh_proof_batch_sizes": (None, [vp, vp]),
    "lmh_proof_copy": (None, [vp, vp]),
    "lmh_prover_load_raw": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint32]),
    "lmh_proof_postcard_size": (C.c_uint64, [vp]),
    "lmh_proof_postcard": (None, [vp, vp]),
    "lmh_proof_compressed_size": (C.c_uint64, [vp]),
    "lmh_proof_compressed": (None, [vp, vp]),
    "lmh_lz4_compress_bound": (C.c_uint64, [C.c_uint64]),
    "lmh_lz4_compress_prepend_size": (C.c_uint64, [vp, C.c_uint64, vp]),
    "lmh_lz4_decompress_size_prepended": (C.c_int64, [vp, C.c_uint64, vp, C.c_uint64]),
    "lmh_proof_from_postcard": (vp, [vp, C.c_uint64]),
    "lmh_proof_decompress": (vp, [vp, C.c_uint64]),
    "lmh_proof_free": (None, [vp]),
    "lmh_proof_decoded_size_fe": (C.c_uint64, [vp]),
    "lmh_proof_decoded_pruned_words": (C.c_uint64, [vp, vp]),
    "lmh_verify_execution": (C.c_int, [vp, vp, vp]),
    "lmh_verify_execution_bytes": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp]),
    "lmh_verify_execution_prover": (C.c_int, [vp, vp, vp]),
    "lmh_verify_execution_raw": (C.c_int, [vp, vp, vp, vp]),
    "lmh_raw_proof_transcript": (vp, [vp, vp]),
    "lmh_raw_proof_whir_claim": (vp, [vp]),
    "lmh_raw_proof_statement_claim": (vp, [vp]),
    "lmh_raw_proof_free": (None, [vp]),
    "lmh_whir_commit": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.POINTER(vp)]),
    "lmh_witness_free": (None, [vp, vp]),
    "lmh_witness_root": (None, [vp, vp]),
    "lmh_default_whir_builder": (None, [C.c_uint32, C.c_int, vp]),
    "lmh_whir_config_new": (C.c_int, [vp, C.c_uint32, vp]),
    "lmh_prove_gkr_quotient_active": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint64, vp, vp, vp]),
    "lmh_prove_gkr_quotient": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, vp]),
    "lmh_prove_batched_air_sumcheck": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, vp]),
    "lmh_stacked_n_vars": (C.c_uint32, [vp]),
    "lmh_prove_execution": (C.c_int, [vp, vp, vp, vp]),
    "lmh_prover_stage_times": (C.c_int, [vp, vp]),
    "lmh_whir_prove": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp]),
    # leanVM (leanmultisig_amd/vm.py)
    "lmh_bytecode_new": (vp, [vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, vp, C.c_uint64, C.c_uint32]),
    "lmh_bytecode_free": (None, [vp]),
    "lmh_bytecode_hash": (None, [vp, vp]),
    "lmh_bytecode_log_size": (C.c_uint32, [vp]),
    "lmh_bytecode_ending_pc": (C.c_uint32, [vp]),
    "lmh_bytecode_multilinear": (vp, [vp]),
    "lmh_execute_bytecode": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(vp)]),
    "lmh_execute_bytecode_device": (C.c_int, [vp, vp, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(vp)]),
    "lmh_execution_on_device": (C.c_int, [vp]),
    "lmh_execution_free": (None, [vp]),
    "lmh_execution_view": (None, [vp, vp]),
    "lmh_get_execution_trace": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "lmh_vm_trace_view": (vp, [vp]),
    "lmh_vm_trace_free": (None, [vp, vp]),
    "lmh_prove_execution_vm": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp]),
    "lmh_prove_execution_vm_info": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp]),
    "lmh_execution_info": (None, [vp, vp]),
    "lmh_bytecode_set_hint_names": (C.c_int, [vp, vp, C.c_uint32]),
    "lmh_bytecode_hint_name_id": (C.c_int, [vp, C.c_char_p]),
    "lmh_bytecode_n_hint_names": (C.c_uint32, [vp]),
    # aggregate_type_1 (leanmultisig_amd/vm.py)
    "lmh_aggregate_type_1_witness": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint32, C.POINTER(vp)]),
    "lmh_type1_witness_vm": (vp, [vp]),
    "lmh_type1_witness_public_input": (vp, [vp]),
    "lmh_type1_witness_input_data": (vp, [vp, vp]),
    "lmh_type1_witness_n_sigs": (C.c_uint64, [vp]),
    "lmh_type1_witness_pubkeys": (vp, [vp]),
    "lmh_type1_witness_free": (None, [vp]),
    "lmh_aggregate_type_1": (C.c_int, [vp, vp, vp, vp, C.c_uint64, vp, C.c_uint32, vp, C.c_uint32, vp, vp]),
    "lmh_poseidon16_compress_many": (None, [vp, C.c_uint64, C.c_uint32]),
}

LM_MAX_WHIR_ROUNDS = 8


class WhirRound(C.Structure):
    _fields_ = [("query_pow_bits", C.c_uint32), ("folding_pow_bits", C.c_uint32), ("num_queries", C.c_uint32),
                ("ood_samples", C.c_uint32)]


class WhirConfig(C.Structure):
    """lm_whir_config"""
    _fields_ = [("num_variables", C.c_uint32), ("starting_log_inv_rate", C.c_uint32),
                ("folding_factor_first", C.c_uint32), ("folding_factor_subsequent", C.c_uint32),
                ("rs_domain_initial_reduction_factor", C.c_uint32), ("commitment_ood_samples", C.c_uint32),
                ("starting_folding_pow_bits", C.c_uint32), ("n_rounds", C.c_uint32), ("final_queries", C.c_uint32),
                ("final_query_pow_bits", C.c_uint32), ("final_sumcheck_rounds", C.c_uint32),
                ("rounds", WhirRound * LM_MAX_WHIR_ROUNDS)]

    @classmethod
    def from_dict(cls, d):
        c = cls()
        c.num_variables = d["num_variables"]
        c.starting_log_inv_rate = d["starting_log_inv_rate"]
        c.folding_factor_first = d["fold_first"]
        c.folding_factor_subsequent = d["fold_sub"]
        c.rs_domain_initial_reduction_factor = d["rs_red"]
        c.commitment_ood_samples = d["commitment_ood_samples"]
        c.starting_folding_pow_bits = d["starting_folding_pow_bits"]
        c.n_rounds = d["n_rounds"]
        c.final_queries = d["final_queries"]
        c.final_query_pow_bits = d["final_query_pow_bits"]
        c.final_sumcheck_rounds = d["final_sumcheck_rounds"]
        for i, r in enumerate(d["rounds"]):
            c.rounds[i].query_pow_bits = r["query_pow_bits"]
            c.rounds[i].folding_pow_bits = r["folding_pow_bits"]
            c.rounds[i].num_queries = r["num_queries"]
            c.rounds[i].ood_samples = r["ood_samples"]
        return c


    def to_dict(self):
        return dict(num_variables=self.num_variables, starting_log_inv_rate=self.starting_log_inv_rate,
                    fold_first=self.folding_factor_first, fold_sub=self.folding_factor_subsequent,
                    rs_red=self.rs_domain_initial_reduction_factor, commitment_ood_samples=self.commitment_ood_samples,
                    starting_folding_pow_bits=self.starting_folding_pow_bits, n_rounds=self.n_rounds,
                    final_queries=self.final_queries, final_query_pow_bits=self.final_query_pow_bits,
                    final_sumcheck_rounds=self.final_sumcheck_rounds,
                    rounds=[dict(query_pow_bits=r.query_pow_bits, folding_pow_bits=r.folding_pow_bits, num_queries=r.num_queries,
                                 ood_samples=r.ood_samples) for r in list(self.rounds)[:self.n_rounds]])

    @classmethod
    def new(cls, builder, num_variables):
        """WhirConfig::new (crates/whir/src/config.rs:186-334) through lmh_whir_config_new."""
        lib = load()
        c = cls()
        rc = lib.lmh_whir_config_new(C.byref(builder), num_variables, C.byref(c))
        if rc != 0:
            raise LmError(f"lmh_whir_config_new -> {rc}: {lib.lm_last_error().decode()}")
        return c


SOUNDNESS = {"UniqueDecoding": 0, "JohnsonBound": 1, "CapacityBound": 2}


class WhirBuilder(C.Structure):
    """lm_whir_builder (WhirConfigBuilder, crates/whir/src/config.rs:82-102)"""
    _fields_ = [("starting_log_inv_rate", C.c_uint32), ("max_num_variables_to_send_coeffs", C.c_uint32),
                ("rs_domain_initial_reduction_factor", C.c_uint32), ("folding_factor_first", C.c_uint32),
                ("folding_factor_subsequent", C.c_uint32), ("soundness_type", C.c_uint32), ("security_level", C.c_uint32),
                ("pow_bits", C.c_uint32)]

    @classmethod
    def default(cls, log_inv_rate, prox_gaps_conjecture=False, **over):
        """default_whir_config (crates/lean_prover/src/lib.rs:22-50); keyword arguments override single fields."""
        b = cls()
        load().lmh_default_whir_builder(log_inv_rate, int(prox_gaps_conjecture), C.byref(b))
        for k, v in over.items():
            setattr(b, k, v)
        return b


def host_poseidon_backend() -> str:
    """which host permutation the transcript uses: "avx512-ifma" or "scalar" (lmh_poseidon_backend)"""
    return load().lmh_poseidon_backend().decode()


def host_poseidon16_permute(state, scalar=False):
    """one Poseidon1-16 permutation on the host (Montgomery words), with the transcript's backend or the scalar code"""
    lib = load()
    s = np.ascontiguousarray(state, dtype=np.uint32).copy()
    assert s.shape == (16,)
    (lib.lmh_poseidon16_permute_scalar if scalar else lib.lmh_poseidon16_permute)(s.ctypes.data_as(C.c_void_p))
    return s


def lz4_compress(data: bytes) -> bytes:
    lib = load()
    src = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(lib.lmh_lz4_compress_bound(src.size), dtype=np.uint8)
    n = lib.lmh_lz4_compress_prepend_size(src.ctypes.data_as(C.c_void_p), src.size, out.ctypes.data_as(C.c_void_p))
    return out[:n].tobytes()


def lz4_decompress(data: bytes):
    """lz4_flex::decompress_size_prepended; None on malformed input"""
    lib = load()
    src = np.frombuffer(data, dtype=np.uint8)
    size = lib.lmh_lz4_decompress_size_prepended(src.ctypes.data_as(C.c_void_p), src.size, None, 0)
    if size < 0 or size > (1 << 30):
        return None
    out = np.empty(max(size, 1), dtype=np.uint8)
    n = lib.lmh_lz4_decompress_size_prepended(src.ctypes.data_as(C.c_void_p), src.size, out.ctypes.data_as(C.c_void_p), size)
    return out[:n].tobytes() if n == size else None


class DecodedProof:
    """lmh_proof: a Proof<F> decoded from the reference's bytes (postcard, optionally lz4 size-prepended)."""

    def __init__(self, data: bytes, compressed=False):
        self.lib = load()
        src = np.frombuffer(data, dtype=np.uint8)
        f = self.lib.lmh_proof_decompress if compressed else self.lib.lmh_proof_from_postcard
        self.h = f(src.ctypes.data_as(C.c_void_p), src.size)
        if not self.h:
            raise LmError("proof decode: " + self.lib.lm_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.lmh_proof_free(self.h)
            self.h = None

    def pruned_words(self):
        out = np.empty(self.lib.lmh_proof_decoded_pruned_words(self.h, None), dtype=np.uint32)
        self.lib.lmh_proof_decoded_pruned_words(self.h, _ptr(out))
        return out

    def size_fe(self):
        return int(self.lib.lmh_proof_decoded_size_fe(self.h))


class VerifyInstance(C.Structure):
    """lm_verify_instance: what the reference's `Bytecode` + public input give its verifier"""
    _fields_ = [("log_bytecode", C.c_uint32), ("ending_pc", C.c_uint32), ("n_public_input", C.c_uint32), ("reserved", C.c_uint32),
                ("public_input", C.c_void_p), ("bytecode_hash", C.c_void_p), ("bytecode", C.c_void_p)]


def verify_execution(w, proof, builder=None, compressed=False):
    """lmh_verify_execution*: the library's verifier.  w: witness dict (log_bytecode, ending_pc, public_input, bytecode_hash,
    bytecode); proof: bytes (postcard, or lz4-framed with compressed=True), a DecodedProof or a Prover.
    Returns (accepted, message)."""
    lib = load()
    pi = np.ascontiguousarray(w["public_input"], dtype=np.uint32)
    bh = np.ascontiguousarray(w["bytecode_hash"], dtype=np.uint32)
    bc = np.ascontiguousarray(w["bytecode"], dtype=np.uint32).reshape(-1)
    assert bc.size == 16 << w["log_bytecode"]
    inst = VerifyInstance(w["log_bytecode"], w["ending_pc"], pi.size, 0, pi.ctypes.data, bh.ctypes.data, bc.ctypes.data)
    b = C.byref(builder) if builder is not None else None
    if isinstance(proof, (bytes, bytearray)):
        src = np.frombuffer(proof, dtype=np.uint8)
        rc = lib.lmh_verify_execution_bytes(C.byref(inst), src.ctypes.data_as(C.c_void_p), src.size, int(compressed), b)
    elif isinstance(proof, DecodedProof):
        rc = lib.lmh_verify_execution(C.byref(inst), proof.h, b)
    else:
        rc = lib.lmh_verify_execution_prover(C.byref(inst), proof.h, b)
    return rc == 0, ("" if rc == 0 else lib.lm_last_error().decode())


class WhirOpeningClaim(C.Structure):
    """lm_whir_opening_claim: the arguments and expected results of the recursion program's whir_open"""
    _fields_ = [("transcript_offset", C.c_uint64), ("challenger_state", C.c_uint32 * 16), ("num_variables", C.c_uint32),
                ("log_inv_rate", C.c_uint32), ("n_ood", C.c_uint32), ("n_statement_values", C.c_uint32), ("root", C.c_uint32 * 8),
                ("ood_points", C.c_uint32 * 20), ("ood_answers", C.c_uint32 * 20), ("combination_gen", C.c_uint32 * 5),
                ("statement_sum", C.c_uint32 * 5), ("statement_weights", C.c_uint32 * 5), ("folding_randomness", C.c_uint32 * 160)]


class PcsStatementClaim(C.Structure):
    """lm_pcs_statement_claim: the points of the PCS statement and where its values lie in the raw transcript"""
    _fields_ = [("log_rows", C.c_uint32 * 3), ("log_memory", C.c_uint32), ("log_bytecode", C.c_uint32), ("gkr_n_vars", C.c_uint32),
                ("n_max", C.c_uint32), ("table_order", C.c_uint32 * 3), ("ending_pc", C.c_uint32), ("log_public_memory", C.c_uint32),
                ("gkr_point", C.c_uint32 * 160), ("air_point", C.c_uint32 * 160), ("pm_point", C.c_uint32 * 40),
                ("off_value_memory_acc", C.c_uint64), ("off_value_memory", C.c_uint64), ("off_value_bytecode_acc", C.c_uint64),
                ("off_inner_evals", C.c_uint64 * 3), ("n_logup_values", C.c_uint32 * 3), ("logup_col", (C.c_uint32 * 40) * 3),
                ("logup_off", (C.c_uint64 * 40) * 3), ("air_offset", C.c_uint64), ("air_challenger_state", C.c_uint32 * 16), ("air_degree", C.c_uint32),
                ("reserved2", C.c_uint32), ("logup_c", C.c_uint32 * 5), ("off_bus_selector", C.c_uint64 * 3), ("off_bus_data", C.c_uint64 * 3),
                ("air_constraint_evals", (C.c_uint32 * 5) * 3), ("bytecode_hash_domsep", C.c_uint32 * 8), ("bytecode_value", C.c_uint32 * 5),
                ("reserved3", C.c_uint32)]


def verify_execution_raw(w, prover, builder=None, with_statement=False):
    """lmh_verify_execution_raw: verify the proof `prover` holds and return (raw transcript words, WhirOpeningClaim) — the
    RawProof::transcript the recursion program reads and what its PCS opening was asked to prove.  Raises LmError on rejection."""
    lib = load()
    pi = np.ascontiguousarray(w["public_input"], dtype=np.uint32)
    bh = np.ascontiguousarray(w["bytecode_hash"], dtype=np.uint32)
    bc = np.ascontiguousarray(w["bytecode"], dtype=np.uint32).reshape(-1)
    inst = VerifyInstance(w["log_bytecode"], w["ending_pc"], pi.size, 0, pi.ctypes.data, bh.ctypes.data, bc.ctypes.data)
    out = C.c_void_p()
    rc = lib.lmh_verify_execution_raw(C.byref(inst), prover.h, C.byref(builder) if builder is not None else None, C.byref(out))
    if rc != 0:
        raise LmError(lib.lm_last_error().decode())
    n = C.c_uint64()
    ptr = lib.lmh_raw_proof_transcript(out.value, C.byref(n))
    raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(int(n.value),)).copy()
    claim = WhirOpeningClaim.from_buffer_copy(C.string_at(lib.lmh_raw_proof_whir_claim(out.value), C.sizeof(WhirOpeningClaim)))
    stmt = PcsStatementClaim.from_buffer_copy(C.string_at(lib.lmh_raw_proof_statement_claim(out.value), C.sizeof(PcsStatementClaim)))
    lib.lmh_raw_proof_free(out.value)
    return (raw, claim, stmt) if with_statement else (raw, claim)


class SparseStatement(C.Structure):
    """lm_sparse_statement"""
    _fields_ = [("point_len", C.c_uint32), ("is_next", C.c_uint32), ("n_values", C.c_uint32), ("reserved", C.c_uint32),
                ("point_offset", C.c_uint64), ("values_offset", C.c_uint64)]


LM_LOGUP_MAX_DATA = 13


class LogupSection(C.Structure):
    """lm_logup_section"""
    _fields_ = [("out_offset", C.c_uint64), ("log_len", C.c_uint32), ("num_mode", C.c_uint32), ("d_num_col", vp),
                ("den_sign", C.c_int32), ("domsep", C.c_uint32), ("n_data", C.c_uint32), ("reserved", C.c_uint32),
                ("d_data", vp * LM_LOGUP_MAX_DATA), ("stride", C.c_uint32 * LM_LOGUP_MAX_DATA),
                ("add", C.c_uint32 * LM_LOGUP_MAX_DATA)]


class AirTable(C.Structure):
    """lm_air_table"""
    _fields_ = [("table", C.c_uint32), ("log_rows", C.c_uint32), ("d_cols", vp), ("eq_point", vp), ("sum", C.c_uint32 * 5),
                ("non_padded_n_rows", C.c_uint32)]


AIR_N_COLUMNS = {0: 20, 1: 29, 2: 109}
AIR_N_SHIFT = {0: 2, 1: 13, 2: 0}
AIR_DEGREE = {0: 5, 1: 6, 2: 10}


class VmTable(C.Structure):
    """lm_vm_table"""
    _fields_ = [("log_rows", C.c_uint32), ("non_padded_n_rows", C.c_uint32), ("d_cols", vp)]


class ExecutionTrace(C.Structure):
    """lm_execution_trace"""
    _fields_ = [("log_inv_rate", C.c_uint32), ("log_memory", C.c_uint32), ("log_bytecode", C.c_uint32), ("ending_pc", C.c_uint32),
                ("public_memory_size", C.c_uint32), ("n_public_input", C.c_uint32), ("public_input", vp), ("bytecode_hash", vp),
                ("d_bytecode", vp), ("d_bytecode_acc", vp), ("d_memory", vp), ("d_memory_acc", vp), ("tables", VmTable * 3),
                ("d_stacked", vp)]


class WeightItem(C.Structure):
    """lm_weight_item"""
    _fields_ = [("offset", C.c_uint64), ("inner_n", C.c_uint32), ("is_next", C.c_uint32), ("point_offset", C.c_uint64)]


_lib = None


def load():
    """Load the HIP library.  Raises LmError if it has not been built — there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LmError(f"{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first; "
                      "leanmultisig_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for sigs in (_SIGS, _HOST_SIGS):
        for name, (res, args) in sigs.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(vp)


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a


class DeviceBuffer:
    """n_words of HBM owned by a Context."""

    def __init__(self, ctx, n_words):
        self.ctx = ctx
        self.n_words = int(n_words)
        p = vp()
        ctx._check(ctx.lib.lm_malloc(ctx.h, self.n_words, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = _u32(arr).reshape(-1)
        assert arr.size <= self.n_words
        self.ctx._check(self.ctx.lib.lm_upload(self.ctx.h, self.ptr, _ptr(arr), arr.size))
        return self

    def download(self, n_words=None, offset=0):
        n = self.n_words - offset if n_words is None else int(n_words)
        out = np.empty(n, dtype=np.uint32)
        self.ctx._check(self.ctx.lib.lm_download(self.ctx.h, _ptr(out), self.ptr + 4 * offset, n))
        return out

    def free(self):
        if self.ptr and self.ctx.h:  # (a closed context has already released every pooled block)
            self.ctx.lib.lm_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Tree:
    def __init__(self, ctx, handle, root):
        self.ctx, self.h, self.root = ctx, handle, root
        self.log_height = ctx.lib.lm_tree_log_height(handle)
        self.leaf_words = ctx.lib.lm_tree_leaf_words(handle)

    def open(self, indices):
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        n = idx.size
        leaves = np.empty((n, self.leaf_words), dtype=np.uint32)
        sib = np.empty((n, self.log_height, 8), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.lm_tree_open(self.ctx.h, self.h, _ptr(idx), n, _ptr(leaves), _ptr(sib)))
        return leaves, sib

    def open_begin(self, indices):
        """lm_tree_open_begin: the opening kernel is enqueued; the returned closure waits for it and returns (leaves, siblings)"""
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        n = idx.size
        h = vp()
        self.ctx._check(self.ctx.lib.lm_tree_open_begin(self.ctx.h, self.h, _ptr(idx), n, C.byref(h)))

        def end():
            leaves = np.empty((n, self.leaf_words), dtype=np.uint32)
            sib = np.empty((n, self.log_height, 8), dtype=np.uint32)
            self.ctx._check(self.ctx.lib.lm_tree_open_end(self.ctx.h, h, _ptr(leaves), _ptr(sib)))
            return leaves, sib
        return end

    def matrix(self):
        out = np.empty((1 << self.log_height, self.leaf_words), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.lm_tree_download_matrix(self.ctx.h, self.h, _ptr(out)))
        return out

    def digests(self):
        out = np.empty(((2 << self.log_height) - 1, 8), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.lm_tree_download_digests(self.ctx.h, self.h, _ptr(out)))
        return out

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.lm_tree_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One lm_ctx (one GPU, one stream)."""

    def __init__(self, device=0):
        self.lib = load()
        h = vp()
        rc = self.lib.lm_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise LmError(f"lm_ctx_create({device}) -> {rc}: {self.lib.lm_last_error().decode()}")
        self.h = h.value

    def _check(self, rc):
        if rc != 0:
            raise LmError(f"leanmultisig error {rc}: {self.lib.lm_last_error().decode()}")

    def close(self):
        if self.h:
            self.lib.lm_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        self._check(self.lib.lm_sync(self.h))

    @property
    def stream(self):
        return self.lib.lm_ctx_stream(self.h)

    def profile_select(self, kernel_name):
        self._check(self.lib.lm_profile_select(self.h, kernel_name.encode() if kernel_name else None))

    def profile_names(self):
        n = self.lib.lm_profile_names(self.h, None, 0)
        buf = C.create_string_buffer(n)
        self.lib.lm_profile_names(self.h, buf, n)
        return sorted(set(x for x in buf.value.decode().split("\n") if x))

    def profile_read(self, kernel_name):
        """-> (n_launches, total_ms) of the selected kernel since the last read"""
        n, ms = C.c_uint64(0), C.c_double(0.0)
        self._check(self.lib.lm_profile_read(self.h, kernel_name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_busy_ms(self):
        """of the launches the last profile_read consumed: the time during which at least one of them ran (streams overlap)"""
        return float(self.lib.lm_profile_busy_ms(self.h))

    def wait_log(self, on=True):
        """start (and clear) / stop the log of host <-> device exchanges (lm_wait_log)"""
        self._check(self.lib.lm_wait_log(self.h, 1 if on else 0))

    def wait_log_read(self):
        """-> float32 array: microseconds the prover thread waited in each exchange since wait_log()"""
        n = int(self.lib.lm_wait_log_read(self.h, None, 0))
        out = np.zeros(n, dtype=np.float32)
        if n:
            self.lib.lm_wait_log_read(self.h, out.ctypes.data, n)
        return out

    def soft_fallbacks(self):
        """GKR layers this context re-ran with one launch per exchange (a resident kernel never got its slots): lm_soft_fallbacks"""
        return int(self.lib.lm_soft_fallbacks(self.h))

    # ---- memory -------------------------------------------------------------------------------------
    def alloc(self, n_words):
        return DeviceBuffer(self, n_words)

    def to_device(self, arr):
        arr = _u32(arr).reshape(-1)
        return DeviceBuffer(self, max(arr.size, 1)).upload(arr)

    def ef_to_device_soa(self, ef_aos):
        """host (n,5) EF array -> device SoA buffer of 5n words"""
        a = _u32(ef_aos).reshape(-1, 5)
        return self.to_device(np.ascontiguousarray(a.T))

    # ---- ops ----------------------------------------------------------------------------------------
    def poseidon16(self, states, compress=False, quad=False):
        """quad=True: the 4-lane cooperative kernel of the proof-of-work search (lm_poseidon16_permute_quad)"""
        st = _u32(states).reshape(-1, 16)
        if st.shape[0] == 0:
            return st.copy()
        buf = self.to_device(st)
        if quad:
            self._check(self.lib.lm_poseidon16_permute_quad(self.h, buf.ptr, st.shape[0], int(compress)))
        else:
            fn = self.lib.lm_poseidon16_compress if compress else self.lib.lm_poseidon16_permute
            self._check(fn(self.h, buf.ptr, st.shape[0]))
        return buf.download().reshape(-1, 16)

    def poseidon_trace(self, col_bufs, n_rows):
        ptrs = np.array([c.ptr for c in col_bufs], dtype=np.uint64)
        assert ptrs.size == 109
        self._check(self.lib.lm_poseidon_trace(self.h, _ptr(ptrs), int(n_rows)))

    def poseidon_trace_outputs_from_memory(self, col_bufs, n_rows, d_memory, memory_len):
        ptrs = np.array([c.ptr for c in col_bufs], dtype=np.uint64)
        assert ptrs.size == 109
        self._check(self.lib.lm_poseidon_trace_outputs_from_memory(self.h, _ptr(ptrs), int(n_rows), d_memory.ptr, int(memory_len)))

    def execution_table_trace(self, d_pcs, d_fps, n_cycles, d_bytecode, bytecode_rows, d_memory, memory_len, col_bufs):
        """get_execution_trace's main loop: 24 execution-table columns from the (pc, fp) log (canonical integers)."""
        dev = lambda b: b.ptr if hasattr(b, "ptr") else int(b)  # noqa: E731
        ptrs = np.array([dev(c) for c in col_bufs], dtype=np.uint64)
        assert ptrs.size == 24
        self._check(self.lib.lm_execution_table_trace(self.h, dev(d_pcs), dev(d_fps), int(n_cycles), dev(d_bytecode), int(bytecode_rows),
                                                      dev(d_memory), int(memory_len), _ptr(ptrs)))

    def extension_op_trace(self, d_memory, memory_len, d_idx_a, va_bufs, n_rows):
        """fill_trace_extension_op: the 5 value_a columns gathered from memory (all arguments device buffers / pointers)."""
        dev = lambda b: b.ptr if hasattr(b, "ptr") else int(b)  # noqa: E731
        ptrs = np.array([dev(c) for c in va_bufs], dtype=np.uint64)
        assert ptrs.size == 5
        self._check(self.lib.lm_extension_op_trace(self.h, dev(d_memory), int(memory_len), dev(d_idx_a), _ptr(ptrs), int(n_rows)))

    def commit(self, d_evals, is_ext, n_vars, folding_factor, log_inv_rate, actual_len=None):
        if actual_len is None:
            actual_len = 1 << n_vars
        t = vp()
        root = np.empty(8, dtype=np.uint32)
        ptr = d_evals.ptr if isinstance(d_evals, DeviceBuffer) else int(d_evals)
        self._check(self.lib.lm_commit(self.h, ptr, int(bool(is_ext)), n_vars, folding_factor, log_inv_rate,
                                       int(actual_len), C.byref(t), _ptr(root)))
        return Tree(self, t.value, root)

    def mle_eval(self, d_evals, is_ext, n_vars, point, n_polys=1, stride_words=None):
        if stride_words is None:
            stride_words = (5 if is_ext else 1) << n_vars
        pt = _u32(point).reshape(-1)
        assert pt.size == n_vars * 5
        out = np.empty((n_polys, 5), dtype=np.uint32)
        ptr = d_evals.ptr if isinstance(d_evals, DeviceBuffer) else int(d_evals)
        self._check(self.lib.lm_mle_eval(self.h, ptr, int(bool(is_ext)), n_vars, n_polys, int(stride_words),
                                         _ptr(pt) if pt.size else None, _ptr(out)))
        return out


    def mle_eval_points(self, d_evals, is_ext, n_vars, points):
        """one polynomial at several points (points: n_points x n_vars x 5) -> n_points x 5"""
        pts = _u32(points).reshape(-1)
        n_points = pts.size // max(1, n_vars * 5) if n_vars else int(np.asarray(points).shape[0])
        assert n_vars == 0 or pts.size == n_points * n_vars * 5
        out = np.empty((n_points, 5), dtype=np.uint32)
        ptr = d_evals.ptr if isinstance(d_evals, DeviceBuffer) else int(d_evals)
        self._check(self.lib.lm_mle_eval_points(self.h, ptr, int(bool(is_ext)), n_vars, n_points, _ptr(pts) if pts.size else None, _ptr(out)))
        return out

    def mle_eval_deferred(self, jobs):
        """jobs: list of (d_evals, is_ext, n_vars, point) evaluated between lm_results_defer_begin / _end -> list of 5-word results"""
        outs = [np.zeros((1, 5), dtype=np.uint32) for _ in jobs]
        keep = []
        self._check(self.lib.lm_results_defer_begin(self.h))
        try:
            for (d_evals, is_ext, n_vars, point), out in zip(jobs, outs):
                pt = _u32(point).reshape(-1)
                keep.append(pt)
                ptr = d_evals.ptr if isinstance(d_evals, DeviceBuffer) else int(d_evals)
                self._check(self.lib.lm_mle_eval(self.h, ptr, int(bool(is_ext)), n_vars, 1, (5 if is_ext else 1) << n_vars,
                                                 _ptr(pt) if pt.size else None, _ptr(out)))
        finally:
            self._check(self.lib.lm_results_defer_end(self.h))
        return [o[0] for o in outs]

    def access_counts(self, length, jobs):
        """jobs: list of (DeviceBuffer index column, n_rows, n_values) -> DeviceBuffer of `length` field elements"""
        out = self.alloc(length)
        n = len(jobs)
        cols = (C.c_void_p * max(n, 1))(*[b.ptr for b, _, _ in jobs])
        rows = (C.c_uint64 * max(n, 1))(*[r for _, r, _ in jobs])
        nv = (C.c_uint32 * max(n, 1))(*[v for _, _, v in jobs])
        self._check(self.lib.lm_access_counts(self.h, out.ptr, length, n, cols, rows, nv))
        return out

    def pad_table(self, table, col_bufs, n_rows, log_rows, zero_vec_ptr, null_hash_ptr, ending_pc):
        """lmh_pad_table: padding rows [n_rows, 2^log_rows) of the committed columns (list of DeviceBuffer)"""
        ptrs = np.array([c.ptr for c in col_bufs], dtype=np.uint64)
        self._check(self.lib.lmh_pad_table(self.h, table, _ptr(ptrs), int(n_rows), int(log_rows), int(zero_vec_ptr), int(null_hash_ptr),
                                           int(ending_pc)))

    def access_errors(self, reset=True):
        """rows of access_counts jobs that pointed outside the image since the last reset (synchronises)"""
        return int(self.lib.lm_access_errors(self.h, int(reset)))

    def stack_columns(self, total_words, jobs):
        """jobs: list of (DeviceBuffer src, src_word_offset, dst_offset, n_words) sorted by dst_offset -> DeviceBuffer"""
        out = self.alloc(total_words)
        n = len(jobs)
        srcs = (C.c_void_p * max(n, 1))(*[b.ptr + 4 * so for b, so, _, _ in jobs])
        offs = (C.c_uint64 * max(n, 1))(*[d for _, _, d, _ in jobs])
        lens = (C.c_uint64 * max(n, 1))(*[w for _, _, _, w in jobs])
        self._check(self.lib.lm_stack_columns(self.h, out.ptr, total_words, n, srcs, offs, lens))
        return out

    def weights_accumulate(self, d_W, n_vars, items, points, scalars, init=False):
        """items: list of (offset, inner_n, is_next, point_offset); points (k,5); scalars (n_items,5).
        init=True: lm_weights_init (W is write-only)"""
        arr = (WeightItem * len(items))()
        for i, (off, inner, nxt, poff) in enumerate(items):
            arr[i].offset, arr[i].inner_n, arr[i].is_next, arr[i].point_offset = off, inner, int(nxt), poff
        pts = _u32(points).reshape(-1)
        sc = _u32(scalars).reshape(-1)
        fn = self.lib.lm_weights_init if init else self.lib.lm_weights_accumulate
        self._check(fn(self.h, d_W.ptr, n_vars, C.cast(arr, vp), len(items), _ptr(pts) if pts.size else None, pts.size // 5,
                       _ptr(sc)))

    def logup_build(self, sections, c, alphas_eq16, n_vars):
        """sections: list of dict(out_offset, log_len, num_mode, num_col (device ptr/None), den_sign, domsep,
        data=[(device ptr or None, stride, add), ...]).  Returns (d_nums, d_dens SoA)."""
        arr = (LogupSection * len(sections))()
        for i, s in enumerate(sections):
            a = arr[i]
            a.out_offset, a.log_len, a.num_mode = s["out_offset"], s["log_len"], s["num_mode"]
            a.d_num_col = s.get("num_col") or None
            a.den_sign, a.domsep, a.n_data = s["den_sign"], s["domsep"], len(s["data"])
            for j, (ptr, stride, add) in enumerate(s["data"]):
                a.d_data[j] = ptr or None
                a.stride[j] = stride
                a.add[j] = add
        d_nums = self.alloc(1 << n_vars)
        d_dens = self.alloc(5 << n_vars)
        cc, al = _u32(c), _u32(alphas_eq16).reshape(-1)
        self._check(self.lib.lm_logup_build(self.h, C.cast(arr, vp), len(sections), _ptr(cc), _ptr(al), n_vars, d_nums.ptr,
                                            d_dens.ptr))
        return d_nums, d_dens

    def prod_round2(self, d_f, f_is_ext, d_W, n_vars):
        """the eight sums of a two-round pass (40 words: p00, p01, p10, q0, q1, t0, t2, t3)"""
        out = np.empty(40, dtype=np.uint32)
        self._check(self.lib.lm_prod_round2(self.h, d_f.ptr, int(bool(f_is_ext)), d_W.ptr, n_vars, _ptr(out)))
        return out

    def prod_round(self, d_f, f_is_ext, d_W, n_vars):
        out = np.empty(10, dtype=np.uint32)
        self._check(self.lib.lm_prod_round(self.h, d_f.ptr, int(bool(f_is_ext)), d_W.ptr, n_vars, _ptr(out)))
        return out[:5].copy(), out[5:].copy()

    def fold(self, d_in, in_is_ext, n_vars, r):
        out = self.alloc(5 << (n_vars - 1))
        r = _u32(r)
        self._check(self.lib.lm_fold(self.h, d_in.ptr, int(bool(in_is_ext)), n_vars, _ptr(r), out.ptr))
        return out

    def fold_round(self, d_f, f_is_ext, d_W, n_vars, r):
        """-> (f' SoA EF, W' SoA EF, c0, c2 of the next round)"""
        half = 1 << (n_vars - 1)
        fo, wo = self.alloc(5 * half), self.alloc(5 * half)
        out = np.zeros(10, dtype=np.uint32)
        rr = _u32(r)
        self._check(self.lib.lm_fold_round(self.h, d_f.ptr, int(f_is_ext), d_W.ptr, n_vars, _ptr(rr), fo.ptr, wo.ptr, _ptr(out)))
        return fo, wo, out[:5].copy(), out[5:].copy()

    def pow_grind(self, capacity, bits):
        cap = _u32(capacity)
        w = C.c_uint32(0)
        self._check(self.lib.lm_pow_grind(self.h, _ptr(cap), bits, C.byref(w)))
        return w.value


def make_execution_trace(ctx, w, device_counters=True, active_prefix=True):
    """Upload a witness dict (tests/synth_witness.py layout) and build the lm_execution_trace; returns (trace, keepalive).
    device_counters: leave memory_acc / bytecode_acc to the library (prove_execution.rs:90-110 on the device) instead of
    uploading the witness's own."""
    keep = []
    tr = ExecutionTrace()
    tr.log_inv_rate, tr.log_memory, tr.log_bytecode = w["log_inv_rate"], w["log_memory"], w["log_bytecode"]
    tr.ending_pc, tr.public_memory_size = w["ending_pc"], w["public_memory_size"]
    pi = _u32(w["public_input"])
    bh = _u32(w["bytecode_hash"])
    keep += [pi, bh]
    tr.n_public_input, tr.public_input, tr.bytecode_hash = pi.size, pi.ctypes.data, bh.ctypes.data
    for name, key in (("d_bytecode", "bytecode"), ("d_bytecode_acc", "bytecode_acc"), ("d_memory", "memory"), ("d_memory_acc", "memory_acc")):
        if device_counters and key.endswith("_acc"):
            setattr(tr, name, None)
            continue
        b = ctx.to_device(w[key])
        keep.append(b)
        setattr(tr, name, b.ptr)
    for t in range(3):
        bufs = [ctx.to_device(c) for c in w["tables"][t]]
        ptrs = np.array([b.ptr for b in bufs], dtype=np.uint64)
        keep += [bufs, ptrs]
        tr.tables[t].log_rows = w["log_rows"][t]
        tr.tables[t].d_cols = ptrs.ctypes.data
        if active_prefix:
            # TableTrace::non_padded_n_rows: the VM knows it; here it is read off the generator's table (the rows behind it
            # all equal the last row, the table's padding row)
            tab = np.asarray(w["tables"][t])[:AIR_N_COLUMNS[t]]
            same = np.all(tab == tab[:, -1:], axis=0)
            n_pad = int(same.size if same.all() else np.argmin(same[::-1]))
            tr.tables[t].non_padded_n_rows = max(1, same.size - n_pad)
    return tr, keep


class Prover:
    """lmh_prover: ProverState of the reference (transcript + challenger) driving the device."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = ctx.lib
        self.h = self.lib.lmh_prover_new()

    def close(self):
        if self.h:
            self.lib.lmh_prover_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_base_scalars(self, s):
        s = _u32(s).reshape(-1)
        self.lib.lmh_add_base_scalars(self.h, _ptr(s), s.size)

    STAGES = ("commit: stack + access counts + FFT + build merkle tree + ood evaluation", "logup fill", "prove GKR quotient", "column evaluations",
              "batched AIR sumcheck", "WHIR prove")

    def stage_times(self):
        """{stage: ms} of the last prove_execution on this prover (lmh_prover_stage_times; the reference's tracing spans)"""
        t = (C.c_double * 8)()
        self.ctx._check(self.lib.lmh_prover_stage_times(self.h, t))
        return {name: float(t[i]) for i, name in enumerate(self.STAGES)}

    def proof(self):
        n = self.lib.lmh_proof_words(self.h)
        out = np.empty(n, dtype=np.uint32)
        self.lib.lmh_proof_copy(self.h, _ptr(out))
        return out

    def proof_pruned(self):
        """Pruned proof blob (MerklePaths::prune per query set; layout in include/leanmultisig_host.h)."""
        out = np.empty(self.lib.lmh_proof_pruned_words(self.h), dtype=np.uint32)
        self.lib.lmh_proof_pruned_copy(self.h, _ptr(out))
        return out

    def proof_bytes(self, compressed=False):
        """The reference's serialised ExecutionProof: postcard(Proof<F>), optionally lz4 size-prepended."""
        if compressed:
            out = np.empty(self.lib.lmh_proof_compressed_size(self.h), dtype=np.uint8)
            self.lib.lmh_proof_compressed(self.h, out.ctypes.data_as(C.c_void_p))
        else:
            out = np.empty(self.lib.lmh_proof_postcard_size(self.h), dtype=np.uint8)
            self.lib.lmh_proof_postcard(self.h, out.ctypes.data_as(C.c_void_p))
        return out.tobytes()

    @classmethod
    def from_raw(cls, raw_words, batch_sizes, lib=None):
        """A prover object holding a proof produced elsewhere (un-pruned blob + openings per batch): host only, no context."""
        self = cls.__new__(cls)
        self.ctx = None
        self.lib = lib or load()
        self.h = self.lib.lmh_prover_new()
        raw, bs = _u32(raw_words).reshape(-1), _u32(batch_sizes).reshape(-1)
        rc = self.lib.lmh_prover_load_raw(self.h, _ptr(raw), raw.size, _ptr(bs), bs.size)
        if rc != 0:
            raise LmError(f"lmh_prover_load_raw -> {rc}")
        return self

    def proof_size_fe(self):
        """Proof::proof_size_fe of the reference (field elements of the pruned proof)."""
        return int(self.lib.lmh_proof_size_fe(self.h))

    def batch_sizes(self):
        out = np.empty(self.lib.lmh_proof_n_batches(self.h), dtype=np.uint32)
        if out.size:
            self.lib.lmh_proof_batch_sizes(self.h, _ptr(out))
        return out

    def state(self):
        out = np.empty(16, dtype=np.uint32)
        self.lib.lmh_challenger_state(self.h, _ptr(out))
        return out

    def prove_gkr_quotient(self, d_nums, d_dens, n_vars, active_len=None):
        """active_len: entries beyond it are the neutral pair (0, 1) and are never read (lm_gkr_build_active)"""
        q = np.empty(5, dtype=np.uint32)
        pt = np.empty((n_vars, 5), dtype=np.uint32)
        cl = np.empty((2, 5), dtype=np.uint32)
        if active_len is None:
            self.ctx._check(self.lib.lmh_prove_gkr_quotient(self.ctx.h, self.h, d_nums.ptr, d_dens.ptr, n_vars, _ptr(q), _ptr(pt), _ptr(cl)))
        else:
            self.ctx._check(self.lib.lmh_prove_gkr_quotient_active(self.ctx.h, self.h, d_nums.ptr, d_dens.ptr, n_vars, active_len, _ptr(q),
                                                                   _ptr(pt), _ptr(cl)))
        return q, pt, cl

    def prove_batched_air_sumcheck(self, tables, alpha, logup_eq16, bus_beta, eta):
        """tables: list of dict(table=int, log_rows=int, cols=[DeviceBuffer or device ptr per column], eq_point=(n,5),
        sum=ef5), already in the reference's order (descending height).  Returns (point (n_max,5), [col_evals per table])."""
        arr = (AirTable * len(tables))()
        keep = []
        n_max = 0
        total = 0
        for i, t in enumerate(tables):
            ptrs = np.array([c.ptr if isinstance(c, DeviceBuffer) else int(c) for c in t["cols"]], dtype=np.uint64)
            assert ptrs.size == AIR_N_COLUMNS[t["table"]]
            eqp = _u32(t["eq_point"]).reshape(-1)
            assert eqp.size == 5 * t["log_rows"]
            keep += [ptrs, eqp]
            arr[i].table, arr[i].log_rows = t["table"], t["log_rows"]
            arr[i].non_padded_n_rows = int(t.get("non_padded_n_rows", 0))
            arr[i].d_cols, arr[i].eq_point = ptrs.ctypes.data, eqp.ctypes.data
            for k in range(5):
                arr[i].sum[k] = int(t["sum"][k])
            n_max = max(n_max, t["log_rows"])
            total += AIR_N_COLUMNS[t["table"]] + AIR_N_SHIFT[t["table"]]
        al, eq16, bb, et = _u32(alpha), _u32(logup_eq16).reshape(-1), _u32(bus_beta), _u32(eta)
        assert eq16.size == 80
        pt = np.empty((n_max, 5), dtype=np.uint32)
        ev = np.empty((total, 5), dtype=np.uint32)
        self.ctx._check(self.lib.lmh_prove_batched_air_sumcheck(self.ctx.h, self.h, C.cast(arr, vp), len(tables), _ptr(al),
                                                                _ptr(eq16), _ptr(bb), _ptr(et), _ptr(pt), _ptr(ev)))
        out, k = [], 0
        for t in tables:
            n = AIR_N_COLUMNS[t["table"]] + AIR_N_SHIFT[t["table"]]
            out.append(ev[k:k + n].copy())
            k += n
        return pt, out

    def prove_execution(self, trace, cfg):
        self.ctx._check(self.lib.lmh_prove_execution(self.ctx.h, self.h, C.byref(trace), C.byref(cfg)))

    def whir_commit(self, cfg, d_poly, actual_len):
        w = vp()
        self.ctx._check(self.lib.lmh_whir_commit(self.ctx.h, self.h, C.byref(cfg), d_poly.ptr, int(actual_len), C.byref(w)))
        return w.value

    def whir_prove(self, cfg, statements, witness, d_poly):
        """statements: list of dict(point=(k,5), is_next, values=[(selector, ef5)])"""
        arr = (SparseStatement * max(len(statements), 1))()
        pts, sels, vals = [], [], []
        for i, s in enumerate(statements):
            pt = np.asarray(s["point"], dtype=np.uint32).reshape(-1, 5)
            arr[i].point_len = pt.shape[0]
            arr[i].is_next = int(bool(s.get("is_next", False)))
            arr[i].n_values = len(s["values"])
            arr[i].point_offset = sum(p.shape[0] for p in pts)
            arr[i].values_offset = len(sels)
            pts.append(pt)
            for sel, v in s["values"]:
                sels.append(sel)
                vals.append(np.asarray(v, dtype=np.uint32))
        pts = np.concatenate(pts).reshape(-1) if pts else np.zeros(0, dtype=np.uint32)
        pts = np.ascontiguousarray(pts, dtype=np.uint32)
        sels = np.array(sels or [0], dtype=np.uint64)
        vals = np.ascontiguousarray(np.array(vals or [[0] * 5], dtype=np.uint32))
        out = np.empty((cfg.num_variables, 5), dtype=np.uint32)
        self.ctx._check(self.lib.lmh_whir_prove(self.ctx.h, self.h, C.byref(cfg), C.cast(arr, vp), len(statements),
                                                _ptr(pts) if pts.size else None, pts.size // 5, _ptr(sels), _ptr(vals),
                                                len(statements) and sum(len(s["values"]) for s in statements),
                                                witness, d_poly.ptr, _ptr(out)))
        return out
