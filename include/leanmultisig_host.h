/* leanmultisig_host.h — host-side mirror (C++ behind a C ABI) of the reference's transcript and WHIR driver.
 *
 * The reference is Rust; no Rust toolchain exists in the build image, so the layer that in the reference sits ABOVE
 * the kernels — ProverState (crates/backend/fiat-shamir/src/prover.rs), WhirConfig::commit (crates/whir/src/commit.rs)
 * and WhirConfig::prove (crates/whir/src/open.rs) — is written in C++ over the device ABI of leanmultisig.h, with the
 * same names, argument meaning and transcript order.  A Rust caller can either bind this coarse layer, or keep its own
 * Rust driver and bind the fine-grained device entry points of leanmultisig.h (INTEGRATION.md shows both).
 */
#ifndef LEANMULTISIG_HOST_H
#define LEANMULTISIG_HOST_H

#include "leanmultisig.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LM_MAX_WHIR_ROUNDS 8

/* The integers of WhirConfig (crates/whir/src/config.rs:118-134).  A Rust caller passes the ones its own WhirConfig::new
 * derived; a standalone caller obtains them from lmh_whir_config_new below. */
typedef struct {
    uint32_t num_variables;
    uint32_t starting_log_inv_rate;
    uint32_t folding_factor_first, folding_factor_subsequent; /* FoldingFactor */
    uint32_t rs_domain_initial_reduction_factor;
    uint32_t commitment_ood_samples;
    uint32_t starting_folding_pow_bits;
    uint32_t n_rounds;
    uint32_t final_queries, final_query_pow_bits, final_sumcheck_rounds;
    struct {
        uint32_t query_pow_bits, folding_pow_bits, num_queries, ood_samples;
    } rounds[LM_MAX_WHIR_ROUNDS]; /* RoundConfig, config.rs:104-116 */
} lm_whir_config;

/* WhirConfigBuilder (config.rs:82-102) and WhirConfig::new (config.rs:186-334): the schedule integers from the security
 * parameters.  lmh_default_whir_builder = default_whir_config (crates/lean_prover/src/lib.rs:22-50): 124-bit, 16 grinding
 * bits, folding 7 then 5, first RS-domain reduction 5, coefficients sent at <= 8 variables, JohnsonBound — or CapacityBound
 * when the reference is built with its `prox-gaps-conjecture` feature. */
#define LM_SOUNDNESS_UNIQUE_DECODING 0
#define LM_SOUNDNESS_JOHNSON_BOUND 1
#define LM_SOUNDNESS_CAPACITY_BOUND 2
typedef struct {
    uint32_t starting_log_inv_rate;
    uint32_t max_num_variables_to_send_coeffs;
    uint32_t rs_domain_initial_reduction_factor;
    uint32_t folding_factor_first, folding_factor_subsequent;
    uint32_t soundness_type; /* LM_SOUNDNESS_* (SecurityAssumption, config.rs:445-455) */
    uint32_t security_level;
    uint32_t pow_bits;
} lm_whir_builder;
void lmh_default_whir_builder(uint32_t starting_log_inv_rate, int prox_gaps_conjecture, lm_whir_builder* out);
/* LM_E_INVALID (with lm_last_error) where the reference asserts / unwraps (config.rs:187-206, 311-314) */
int lmh_whir_config_new(const lm_whir_builder* builder, uint32_t num_variables, lm_whir_config* out);

/* SparseStatement / SparseValue (crates/whir/src/lib.rs:31-108), flattened:
 * statement s uses point coordinates points[point_offset .. +point_len) and values [values_offset .. +n_values) of
 * the parallel arrays selectors[] / values[] (EF = 5 words). */
typedef struct {
    uint32_t point_len;
    uint32_t is_next;
    uint32_t n_values;
    uint32_t reserved;
    uint64_t point_offset;
    uint64_t values_offset;
} lm_sparse_statement;

/* ---- ProverState (fiat-shamir/src/prover.rs:27-177; challenger.rs:9-76) ------------------------------------------ */
typedef struct lmh_prover lmh_prover;
lmh_prover* lmh_prover_new(void);
void lmh_prover_free(lmh_prover* p);
void lmh_add_base_scalars(lmh_prover* p, const uint32_t* scalars, uint64_t n);          /* FSProver::add_base_scalars */
void lmh_observe_scalars(lmh_prover* p, const uint32_t* scalars, uint64_t n);           /* FSProver::observe_scalars */
void lmh_add_extension_scalars(lmh_prover* p, const uint32_t* ef, uint64_t n_ef);       /* add_extension_scalars */
void lmh_duplex(lmh_prover* p);
int lmh_sample_vec(lmh_prover* p, uint64_t n_ef, uint32_t* out_ef);                     /* ChallengeSampler::sample_vec */
int lmh_sample_in_range(lmh_prover* p, uint32_t bits, uint64_t n, uint64_t* out);
/* add_sumcheck_polynomial(coeffs, eq_alpha): eq_alpha may be NULL */
void lmh_add_sumcheck_polynomial(lmh_prover* p, const uint32_t* coeffs_ef, uint32_t n_coeffs, const uint32_t* eq_alpha);
/* pow_grinding on the GPU with the canonical (smallest) witness */
int lmh_pow_grinding(lm_ctx* ctx, lmh_prover* p, uint32_t bits);
void lmh_challenger_state(const lmh_prover* p, uint32_t out16[16]);
/* Proof = transcript + un-pruned Merkle hints (RawProof, fiat-shamir/src/transcript.rs:20-31), serialised as u32 words:
 *   [T][transcript x T][M] then M x { index_lo, index_hi, leaf_len, path_len, leaf x leaf_len, path x path_len } */
uint64_t lmh_proof_words(const lmh_prover* p);
void lmh_proof_copy(const lmh_prover* p, uint32_t* out);

/* Merkle-path pruning, the first step of the reference's proof wire format (MerklePaths::prune,
 * crates/backend/fiat-shamir/src/merkle_pruning.rs:18-86; `Proof { transcript, merkle_paths }`, transcript.rs:33-36).  One
 * batch = one hint_merkle_paths call (the query set of one commitment).  Blob of u32 words:
 *   [T][transcript x T][B]  B x { height, n_trailing_zeros, n_orig, original_order x n_orig, n_paths,
 *                                 n_paths x { index_lo, index_hi, leaf_len, leaf x leaf_len, n_sib, sibling x 8 n_sib } }
 * lmh_proof_size_fe = Proof::proof_size_fe (transcript.rs:39-53); the reference quotes proof sizes as
 * size_fe * 31 / 8192 KiB (rec_aggregation/src/benchmark.rs:447).  postcard + lz4 framing is not produced here.
 * lmh_proof_batch_sizes: the openings per batch of the un-pruned blob (what a verifier-side restore needs to re-batch). */
uint64_t lmh_proof_pruned_words(const lmh_prover* p);
void lmh_proof_pruned_copy(const lmh_prover* p, uint32_t* out);
uint64_t lmh_proof_size_fe(const lmh_prover* p);
uint32_t lmh_proof_n_batches(const lmh_prover* p);
void lmh_proof_batch_sizes(const lmh_prover* p, uint32_t* out);

/* Load a proof produced elsewhere (the un-pruned u32 blob of lmh_proof_copy + the openings per batch) into a prover object,
 * e.g. to prune / serialise it.  The challenger is not replayed. */
int lmh_prover_load_raw(lmh_prover* p, const uint32_t* blob, uint64_t n_words, const uint32_t* batch_sizes, uint32_t n_batches);

/* ---- proof bytes -------------------------------------------------------------------------------
 * lmh_proof_postcard: the reference's serialised ExecutionProof — serde + postcard of Proof { transcript: Vec<F>,
 * merkle_paths: Vec<PrunedMerklePaths<F, F>> } (crates/backend/fiat-shamir/src/transcript.rs:33-36, merkle_pruning.rs:5-12,
 * crates/lean_prover/src/prove_execution.rs:12-18; a field element is the varint of its Montgomery word, monty_31.rs:152-157).
 * lmh_proof_compressed: the same bytes through lz4_flex::compress_prepend_size (u32 LE length + one LZ4 block,
 * rec_aggregation/src/type_1_aggregation.rs:81-89) — any valid block decodes to the same postcard stream.
 * lmh_proof_from_postcard / lmh_proof_decompress: the inverse (postcard::from_bytes / decompress_size_prepended), NULL with
 * lm_last_error on malformed input. */
uint64_t lmh_proof_postcard_size(const lmh_prover* p);
void lmh_proof_postcard(const lmh_prover* p, uint8_t* out);
uint64_t lmh_proof_compressed_size(const lmh_prover* p);
void lmh_proof_compressed(const lmh_prover* p, uint8_t* out);
uint64_t lmh_lz4_compress_bound(uint64_t n);
uint64_t lmh_lz4_compress_prepend_size(const uint8_t* in, uint64_t n, uint8_t* out);
/* out == NULL: returns the size prefix; else the number of bytes written (== prefix) or -1 */
int64_t lmh_lz4_decompress_size_prepended(const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap);
typedef struct lmh_proof lmh_proof; /* a decoded Proof<F> */
lmh_proof* lmh_proof_from_postcard(const uint8_t* bytes, uint64_t n);
lmh_proof* lmh_proof_decompress(const uint8_t* bytes, uint64_t n);
void lmh_proof_free(lmh_proof* p);
uint64_t lmh_proof_decoded_size_fe(const lmh_proof* p);
/* the decoded proof in the word layout of lmh_proof_pruned_copy; out may be NULL (size query) */
uint64_t lmh_proof_decoded_pruned_words(const lmh_proof* p, uint32_t* out);

/* ---- WHIR ------------------------------------------------------------------------------------------------------- */
typedef struct lmh_witness lmh_witness; /* Witness, commit.rs:49-57: device-resident tree + OOD points/answers */
/* WhirConfig::commit (commit.rs:64-99): d_poly = 2^num_variables base words in HBM. */
int lmh_whir_commit(lm_ctx* ctx, lmh_prover* p, const lm_whir_config* cfg, const uint32_t* d_poly, uint64_t actual_len,
                    lmh_witness** out);
void lmh_witness_free(lm_ctx* ctx, lmh_witness* w);
void lmh_witness_root(const lmh_witness* w, uint32_t root[8]);
/* WhirConfig::prove (open.rs:37-56).  Consumes the witness.  out_point: num_variables x 5 words (the folding
 * randomness, MultilinearPoint). */
int lmh_whir_prove(lm_ctx* ctx, lmh_prover* p, const lm_whir_config* cfg, const lm_sparse_statement* statements,
                   uint32_t n_statements, const uint32_t* points, uint64_t n_point_coords, const uint64_t* selectors,
                   const uint32_t* values, uint64_t n_values, lmh_witness* witness, const uint32_t* d_poly,
                   uint32_t* out_point);

/* ---- logup GKR ---------------------------------------------------------------------------------------------------
 * prove_gkr_quotient (crates/sub_protocols/src/quotient_gkr/mod.rs:31-78): returns the quotient sum n_i/d_i, the claim
 * point (n_vars x 5) and the two final claims (numerators MLE, denominators MLE at the point; mod.rs:74-77). */
int lmh_prove_gkr_quotient(lm_ctx* ctx, lmh_prover* p, const uint32_t* d_nums, const uint32_t* d_dens, uint32_t n_vars,
                           uint32_t out_quotient[5], uint32_t* out_point, uint32_t out_claims[10]);

/* entries [active_len, 2^n_vars) are the neutral pair (0, 1) and are never read (lm_gkr_build_active) */
int lmh_prove_gkr_quotient_active(lm_ctx* ctx, lmh_prover* p, const uint32_t* d_nums, const uint32_t* d_dens, uint32_t n_vars,
                                  uint64_t active_len, uint32_t out_quotient[5], uint32_t* out_point, uint32_t out_claims[10]);

/* ---- batched AIR sumcheck -----------------------------------------------------------------------------------------
 * prove_batched_air_sumcheck (crates/sub_protocols/src/air_sumcheck.rs:636-681) over one session per table, followed by
 * sending every table's final column evaluations (crates/lean_prover/src/prove_execution.rs:212-214).  Tables must be
 * given in the reference's order (sort_tables_by_height: descending height, stable).  out_point: n_max x 5 challenges
 * in sumcheck order (LSB first); out_col_evals: concatenation over tables of (n_columns + n_shift) x 5 words. */
typedef struct {
    uint32_t table;                /* 0 execution, 1 extension_op, 2 poseidon16 */
    uint32_t log_rows;
    const uint32_t* const* d_cols; /* host array of device column pointers */
    const uint32_t* eq_point;      /* log_rows x 5 (from_end(gkr_point, log_rows)) */
    uint32_t sum[5];               /* bus_final_value */
    uint32_t non_padded_n_rows;    /* TableTrace::non_padded_n_rows (rows behind it are the padding row); 0 = unknown: all rows */
} lm_air_table;
int lmh_prove_batched_air_sumcheck(lm_ctx* ctx, lmh_prover* p, const lm_air_table* tables, uint32_t n_tables,
                                   const uint32_t alpha[5], const uint32_t* logup_eq16, const uint32_t bus_beta[5],
                                   const uint32_t eta[5], uint32_t* out_point, uint32_t* out_col_evals);

/* ---- prove_execution after witness generation --------------------------------------------------------------------
 * crates/lean_prover/src/prove_execution.rs:47-274: Fiat–Shamir preamble, stack_polynomials_and_commit, prove_generic_logup
 * (fill + GKR + column evaluations), batched AIR sumcheck, statement assembly (stacked_pcs_global_statements) and
 * WhirConfig::prove.  Inputs are what get_execution_trace (lean_prover/src/trace_gen.rs) hands to the prover, resident in
 * HBM: memory, the access counters, the bytecode multilinear (2^log_bytecode rows x 16 words) and per table the TOTAL
 * column set (committed columns first, then the virtual bus columns: execution 24, extension_op 31, poseidon16 111).
 * cfg must be the WhirConfig integers for lmh_stacked_n_vars(trace) variables. */
typedef struct {
    uint32_t log_rows;
    uint32_t non_padded_n_rows;    /* TableTrace::non_padded_n_rows (trace_gen.rs:183): the rows behind it are the table's padding
                                      row, as pad_table / lmh_pad_table write them; 0 = unknown (every row is summed) */
    const uint32_t* const* d_cols; /* host array of device column pointers (n_columns_total of the table) */
} lm_vm_table;
typedef struct {
    uint32_t log_inv_rate, log_memory, log_bytecode, ending_pc, public_memory_size, n_public_input;
    const uint32_t* public_input;  /* host */
    const uint32_t* bytecode_hash; /* host, 8 words */
    const uint32_t* d_bytecode;    /* device */
    const uint32_t* d_bytecode_acc; /* access counters: NULL = computed on the device (prove_execution.rs:90-110) */
    const uint32_t* d_memory;
    const uint32_t* d_memory_acc;   /* NULL = computed on the device */
    lm_vm_table tables[3]; /* indexed by table id: execution, extension_op, poseidon16 */
} lm_execution_trace;
uint32_t lmh_stacked_n_vars(const lm_execution_trace* trace); /* compute_stacked_n_vars, stacked_pcs.rs:183-196 */
/* returns LM_E_INVALID with lm_last_error "logup sum != 0" when the witness is inconsistent (prove_generic_logup asserts) */
int lmh_prove_execution(lm_ctx* ctx, lmh_prover* p, const lm_execution_trace* trace, const lm_whir_config* cfg);

/* ---- verifier --------------------------------------------------------------------------------
 * verify_execution (crates/lean_prover/src/verify_execution.rs:14-214) with everything below it — VerifierState and Merkle
 * path restoration (fiat-shamir/src/{verifier,merkle_pruning}.rs), verify_gkr_quotient, verify_generic_logup, the batched
 * AIR check, stacked_pcs_global_statements, WhirConfig::verify (whir/src/verify.rs:83-435) — on the host (the reference's
 * verifier is CPU code; it is a few ms plus one pass over the bytecode table).  The instance is what the reference's
 * `Bytecode` + public input provide.  builder = NULL: default_whir_config(rate read from the proof), as the reference does.
 * LM_OK = accepted; LM_E_INVALID with the failing check in lm_last_error otherwise. */
typedef struct {
    uint32_t log_bytecode, ending_pc, n_public_input, reserved;
    const uint32_t* public_input;  /* n_public_input words */
    const uint32_t* bytecode_hash; /* 8 words */
    const uint32_t* bytecode;      /* instructions_multilinear: 2^log_bytecode rows x 16 words (12 used), host */
} lm_verify_instance;
int lmh_verify_execution(const lm_verify_instance* instance, const lmh_proof* proof, const lm_whir_builder* builder);
int lmh_verify_execution_bytes(const lm_verify_instance* instance, const uint8_t* bytes, uint64_t n, int compressed,
                               const lm_whir_builder* builder);
/* the proof still held by a prover object (pruned and restored on the way, like a proof that travelled) */
int lmh_verify_execution_prover(const lm_verify_instance* instance, const lmh_prover* p, const lm_whir_builder* builder);

/* ---- pad_table (crates/lean_prover/src/trace_gen.rs:170-191) ----------------------------------------------------------------
 * lmh_table_log_rows = log2_ceil(n_rows + 1).max(MIN_LOG_N_ROWS_PER_TABLE): the height get_execution_trace gives a table.
 * lmh_pad_table fills rows [n_rows, 2^log_rows) of every committed column (device pointers in the host array d_cols; 20 / 29 /
 * 109 for table 0 / 1 / 2) with the table's padding row (execution/mod.rs:59-74, extension_op/mod.rs:125-134,
 * poseidon_16/mod.rs:182-205; canonical pointers in, Montgomery words out).  Poseidon table: the 84 derived columns of the
 * padded rows are zero-filled — run lm_poseidon_trace (and lm_poseidon_trace_outputs_from_memory) over the whole table
 * afterwards, exactly as for the active rows (the reference fills them "later with SIMD" too). */
uint32_t lmh_table_log_rows(uint64_t n_rows);
int lmh_pad_table(lm_ctx* ctx, uint32_t table, uint32_t* const* d_cols, uint64_t n_rows, uint32_t log_rows, uint32_t zero_vec_ptr,
                  uint32_t null_hash_ptr, uint32_t ending_pc);

/* ---- host Poseidon1-16 (poseidon1_koalabear_16.rs:873-1030) ----------------------------------------------------------------
 * The transcript's permutation: ~1300 strictly sequential calls per proof between device launches, so its latency is on the
 * critical path.  "avx512-ifma" when the CPU has AVX-512 F/DQ/BW/VL/IFMA (the reference's own Poseidon is AVX2/AVX-512/NEON
 * code), else "scalar"; LM_HOST_POSEIDON_SCALAR=1 forces the scalar code.  Both are the same function. */
const char* lmh_poseidon_backend(void);
void lmh_poseidon16_permute(uint32_t state[16]);        /* the backend in use */
void lmh_poseidon16_permute_scalar(uint32_t state[16]); /* always the scalar code */

#ifdef __cplusplus
}
#endif
#endif
