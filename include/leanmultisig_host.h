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
    /* Optional (NULL = absent): the trace already lives in its committed layout.  d_stacked = 2^lmh_stacked_n_vars(trace) words
     * laid out as stack_polynomials does (stacked_pcs.rs:118-136: memory | memory_acc | bytecode_acc (region padded) | the
     * committed columns of the tables by descending height | zero tail), d_memory and every committed column pointer above point
     * INTO it at their stacked offsets, and everything outside memory, the access counters and the columns is zero.  The prover
     * then commits this buffer as it is — no 2^n_vars-word copy ("TODO avoid cloning", stacked_pcs.rs:115) — and writes the
     * access counters it computes into their slots.  lmh_get_execution_trace builds its traces this way. */
    uint32_t* d_stacked;
} lm_execution_trace;
uint32_t lmh_stacked_n_vars(const lm_execution_trace* trace); /* compute_stacked_n_vars, stacked_pcs.rs:183-196 */
/* Wall clock of the last lmh_prove_execution on this prover, per stage, in ms — the reference's tracing spans around the same code
 * (`--tracing`, crates/utils/src/logs.rs): no synchronisation is added, a boundary is where the host holds the stage's result. */
#define LMH_STAGE_COMMIT 0        /* stack_polynomials_and_commit (stacked_pcs.rs:98): access counts, "FFT", "build merkle tree", "ood evaluation" */
#define LMH_STAGE_LOGUP_FILL 1    /* prove_generic_logup (logup.rs:26), numerators / denominators */
#define LMH_STAGE_GKR 2           /* prove_gkr_quotient (quotient_gkr/mod.rs:31) */
#define LMH_STAGE_COLUMN_EVALS 3  /* the column evaluations at the GKR point (logup.rs:224-308) */
#define LMH_STAGE_AIR 4           /* "batched AIR sumcheck" (prove_execution.rs:209) */
#define LMH_STAGE_WHIR_PROVE 5    /* statement assembly + "WHIR prove" (open.rs:36) */
#define LMH_N_STAGES 8
int lmh_prover_stage_times(const lmh_prover* p, double out_ms[LMH_N_STAGES]);

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

/* ---- RawProof: the input of the recursion program ------------------------------------------------------------------------------
 * VerifierState::into_raw_proof (fiat-shamir/src/verifier.rs:21,47-60,126-195): while it verifies, the reference's verifier rebuilds
 * the transcript in the format its in-VM verifier reads (rec_aggregation/zkdsl_implem/fiat_shamir.py) — every absorbed slice
 * zero-padded to the rate, sumcheck polynomials with ALL their coefficients, a grinding witness as a block of its own — and
 * aggregate_type_1 hands it to the VM as the `proof_transcript` hint (type_1_aggregation.rs:310-356); the Merkle openings travel
 * un-pruned beside it (lmh_proof_copy has them in opening order).  lm_whir_opening_claim is what the PCS opening of that
 * verification was asked to prove, i.e. the arguments and the expected results of `whir_open` (zkdsl_implem/whir.py:18-164,
 * called at recursion.py:470-532, its results used at :534-654). */
typedef struct {
    uint64_t transcript_offset;    /* raw-transcript word WhirConfig::verify reads first (after the duplex of recursion.py:470) */
    uint32_t challenger_state[16]; /* the sponge at that point: [capacity | rate] */
    uint32_t num_variables, log_inv_rate, n_ood, n_statement_values;
    uint32_t root[8];                           /* the stacked commitment (parse_commitment, recursion.py:94) */
    uint32_t ood_points[4 * 5], ood_answers[4 * 5];
    uint32_t combination_gen[5];                /* recursion.py:472 */
    uint32_t statement_sum[5];     /* sum_i gen^(n_ood + i) value_i over the statement: whir_sum without its OOD part (:478-518) */
    uint32_t statement_weights[5]; /* sum_i gen^(n_ood + i) weight_i(folding randomness): what :534-652 add to `s` */
    uint32_t folding_randomness[32 * 5];        /* folding_randomness_global, num_variables entries */
} lm_whir_opening_claim;
/* The statement the PCS opening proves, as the recursion program assembles it (recursion.py:469-518 before whir_open, :534-652 after):
 * the three evaluation points the verifier sampled, the table heights, and WHERE in the raw transcript the claimed evaluations lie
 * (every value the statement carries was received from the prover: value_acc / value_memory / value_bytecode_acc, the logup column
 * evaluations of every table, the column evaluations behind the batched AIR sumcheck). */
typedef struct {
    uint32_t log_rows[3], log_memory, log_bytecode, gkr_n_vars, n_max, table_order[3]; /* table_order: table ids by descending height */
    uint32_t ending_pc, log_public_memory; /* log2 of the public memory (the public input padded to a power of two) */
    uint32_t gkr_point[32 * 5];   /* point_gkr, gkr_n_vars entries */
    uint32_t air_point[32 * 5];   /* the batched AIR sumcheck's challenges in sumcheck order, n_max entries */
    uint32_t pm_point[8 * 5];     /* public_memory_random_point, log2(public memory size) entries */
    uint64_t off_value_memory_acc, off_value_memory, off_value_bytecode_acc; /* raw-transcript word offsets (5 words each) */
    uint64_t off_inner_evals[3];  /* per table id: (n_columns + n_shift) x 5 words, flat columns first */
    uint32_t n_logup_values[3];   /* per table id: logup column evaluations, in the order the prover sent them */
    uint32_t logup_col[3][40];
    uint64_t logup_off[3][40];
    /* the batched AIR sumcheck in front of the statement (recursion.py:383-467): where it starts, and what it needs from before */
    uint64_t air_offset;               /* raw-transcript word of the first round polynomial */
    uint32_t air_challenger_state[16]; /* the sponge when bus_beta is sampled (:385) */
    uint32_t air_degree, reserved2;    /* MAX_AIR_FULL_DEGREE: a round polynomial has air_degree + 1 coefficients */
    uint32_t logup_c[5];
    uint64_t off_bus_selector[3], off_bus_data[3]; /* per table id: eval_on_selector / eval_on_data (:331-339) */
    uint32_t air_constraint_evals[3][5];           /* per table id: evaluate_air_constraints at the column evaluations (:434) */
    /* the head of the verifier (recursion.py:48-378): what it does not derive itself */
    uint32_t bytecode_hash_domsep[8];              /* compress(bytecode hash | SNARK_DOMAIN_SEP), observed at :56 */
    uint32_t bytecode_value[5];                    /* the `bytecode_value_hint` of :138: the bytecode table at (point, alphas) */
    uint32_t reserved3;
} lm_pcs_statement_claim;
typedef struct lmh_raw_proof lmh_raw_proof;
/* verify_execution on the proof a prover object holds; on success *out owns the raw transcript and the claim */
int lmh_verify_execution_raw(const lm_verify_instance* instance, const lmh_prover* p, const lm_whir_builder* builder, lmh_raw_proof** out);
const uint32_t* lmh_raw_proof_transcript(const lmh_raw_proof* r, uint64_t* n_words);
const lm_whir_opening_claim* lmh_raw_proof_whir_claim(const lmh_raw_proof* r);
const lm_pcs_statement_claim* lmh_raw_proof_statement_claim(const lmh_raw_proof* r);
void lmh_raw_proof_free(lmh_raw_proof* r);

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

/* ---- leanVM: bytecode, runner, witness generation ------------------------------------------------
 * The reference's prove_execution starts from (bytecode, public input, witness hints) and runs the VM itself
 * (crates/lean_prover/src/prove_execution.rs:20-39: try_execute_bytecode + get_execution_trace are inside the timed region
 * of the benchmark, rec_aggregation/src/benchmark.rs:397-410).  This section is that part:
 *   lmh_bytecode_*            `Bytecode` (crates/lean_vm/src/isa/bytecode.rs:17-31) as an opaque object built from the same two
 *                             things the reference's compiler emits: instructions_multilinear (the field representation of every
 *                             instruction, lean_compiler/src/instruction_encoder.rs:4-113 — injective, so the runner decodes
 *                             its instructions from it) and the hints attached to each pc (lean_vm/src/isa/hint.rs:17-81);
 *   lmh_execute_bytecode      try_execute_bytecode / execute_bytecode_helper (lean_vm/src/execution/runner.rs:27-343) including
 *                             the parallel loop batches (handle_parallel_batch :361-482, SegmentMemory memory.rs:118-189) on a
 *                             host thread pool and resolve_deref_hints (:206-236);
 *   lmh_get_execution_trace   get_execution_trace (lean_prover/src/trace_gen.rs:14-168): upload of the VM log and construction of
 *                             every table column on the device;
 *   lmh_prove_execution_vm    prove_execution (prove_execution.rs:20-274) whole.
 * All addresses, offsets, pcs and fps are plain (canonical) integers; field VALUES (memory words, hint data, constants of
 * instructions_multilinear) are Montgomery words like everywhere else. */
#define LM_VM_ARG_CONST 0 /* MemOrConstant::Constant / MemOrFpOrConstant::Constant: canonical value */
#define LM_VM_ARG_MEM 1   /* MemoryAfterFp { offset }: m[fp + value] */
#define LM_VM_ARG_FP 2    /* FpRelative { offset }: fp + value */
/* Hint kinds (lean_vm/src/isa/hint.rs:17-81).  Print / LocationReport / Label / Panic have no effect on the execution
 * result and are not represented. */
#define LM_VM_HINT_INVERSE 1                    /* args: arg, res_offset */
#define LM_VM_HINT_REQUEST_MEMORY 2             /* args: offset, size */
#define LM_VM_HINT_DEREF 3                      /* DerefHint: args: offset_src, offset_target */
#define LM_VM_HINT_DECOMPOSE_BITS_XMSS 4        /* CustomHint (hint.rs:137-203), args as in the reference */
#define LM_VM_HINT_DECOMPOSE_BITS_MERKLE_WHIR 5
#define LM_VM_HINT_DECOMPOSE_BITS 6
#define LM_VM_HINT_LESS_THAN 7
#define LM_VM_HINT_LOG2_CEIL 8
#define LM_VM_HINT_WITNESS_INLINE 9             /* HintWitness, destination Inline: args: name id, offset */
#define LM_VM_HINT_WITNESS_INDIRECT 10          /* destination Indirect: args: name id, ptr_offset */
#define LM_VM_HINT_PARALLEL_BATCH_START 11      /* args: n_args, end_value */
#define LM_VM_HINT_DEBUG_ASSERT 12              /* args: left, right, kind (0 ==, 1 !=, 2 <, 3 <=), preceds_runtime_inequality */
typedef struct {
    uint32_t pc;      /* executed before the instruction at pc; hints of one pc run in array order */
    uint32_t kind;    /* LM_VM_HINT_* */
    uint32_t args[4];
    uint8_t mode[4];  /* LM_VM_ARG_* of each argument that is an operand in the reference; 0 for plain integers */
} lm_vm_hint;

typedef struct lmh_bytecode lmh_bytecode;
/* instructions_multilinear: 2^log_size rows x 16 words (12 used, execution/air.rs:18-30 order: operand_a/b/c, flag_a/b/c,
 * flag_c_fp, flag_ab_fp, mul, jump, aux, precompile_data); rows [n_instructions, 2^log_size) are zero (code.len() = n_instructions).
 * hints sorted by pc.  NULL with lm_last_error on an undecodable row. */
lmh_bytecode* lmh_bytecode_new(const uint32_t* instructions_multilinear, uint32_t log_size, uint64_t n_instructions,
                               uint32_t ending_pc, uint32_t starting_frame_memory, const lm_vm_hint* hints, uint64_t n_hints,
                               uint32_t n_hint_names);
void lmh_bytecode_free(lmh_bytecode* bc);
/* Bytecode::hash = poseidon_compress_slice(instructions_multilinear, true) (lean_compiler/src/c_compile_final.rs:158,
 * utils/src/poseidon.rs:41-67); computed once, 2^(log_size + 1) sequential compressions */
void lmh_bytecode_hash(const lmh_bytecode* bc, uint32_t out[8]);
/* The names of the HintWitness streams by id (the keys of ExecutionWitness::hints, runner.rs:17-25): needed only by callers that
 * build a witness by NAME (lmh_aggregate_type_1_witness); lmh_bytecode_hint_name_id: -1 when the bytecode has no such stream. */
int lmh_bytecode_set_hint_names(lmh_bytecode* bc, const char* const* names, uint32_t n_names);
int lmh_bytecode_hint_name_id(const lmh_bytecode* bc, const char* name);
uint32_t lmh_bytecode_n_hint_names(const lmh_bytecode* bc);
uint32_t lmh_bytecode_log_size(const lmh_bytecode* bc);
uint32_t lmh_bytecode_ending_pc(const lmh_bytecode* bc);
const uint32_t* lmh_bytecode_multilinear(const lmh_bytecode* bc);

/* ExecutionWitness (runner.rs:17-25): hints: HashMap<String, Vec<Vec<F>>> flattened; names are the ids of the bytecode's
 * HintWitness hints. */
typedef struct {
    uint32_t preamble_memory_len;
    uint32_t n_names;
    const uint64_t* name_entry_begin; /* n_names + 1: the entries of name k are [begin[k], begin[k + 1]) */
    const uint64_t* entry_offset;     /* n_entries + 1: the words of entry e are data[offset[e] .. offset[e + 1]) */
    const uint32_t* data;             /* Montgomery words */
} lm_vm_witness;

#define LM_VM_POSEIDON_CALL_WORDS 9  /* arg_a, arg_b, res, half_output, hardcoded_left, offset, left_first, left_second, permute */
#define LM_VM_EXTENSION_ROW_WORDS 24 /* is_be, start, add, mul, poly_eq, len, idx_a, idx_b, idx_res (canonical) | VB(5) VRES(5) COMP(5) (Montgomery) */
typedef struct lmh_execution lmh_execution; /* ExecutionResult (lean_vm/src/diagnostics/exec_result.rs) */
typedef struct {
    uint64_t n_cycles;            /* pcs.len(), the final ending_pc row included */
    const uint32_t* pcs;
    const uint32_t* fps;
    uint64_t memory_len;          /* memory.0.len() */
    const uint32_t* memory;       /* Montgomery words, undefined cells read 0; the 24 words behind memory_len hold the zero vector and
                                     poseidon16(0) that get_execution_trace appends (trace_gen.rs:106-110) */
    const uint8_t* memory_defined; /* 1 = Some(_) */
    uint64_t public_memory_size, runtime_memory_size;
    uint64_t n_poseidon_calls;
    const uint32_t* poseidon_calls;  /* n x LM_VM_POSEIDON_CALL_WORDS: Poseidon16Precompile::execute's pushes (poseidon_16/mod.rs:262-286) */
    uint64_t n_extension_rows;
    const uint32_t* extension_rows;  /* n x LM_VM_EXTENSION_ROW_WORDS: exec_multi_row's pushes (extension_op/exec.rs:149-186) */
    uint64_t n_add, n_mul, n_deref, n_jump; /* InstructionCounts */
} lm_vm_execution_view;
/* n_threads = 0: all hardware threads (capped at 128).  LM_E_INVALID + lm_last_error = the RunnerError and its pc. */
int lmh_execute_bytecode(const lmh_bytecode* bc, const uint32_t* public_input, uint32_t n_public_input, const lm_vm_witness* witness,
                         uint32_t n_threads, lmh_execution** out);
/* The same run with the PARALLEL LOOP BATCHES of the program (Hint::ParallelBatchStart, handle_parallel_batch, runner.rs:369-482) on
 * the device of `ctx`: the host runs the sequential parts, every independent segment of a batch is interpreted by one wavefront
 * (csrc/lm_vm_device.hip: frame in LDS, Poseidon16 on 16 lanes, SegmentMemory semantics), the segments' logs and frames stay in
 * HBM, resolve_deref_hints runs there, and lmh_get_execution_trace on the same context builds the tables from them without an
 * upload.  A batch the device cannot take (fewer than 32 segments, a frame above 14000 words, more than 16 call-frame arguments)
 * or in which anything irregular happens (a RunnerError in a segment, conflicting deferred writes) is run by the host pool exactly
 * as lmh_execute_bytecode does — results and errors are the same on both paths.  lmh_execution_view downloads the log on its
 * first call (on the context's thread); an execution released after its context was destroyed only drops its host side (the device
 * buffers went with the context's pool). */
int lmh_execute_bytecode_device(lm_ctx* ctx, const lmh_bytecode* bc, const uint32_t* public_input, uint32_t n_public_input,
                                const lm_vm_witness* witness, uint32_t n_threads, lmh_execution** out);
int lmh_execution_on_device(const lmh_execution* e); /* 1: at least one batch ran on the device and the log is resident there */
/* Where the parallel batches of a run executed.  A batch the device hands back to the host pool is not an error (results and errors are
 * the same on both paths), but a node that sizes itself for the device path wants to know: bench.py reports these fields and fails
 * when its default workload falls back. */
typedef struct {
    uint32_t on_device;         /* = lmh_execution_on_device */
    uint32_t n_device_batches;  /* batches interpreted by k_vm_segments */
    uint32_t n_host_batches;    /* batches run by the host pool (handle_parallel_batch) */
    uint32_t run_repeated;      /* 1: the device could not decide resolve_deref_hints and the whole run was repeated on the host */
    char host_batch_reason[256]; /* why the FIRST host batch did not run on the device ("" when none did) */
} lm_vm_run_info;
void lmh_execution_info(const lmh_execution* e, lm_vm_run_info* out);
void lmh_execution_free(lmh_execution* e);
void lmh_execution_view(const lmh_execution* e, lm_vm_execution_view* out);

/* get_execution_trace on the device: every column of the three tables (padding rows included), the padded memory image
 * (+ zero vector and poseidon16(0) behind the VM's memory, trace_gen.rs:103-113, grown to >= 2^16 and >= the bytecode as
 * prove_execution.rs:41-46 does), the bytecode table.  The result owns its device buffers. */
typedef struct lmh_vm_trace lmh_vm_trace;
int lmh_get_execution_trace(lm_ctx* ctx, const lmh_bytecode* bc, const lmh_execution* e, const uint32_t* public_input,
                            uint32_t n_public_input, uint32_t log_inv_rate, lmh_vm_trace** out);
const lm_execution_trace* lmh_vm_trace_view(const lmh_vm_trace* t);
void lmh_vm_trace_free(lm_ctx* ctx, lmh_vm_trace* t);

/* prove_execution(bytecode, public_input, witness, whir_config) -> proof in `p`.  times_ms (nullable): [0] VM run, [1] trace
 * upload + column construction (until the device work is enqueued), [2] proving. */
int lmh_prove_execution_vm(lm_ctx* ctx, lmh_prover* p, const lmh_bytecode* bc, const uint32_t* public_input, uint32_t n_public_input,
                           const lm_vm_witness* witness, const lm_whir_builder* builder, uint32_t n_threads, double times_ms[3]);
/* the same, also reporting where the VM's parallel batches ran (info nullable) */
int lmh_prove_execution_vm_info(lm_ctx* ctx, lmh_prover* p, const lmh_bytecode* bc, const uint32_t* public_input, uint32_t n_public_input,
                                const lm_vm_witness* witness, const lm_whir_builder* builder, uint32_t n_threads, double times_ms[3],
                                lm_vm_run_info* info);

/* ---- aggregate_type_1 (crates/rec_aggregation/src/type_1_aggregation.rs:206-377), raw signatures only -------------------------------
 * What the reference's benchmark times per aggregation (rec_aggregation/src/benchmark.rs:397-410) is aggregate_type_1 WHOLE: before it
 * calls prove_execution it sorts and de-duplicates the (public key, signature) pairs (:232-233), hashes the sorted public keys
 * (hash_pubkeys :118-121), builds the tweak table of the slot and hashes it (compute_tweak_table :124-151), assembles the public-input
 * buffer (build_type1_input_data :163-186) and hashes it onto the 8-word public input (:262), and flattens the signatures into the
 * named hint streams of the ExecutionWitness (:264-365).  lmh_aggregate_type_1_witness is that part, lmh_aggregate_type_1 the whole
 * function for n_recursions = 0 (children = &[]: the recursion branch needs the in-VM verifier, which this library does not assemble).
 * The three hashes are chains of dependent compressions (n_sigs + 184 + 21 of them at 1550 signatures): they run on the calling
 * thread with the transcript's AVX-512 permutation — a chain has no parallelism for a device kernel to use.
 * One signature = LM_XMSS_SIG_WORDS Montgomery words: XmssPublicKey { merkle_root[4], public_param[4] }, then XmssSignature {
 * wots_signature.randomness[6], wots_signature.chain_tips[42][4], merkle_proof[32][4] } (crates/xmss/src/{xmss.rs:19-29,wots.rs:17-25}). */
#define LM_XMSS_V 42
#define LM_XMSS_LOG_LIFETIME 32
#define LM_XMSS_SIG_WORDS (4 + 4 + 6 + LM_XMSS_V * 4 + LM_XMSS_LOG_LIFETIME * 4)
typedef struct lmh_type1_witness lmh_type1_witness;
/* bc must carry its hint names (lmh_bytecode_set_hint_names); streams the program does not read are dropped, streams it reads and a raw
 * aggregation leaves empty stay empty.  message: 8 words.  LM_E_INVALID: no signatures, more than MAX_XMSS_AGGREGATED = 2^15. */
int lmh_aggregate_type_1_witness(const lmh_bytecode* bc, const uint32_t* raw_xmss, uint64_t n_raw, const uint32_t message[8], uint32_t slot,
                                 lmh_type1_witness** out);
const lm_vm_witness* lmh_type1_witness_vm(const lmh_type1_witness* w);          /* ExecutionWitness { preamble_memory_len, hints } */
const uint32_t* lmh_type1_witness_public_input(const lmh_type1_witness* w);     /* 8 words: poseidon_compress_slice(pub_input_data, true) */
const uint32_t* lmh_type1_witness_input_data(const lmh_type1_witness* w, uint64_t* n_words); /* pub_input_data */
uint64_t lmh_type1_witness_n_sigs(const lmh_type1_witness* w);                  /* global_pub_keys.len() after sort + dedup */
const uint32_t* lmh_type1_witness_pubkeys(const lmh_type1_witness* w);          /* TypeOneInfo::pubkeys, sorted, 8 words each */
void lmh_type1_witness_free(lmh_type1_witness* w);
/* aggregate_type_1(&[], raw_xmss, message, slot, log_inv_rate) -> proof in `p`.  times_ms (nullable): [0] the input assembly above,
 * [1] VM run, [2] trace, [3] proving.  info (nullable): where the VM's parallel batches ran. */
int lmh_aggregate_type_1(lm_ctx* ctx, lmh_prover* p, const lmh_bytecode* bc, const uint32_t* raw_xmss, uint64_t n_raw, const uint32_t message[8],
                         uint32_t slot, const lm_whir_builder* builder, uint32_t n_threads, double times_ms[4], lm_vm_run_info* info);

/* n independent Poseidon1-16 compressions perm(x) + x of 16-word states on the host thread pool (signers, hint builders) */
void lmh_poseidon16_compress_many(uint32_t* states, uint64_t n, uint32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
