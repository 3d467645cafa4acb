/* leanmultisig.h — C ABI of the MI355X-native proving hot path of leanMultisig.
 *
 * Drop-in boundary: the reference has no FFI; the entry points below are what a Rust `extern "C"`
 * shim inside crates/whir, crates/sub_protocols and crates/backend/{sumcheck,fiat-shamir} would bind to replace the
 * bodies of the cited Rust functions (citations are relative to the reference checkout).  INTEGRATION.md shows the shim.
 *
 * Conventions
 *   - every field value is a u32 in Montgomery form (R = 2^32) of KoalaBear p = 0x7f000001 — the in-memory
 *     representation of the reference (crates/backend/koala-bear/src/monty_31/monty_31.rs:33-41);
 *   - an extension element (EF) handed over on the HOST is 5 consecutive u32 (quintic_extension/extension.rs:25-35);
 *   - a DEVICE array of EF of length n is SoA: 5 planes of n u32 (plane k = coefficient of X^k); lm_ef_aos_to_soa /
 *     lm_ef_soa_to_aos convert on the device;
 *   - pointers named d_* are device (HBM) pointers, everything else is host memory;
 *   - multilinear convention of the reference: point[0] <-> most significant index bit (poly/src/evals.rs:142-347);
 *   - one lm_ctx per GPU/stream, not re-entrant; calls are serialised by the (single) prover thread, like the single
 *     `&mut impl FSProver` of the reference (fiat-shamir/src/traits.rs:15-44);
 *   - every function returns 0 on success, a negative LM_E_* code otherwise; nothing unwinds across the ABI;
 *   - integers derived from f64 maths in WhirConfig::new (query counts, PoW bits, folding factors;
 *     whir/src/config.rs:146-334) are inputs, never re-derived here.
 */
#ifndef LEANMULTISIG_H
#define LEANMULTISIG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LM_OK 0
#define LM_E_INVALID (-1)  /* bad argument */
#define LM_E_DEVICE (-2)   /* HIP runtime error (see lm_last_error) */
#define LM_E_NOMEM (-3)

#define LM_DIGEST_WORDS 8 /* symetric/src/merkle.rs:11 */
#define LM_EF_DIM 5

typedef struct lm_ctx lm_ctx;
typedef struct lm_tree lm_tree;

/* ---- context, memory ------------------------------------------------------------------------------------------- */
int lm_ctx_create(int device, lm_ctx** out);
void lm_ctx_destroy(lm_ctx* ctx);
const char* lm_last_error(void);
/* The HIP "current device" is a property of the host thread: a thread that did not create the context calls this once
 * before using it (the lmh_* drivers do it themselves).  One context = one stream = one prover; several contexts on the
 * same GPU may be driven concurrently from different threads (independent proofs in flight). */
int lm_bind_thread(lm_ctx* ctx);
int lm_sync(lm_ctx* ctx);
/* the HIP stream every kernel of this context is launched on (hipStream_t), for event timing by the caller */
void* lm_ctx_stream(lm_ctx* ctx);

/* Per-kernel HIP-event timing on the context's stream (bench.py's roofline leg).  lm_profile_select(ctx, "k_name")
 * brackets every later launch of that kernel with a pair of events ("*" = every kernel, NULL/"" = off);
 * lm_profile_read synchronises, returns launch count and summed duration for one kernel and clears its records. */
int lm_profile_select(lm_ctx* ctx, const char* kernel_name);
int lm_profile_read(lm_ctx* ctx, const char* kernel_name, uint64_t* n_launches, double* total_ms);
/* of the launches the last lm_profile_read consumed: the time during which at least one of them was running (launches of one
 * kernel family on several streams overlap — the AIR sessions —, so this is less than their summed duration) */
double lm_profile_busy_ms(lm_ctx* ctx);
/* the ALGORITHMIC HBM bytes of the recorded launches of an HBM-bound kernel (k_prod_round2 / k_fold2_round: every f and W value
 * read once, the folded tables written once), for GB/s = bytes / HIP-event time; clears the counter.  lm_profile_select accepts
 * a comma-separated list of kernel names. */
uint64_t lm_profile_read_bytes(lm_ctx* ctx, const char* kernel_name);
/* names of the kernels with recorded launches, '\n'-separated; returns the buffer size needed (buf may be NULL) */
uint64_t lm_profile_names(lm_ctx* ctx, char* buf, uint64_t cap);
/* Host <-> device exchanges: with the log on, every wait of the prover thread for a published result (one Fiat-Shamir step: a
 * launched kernel or a message to a resident one) records its duration in microseconds.  lm_wait_log clears the log;
 * lm_wait_log_read copies up to `cap` entries and returns how many there are. */
int lm_wait_log(lm_ctx* ctx, int on);
uint64_t lm_wait_log_read(lm_ctx* ctx, float* out_us, uint64_t cap);
/* How often this context re-ran a GKR layer with one launch per exchange because a resident kernel (the tail of a layer, a launch
 * enqueued ahead of its challenges) never got its wave slots on a shared device: an internal scheduling event, not a prover error —
 * the proof is unchanged (lm_gkr_round). */
uint32_t lm_soft_fallbacks(const lm_ctx* ctx);

int lm_malloc(lm_ctx* ctx, uint64_t n_words, uint32_t** d_out);
int lm_free(lm_ctx* ctx, uint32_t* d_ptr);
int lm_upload(lm_ctx* ctx, uint32_t* d_dst, const uint32_t* src, uint64_t n_words);
/* lm_upload without the synchronisation: stream-ordered with the kernels that follow; `src` must stay valid (and should be
 * pinned host memory for full PCIe rate) until the stream has passed the copy. */
int lm_upload_async(lm_ctx* ctx, uint32_t* d_dst, const uint32_t* src, uint64_t n_words);
int lm_download(lm_ctx* ctx, uint32_t* dst, const uint32_t* d_src, uint64_t n_words);
int lm_memset_zero(lm_ctx* ctx, uint32_t* d_dst, uint64_t n_words);
/* d_cols[c][offset .. offset + count) = values[c] for c < n_cols in one launch (d_cols, values: HOST arrays; the padding rows of a
 * table's columns, lmh_pad_table). */
int lm_fill_columns(lm_ctx* ctx, uint32_t* const* d_cols, const uint32_t* values, uint32_t n_cols, uint64_t offset, uint64_t count);
int lm_ef_aos_to_soa(lm_ctx* ctx, const uint32_t* d_aos, uint32_t* d_soa, uint64_t n);
int lm_ef_soa_to_aos(lm_ctx* ctx, const uint32_t* d_soa, uint32_t* d_aos, uint64_t n);

/* ---- hash ------------------------------------------------------------------------------------------------------ */
/* n independent Poseidon1-16 permutations / compressions (perm(x)+x) of 16-word states, d_states = n x 16 words,
 * in place.  Replaces Poseidon1KoalaBear16::{permute_mut,compress_in_place}
 * (crates/backend/koala-bear/src/poseidon1_koalabear_16.rs:873-912,1018-1030). */
int lm_poseidon16_permute(lm_ctx* ctx, uint32_t* d_states, uint64_t n);
int lm_poseidon16_compress(lm_ctx* ctx, uint32_t* d_states, uint64_t n);
/* the same through the 4-lane cooperative kernel (csrc/poseidon16_quad.h: one state per DPP quad) that the proof-of-work
 * search uses; compress != 0: compression mode.  Exposed for parity tests. */
int lm_poseidon16_permute_quad(lm_ctx* ctx, uint32_t* d_states, uint64_t n, int compress);

/* fill_trace_poseidon_16 / generate_trace_rows_for_perm (crates/lean_vm/src/tables/poseidon_16/trace_gen.rs:10-165):
 * d_cols = host array of the 109 DEVICE column pointers of the Poseidon16 table (Poseidon1Cols16 order,
 * poseidon_16/mod.rs:366-383).  Columns 0..24 (flags, indices, 16 inputs) are inputs; the 84 derived columns are written. */
int lm_poseidon_trace(lm_ctx* ctx, uint32_t* const* d_cols, uint64_t n_rows);

/* The step of get_execution_trace after fill_trace_poseidon_16 (crates/lean_prover/src/trace_gen.rs:118-147): on rows with
 * flag_permute = 0 (column 8) the unconstrained output columns are overwritten with the memory words their lookup reads:
 * columns 101..108 <- memory[index_res + 8 ..], and with flag_half_output = 1 (column 3) also 97..100 <- memory[index_res + 4 ..].
 * d_cols = the same host array of 109 device column pointers as lm_poseidon_trace (index_res = column 2). */
int lm_poseidon_trace_outputs_from_memory(lm_ctx* ctx, uint32_t* const* d_cols, uint64_t n_rows, const uint32_t* d_memory,
                                          uint64_t memory_len);

/* fill_trace_extension_op (crates/lean_vm/src/tables/extension_op/exec.rs:192-203): the five VALUE_A columns of the
 * ExtensionOp table, value_a[k][row] = memory[idx_a[row] + k] (idx_a = column COL_IDX_A, Montgomery form like every
 * column).  d_va_cols = host array of the 5 DEVICE column pointers (COL_VA..COL_VA+4, extension_op/air.rs:24).  A row whose
 * address range leaves [0, memory_len) gets zeros (the reference would panic on the slice index). */
int lm_extension_op_trace(lm_ctx* ctx, const uint32_t* d_memory, uint64_t memory_len, const uint32_t* d_idx_a,
                          uint32_t* const* d_va_cols, uint64_t n_rows);

/* get_execution_trace, main loop (crates/lean_prover/src/trace_gen.rs:27-100): the 24 columns of the execution table
 * (execution/air.rs:9-37: pc, fp, addr_a/b/c, value_a/b/c, the 12 instruction columns, is_precompile, nu_a/b/c) for cycles
 * 0..n_cycles from the VM's log — d_pcs / d_fps are CANONICAL integers (the reference's Vec<usize>), everything else is in
 * Montgomery form.  d_bytecode = instructions_multilinear (bytecode_rows x 16 words, 12 used); d_memory = the padded memory
 * image (undefined cells are 0, as trace_gen.rs:103).  d_cols = host array of 24 device column pointers, n_cycles words
 * each; padding rows (pad_table) stay with the caller. */
int lm_execution_table_trace(lm_ctx* ctx, const uint32_t* d_pcs, const uint32_t* d_fps, uint64_t n_cycles,
                             const uint32_t* d_bytecode, uint64_t bytecode_rows, const uint32_t* d_memory, uint64_t memory_len,
                             uint32_t* const* d_cols);

/* The precompile tables from the VM runner's call records (lmh_execute_bytecode, leanmultisig_host.h):
 * lm_poseidon_table_from_calls: Poseidon16Precompile::execute's pushes (crates/lean_vm/src/tables/poseidon_16/mod.rs:262-286) —
 *   columns 0..8 (flag, index_b, index_res, half_output, hardcoded_left, offset, effective left indices, permute), the 16 input
 *   columns 9..24 (read from the final memory image: memory is write-once, so it holds what the call read) and the virtual bus
 *   columns 109 (index_input_left) and 110 (precompile_data).  d_calls = n_calls x 9 canonical words (LM_VM_POSEIDON_CALL_WORDS);
 *   d_cols = host array of 111 device column pointers (25..108 are left to lm_poseidon_trace).
 * lm_extension_table_from_rows: exec_multi_row's pushes (extension_op/exec.rs:149-186): every column of the ExtensionOp table except
 *   VALUE_A (lm_extension_op_trace) from n_rows x 24-word records (LM_VM_EXTENSION_ROW_WORDS); d_cols = 31 device column pointers. */
int lm_poseidon_table_from_calls(lm_ctx* ctx, const uint32_t* d_calls, uint64_t n_calls, const uint32_t* d_memory, uint64_t memory_len,
                                 uint32_t* const* d_cols);
int lm_extension_table_from_rows(lm_ctx* ctx, const uint32_t* d_rows, uint64_t n_rows, uint32_t* const* d_cols);

/* ---- WHIR commitment: LDE + Merkle tree -------------------------------------------------------------------------
 * lm_commit replaces reorder_and_dft (crates/whir/src/utils.rs:69-98: prepare_evals_for_fft_unpacked :128-150 +
 * EvalsDft::dft_algebra_batch_by_evals crates/whir/src/dft.rs:79-155) followed by MerkleData::build
 * (crates/whir/src/commit.rs:17-32 -> merkle_commit/build_merkle_tree_koalabear crates/whir/src/merkle.rs:28-88).
 *
 *   d_evals        2^n_vars evaluations of the multilinear polynomial; base words, or (is_ext) SoA EF
 *   folding_factor k: the matrix has 2^k columns, column c = c-th contiguous 2^(n_vars-k) slice
 *   log_inv_rate   each value repeated 2^log_inv_rate times before the transform; h = 2^(n_vars+log_inv_rate-k) rows
 *   actual_len     d_evals[actual_len..] is all zero (WhirConfig::commit, commit.rs:64-99); columns that are
 *                  entirely zero are neither transformed nor stored, and are absorbed through the zero-suffix sponge
 *                  state like first_digest_layer_with_initial_state (merkle.rs:251-287).  Pass 2^n_vars if unknown.
 *   root           8 words (host)
 * The LDE matrix and all digest layers stay resident in HBM inside *out until lm_tree_free. */
int lm_commit(lm_ctx* ctx, const uint32_t* d_evals, int is_ext, uint32_t n_vars, uint32_t folding_factor,
              uint32_t log_inv_rate, uint64_t actual_len, lm_tree** out, uint32_t root[LM_DIGEST_WORDS]);
void lm_tree_free(lm_ctx* ctx, lm_tree* tree);
uint32_t lm_tree_log_height(const lm_tree* tree);
/* number of base words of one (zero-padded) leaf: 2^k, or 5 * 2^k for EF */
uint32_t lm_tree_leaf_words(const lm_tree* tree);
/* Batched MerkleData::open (commit.rs:34-46 -> WhirMerkleTree::open merkle.rs:205-211 + open_siblings
 * symetric/src/merkle.rs:43-47).  leaves: n_idx x leaf_words (EF leaves flattened coefficient-minor exactly like
 * flatten_to_base); siblings: n_idx x log_height x 8, bottom-up. */
int lm_tree_open(lm_ctx* ctx, const lm_tree* tree, const uint64_t* indices, uint32_t n_idx, uint32_t* leaves,
                 uint32_t* siblings);
/* The same opening as two calls: _begin enqueues the kernel, _end waits for it and copies leaves and paths out (it consumes the
 * handle, also on failure; leaves = NULL abandons the opening).  Between the two the caller may enqueue other work on the context
 * (it is ordered behind the opening kernel): WHIR's weight kernels of a round depend on the query indices only (open.rs:337-382). */
typedef struct lm_tree_opening lm_tree_opening;
int lm_tree_open_begin(lm_ctx* ctx, const lm_tree* tree, const uint64_t* indices, uint32_t n_idx, lm_tree_opening** out);
int lm_tree_open_end(lm_ctx* ctx, lm_tree_opening* opening, uint32_t* leaves, uint32_t* siblings);
/* test/debug access: copy the device-resident LDE matrix out in the reference's row-major order (h x leaf_words) */
int lm_tree_download_matrix(lm_ctx* ctx, const lm_tree* tree, uint32_t* rows);
/* test/debug access: all digest layers bottom-up, (2h - 1) x 8 words */
int lm_tree_download_digests(lm_ctx* ctx, const lm_tree* tree, uint32_t* digests);

/* ---- multilinear evaluation --------------------------------------------------------------------------------------
 * EvaluationsList::evaluate (crates/backend/poly/src/evals.rs:26,142-347): out = sum_i v[i] * eq(point, i).
 * n_polys polynomials of 2^n_vars values each at d_evals + p * stride_words (base) — or SoA EF with plane stride
 * 2^n_vars at d_evals + p * stride_words (is_ext) — all at the same point (n_vars x 5 host words).
 * out: n_polys x 5 host words. */
int lm_mle_eval(lm_ctx* ctx, const uint32_t* d_evals, int is_ext, uint32_t n_vars, uint32_t n_polys,
                uint64_t stride_words, const uint32_t* point, uint32_t* out);
/* Evaluations that no transcript step separates (the column evaluations behind the GKR, logup.rs:224-308) need not wait for each other:
 * between _begin and _end, lm_mle_eval (pinned results) and lm_mle_eval_cols enqueue their kernels, return at once and leave `out`
 * untouched; _end waits for the last one and fills every `out`.  No other call that publishes a result may be made in between. */
int lm_results_defer_begin(lm_ctx* ctx);
int lm_results_defer_end(lm_ctx* ctx);
/* ONE polynomial at n_points points (points: n_points x n_vars x 5 words, out: n_points x 5): the OOD samples of a commitment
 * (whir/src/utils.rs:30-57) are drawn together; the device reads the polynomial once per pair of points. */
int lm_mle_eval_points(lm_ctx* ctx, const uint32_t* d_evals, int is_ext, uint32_t n_vars, uint32_t n_points, const uint32_t* points,
                       uint32_t* out);
/* Same for n_cols base columns given by a host array of DEVICE pointers (the 91 column evaluations that follow the GKR,
 * crates/sub_protocols/src/logup.rs:224-308, are batches of this form). */
int lm_mle_eval_cols(lm_ctx* ctx, const uint32_t* const* d_cols, uint32_t n_cols, uint32_t n_vars, const uint32_t* point,
                     uint32_t* out);
/* device-to-device copy of n_words (stack_polynomials_and_commit, crates/sub_protocols/src/stacked_pcs.rs:118-136) */
/* Access counters of prove_execution (crates/lean_prover/src/prove_execution.rs:90-110): d_acc[0..len) = number of times
 * each address is read, as field elements: for every job (an index column of a table lookup, canonical value = address)
 * and every row, d_acc[address + j] += 1 for j < n_values.  Rows whose address range falls outside [0, len) are ignored
 * (the reference would panic).  Used for memory_acc (all table lookups) and bytecode_acc (the pc column, n_values = 1).
 * n_jobs <= 16, n_values <= 16, len < 2^28.  Skipped rows are counted: lm_access_errors returns the number of rows outside
 * the image since the last reset (it synchronises the stream); lmh_prove_execution fails with LM_E_INVALID if there were any. */
int lm_access_counts(lm_ctx* ctx, uint32_t* d_acc, uint64_t len, uint32_t n_jobs, const uint32_t* const* d_index_cols,
                     const uint64_t* n_rows, const uint32_t* n_values);
uint32_t lm_access_errors(lm_ctx* ctx, int reset);

/* stack_polynomials (crates/sub_protocols/src/stacked_pcs.rs:99-157): d_dst[0..total_words) = zero everywhere except
 * d_dst[dst_offset[i] .. +n_words[i]) = d_src[i][0..n_words[i]).  d_src is a HOST array of device pointers; jobs must be
 * sorted by dst_offset and disjoint.  One pass over the destination (no separate zero-fill). */
int lm_stack_columns(lm_ctx* ctx, uint32_t* d_dst, uint64_t total_words, uint32_t n_jobs, const uint32_t* const* d_src,
                     const uint64_t* dst_offset, const uint64_t* n_words);
int lm_copy_d2d(lm_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, uint64_t n_words);

/* ---- weight polynomial of the WHIR sumcheck -----------------------------------------------------------------------
 * combine_statement (crates/whir/src/open.rs:518-584), SumcheckSingle::add_new_equality / add_new_base_equality
 * (open.rs:337-382 -> compute_eval_eq_packed / compute_eval_eq_base_packed_batched, crates/backend/poly/src/eq_mle.rs):
 *     W[offset_j + i] += scalar_j * w_j(i),  i < 2^inner_j,
 * with w_j = eq(point_j, .) or (is_next) matrix_next_mle_folded(point_j) (crates/backend/poly/src/next_mle.rs:35-53).
 * offset_j = selector << inner_j.  Items that share (offset, inner) are summed in registers and W is touched once.
 * d_W: SoA EF of 2^n_vars.  points: concatenated host EF coordinates, item j uses inner_j of them starting at
 * point_offset_j (in EF elements).  scalars: n_items x 5 host words. */
typedef struct {
    uint64_t offset;       /* selector << inner_n */
    uint32_t inner_n;      /* number of coordinates of the point */
    uint32_t is_next;      /* 0: eq, 1: next */
    uint64_t point_offset; /* index of the first coordinate in `points` */
} lm_weight_item;
int lm_weights_accumulate(lm_ctx* ctx, uint32_t* d_W, uint32_t n_vars, const lm_weight_item* items, uint32_t n_items,
                          const uint32_t* points, uint64_t n_point_coords, const uint32_t* scalars);
/* Same, but W is write-only: on exit W = sum of the items (uninitialised memory on entry is fine).  combine_statement's
 * first use (open.rs:518-584 starts from a zeroed weight polynomial); saves the zero-fill and one read of W. */
int lm_weights_init(lm_ctx* ctx, uint32_t* d_W, uint32_t n_vars, const lm_weight_item* items, uint32_t n_items,
                    const uint32_t* points, uint64_t n_point_coords, const uint32_t* scalars);

/* ---- product sumcheck (WHIR) --------------------------------------------------------------------------------------
 * One round of run_product_sumcheck / sumcheck_prove_many_rounds with ProductComputation
 * (crates/backend/sumcheck/src/product_computation.rs:37-315): pairs (i, i + 2^(n_vars-1)) — MSB-first —
 *     c0 = sum f[i] W[i],   c2 = sum (f[i+half] - f[i]) (W[i+half] - W[i]);   out = c0 || c2 (10 host words).
 * d_f is base words or SoA EF, d_W is SoA EF. */
int lm_prod_round(lm_ctx* ctx, const uint32_t* d_f, int f_is_ext, const uint32_t* d_W, uint32_t n_vars,
                  uint32_t out_c0_c2[10]);
/* fold_multilinear (crates/backend/poly/src/utils.rs:161-186): out[i] = in[i] + r (in[i + half] - in[i]), out is SoA
 * EF of 2^(n_vars-1) (d_out may not alias d_in). */
int lm_fold(lm_ctx* ctx, const uint32_t* d_in, int in_is_ext, uint32_t n_vars, const uint32_t r[LM_EF_DIM],
            uint32_t* d_out);
/* lm_fold of f and W by r fused with lm_prod_round of the folded tables (the next round of run_product_sumcheck,
 * product_computation.rs:37-125): d_f_out / d_W_out receive the folded SoA EF tables of n_vars - 1 variables and
 * out_c0_c2 the next round's (c0, c2).  One pass over the data instead of three.  n_vars >= 2. */
int lm_fold_round(lm_ctx* ctx, const uint32_t* d_f, int f_is_ext, const uint32_t* d_W, uint32_t n_vars,
                  const uint32_t r[LM_EF_DIM], uint32_t* d_f_out, uint32_t* d_W_out, uint32_t out_c0_c2[10]);

/* Two rounds per pass over the tables (exact identities, the transcript is unchanged; see lm_whir_ops.hip).
 * lm_prod_round2: eight sums over quads (x[i], x[i + n/4], x[i + n/2], x[i + 3n/4]) of the tables as they are, n = 2^n_vars:
 *   out = P00, P01, P10, Q0, Q1, T0, T2, T3 (8 EF).  This round: c0 = P00 + P01, c2 = Q0 + Q1.  Next round, as polynomials
 *   in this round's challenge r: c0'(r) = P00 + r (P10 - P00 - Q0) + r^2 Q0,  c2'(r) = T0 + r (T3 - T0 - T2) + r^2 T2.
 * lm_fold2_round: fold f and W by r0 then r1 (2^(n_vars-2) entries out), and on the folded tables compute
 *   sums = 2: the eight sums (40 words, n_vars >= 4), sums = 1: (c0, c2) of the next round (10 words, n_vars >= 3), 0: nothing. */
int lm_prod_round2(lm_ctx* ctx, const uint32_t* d_f, int f_is_ext, const uint32_t* d_W, uint32_t n_vars, uint32_t out_sums[40]);
int lm_fold2_round(lm_ctx* ctx, const uint32_t* d_f, int f_is_ext, const uint32_t* d_W, uint32_t n_vars, const uint32_t r0[LM_EF_DIM],
                   const uint32_t r1[LM_EF_DIM], uint32_t* d_f_out, uint32_t* d_W_out, int sums, uint32_t* out_sums);

/* ---- logup numerators / denominators -----------------------------------------------------------------------------
 * The fill loops of prove_generic_logup (crates/sub_protocols/src/logup.rs:88-199) as a list of sections, natural order:
 *     num[out_offset + i] = 0 | 1 | +col[i] | -col[i]
 *     den[out_offset + i] = c  +/-  ( sum_j alpha_eq[j] * data_j(i)  +  alpha_eq[15] * domsep )     (finger_print,
 *                                                                     crates/utils/src/multilinear.rs:76-97)
 *     data_j(i) = d_data[j][i * stride[j]] + add[j]        or, when d_data[j] == NULL,  i + add[j]  (the row index)
 * Everything outside the sections (bytecode padding, tail up to 2^n_vars) is the neutral pair (0, 1).
 * d_nums: 2^n_vars base words; d_dens: SoA EF of 2^n_vars. */
#define LM_LOGUP_MAX_DATA 13 /* max_bus_width_including_domainsep(), lean_vm/src/tables/table_enum.rs:109-111 */
typedef struct {
    uint64_t out_offset;
    uint32_t log_len;
    uint32_t num_mode;      /* 0: zero, 1: one, 2: +d_num_col, 3: -d_num_col */
    const uint32_t* d_num_col;
    int32_t den_sign;       /* +1: c + fingerprint (bus), -1: c - fingerprint (memory / bytecode lookups) */
    uint32_t domsep;        /* LOGUP_{MEMORY,PRECOMPILE,BYTECODE}_DOMAINSEP = 0, 1, 2 (lean_vm/src/core/constants.rs:4-6) */
    uint32_t n_data;
    uint32_t reserved;
    const uint32_t* d_data[LM_LOGUP_MAX_DATA];
    uint32_t stride[LM_LOGUP_MAX_DATA];
    uint32_t add[LM_LOGUP_MAX_DATA]; /* canonical small integer */
} lm_logup_section;
int lm_logup_build(lm_ctx* ctx, const lm_logup_section* sections, uint32_t n_sections, const uint32_t c[LM_EF_DIM],
                   const uint32_t* alphas_eq16, uint32_t n_vars, uint32_t* d_nums, uint32_t* d_dens);

/* Same, but only the prefix that holds sections is written (holes between sections get the neutral pair); *out_active_len =
 * end of the last section.  For lm_gkr_build_active: the tail [active_len, 2^n_vars) is never read. */
int lm_logup_build_active(lm_ctx* ctx, const lm_logup_section* sections, uint32_t n_sections, const uint32_t c[LM_EF_DIM],
                          const uint32_t* alphas_eq16, uint32_t n_vars, uint32_t* d_nums, uint32_t* d_dens, uint64_t* out_active_len);

/* ---- GKR for a sum of fractions (logup) -----------------------------------------------------------------------------
 * prove_gkr_quotient (crates/sub_protocols/src/quotient_gkr/mod.rs:31-78).  The transcript stays with the caller; the
 * device holds the layers and runs the per-round kernels.  d_nums: 2^n_vars base words, d_dens: SoA EF of 2^n_vars,
 * NATURAL index order, already padded with (0, 1) to the power of two (the reference's chunk-bit-reversed packing and
 * symbolic padding — logup.rs:61-86, sumcheck_utils.rs:136 — are CPU layout choices; transcript values are identical).
 *   lm_gkr_build        sum_quotients_2_by_2 down to 2^5 entries (layers.rs:124-189); inputs must stay alive
 *   lm_gkr_top          the 32 + 32 values sent first (mod.rs:64-66), host AoS EF
 *   lm_gkr_layer_begin  start prove_gkr_layer (mod.rs:80-141) for the layer with 2^(K+1) entries: claim point (K x 5), alpha
 *   lm_gkr_round        one sumcheck round, LSB first: out = (c0_raw, c2_raw) of finalize_round (sumcheck_utils.rs:90-109);
 *                       prev_r = NULL for the first round, else the previous challenge (fold_and_compute_round)
 *   lm_gkr_layer_end    fold by the last challenge; inner_evals = [n_l, n_r, d_l, d_r] (4 EF)
 * Between lm_gkr_layer_begin and lm_gkr_layer_end (or lm_gkr_free) the context's stream belongs to the layer: once the arrays
 * are small, a kernel stays RESIDENT on it and receives the challenges of lm_gkr_round through a pinned mailbox instead of
 * being relaunched, so any other call that launches work on the same context would queue behind it — it
 * fails after the resident kernel's 3 s timeout and the layer is lost.  Use another lm_ctx for concurrent work (the
 * reference's prove_gkr_quotient does nothing else between the rounds of a layer either).  lm_gkr_free / a new
 * lm_gkr_layer_begin dismiss a resident kernel at once.  LM_GKR_NO_TAIL=1 (environment) restores one launch per round pair.
 * Before the arrays are that small, the NEXT launch of the layer is enqueued behind the current one ahead of its two challenges and
 * receives them as a message on the device (csrc/lm_common.h: lm_mail_*); it is dismissed the same way when the layer is abandoned
 * (lm_gkr_free / lm_gkr_layer_begin) and gives up by itself after 3 s.  LM_GKR_NO_AHEAD=1 launches only when the challenges exist. */
typedef struct lm_gkr lm_gkr;
int lm_gkr_build(lm_ctx* ctx, const uint32_t* d_nums, const uint32_t* d_dens, uint32_t n_vars, lm_gkr** out);
/* Same with an ACTIVE PREFIX: entries [active_len, 2^n_vars) are the neutral pair (0, 1) and are never read (they need not
 * exist in memory beyond active_len rounded up to a multiple of 8); layers, folds and round sums only touch the prefix, the
 * all-padding part enters the round polynomials in closed form — what the reference does with its symbolic padding
 * (quotient_gkr/sumcheck_utils.rs:136,225,331).  Identical transcript. */
int lm_gkr_build_active(lm_ctx* ctx, const uint32_t* d_nums, const uint32_t* d_dens, uint32_t n_vars, uint64_t active_len, lm_gkr** out);
void lm_gkr_free(lm_ctx* ctx, lm_gkr* g);
int lm_gkr_top(lm_ctx* ctx, const lm_gkr* g, uint32_t* nums32, uint32_t* dens32);
int lm_gkr_layer_begin(lm_ctx* ctx, lm_gkr* g, uint32_t K, const uint32_t* point, const uint32_t alpha[LM_EF_DIM]);
int lm_gkr_round(lm_ctx* ctx, lm_gkr* g, const uint32_t* prev_r, uint32_t out_c0_c2[10]);
int lm_gkr_layer_end(lm_ctx* ctx, lm_gkr* g, const uint32_t last_r[LM_EF_DIM], uint32_t inner_evals[20]);

/* ---- AIR sumcheck sessions ----------------------------------------------------------------------------------------
 * The OuterSumcheckSession trait object of the reference (crates/sub_protocols/src/air_sumcheck.rs:34-42, implemented by
 * AirSumcheckSession :45-296) as an opaque handle.  table: 0 = execution, 1 = extension_op, 2 = poseidon16
 * (constraint bodies: crates/lean_vm/src/tables/{execution/air.rs, extension_op/air.rs, poseidon_16/mod.rs}).
 *   lm_air_new         AirSumcheckSession::new: d_cols = host array of n_columns DEVICE pointers to the committed base
 *                      columns (2^log_rows words each, natural row order — the chunk-bit-reversal of :87-111 is a CPU
 *                      layout); the next-row ("shift") views of the first n_shift columns are derived on the device
 *                      (compute_shifted_columns :683-694).  eq_point = eq_factor (log_rows x 5), alpha = air_alpha
 *                      (powers are taken on the device side), logup_eq16 = logup_alphas_eq_poly (16 x 5), bus_beta.
 *   lm_air_round       the raw sums of compute_bare_round_poly before missing_mul_factor / padding handling:
 *                      out[zi] = sum_pairs eq_prefix * sum_k alpha^k C_k(lo + z (hi - lo)), z = 0, 2, 3, .., degree
 *                      (degree x 5 words); the full padded domain is summed, which equals the reference's active
 *                      prefix + constraints_eval_at_padding shortcut (:236-240)
 *   lm_air_bind        process_challenge's fold (:268-292)
 *   lm_air_final_evals final_column_evals (:294-296): (n_columns + n_shift) x 5 words */
typedef struct lm_air lm_air;
int lm_air_new(lm_ctx* ctx, uint32_t table, const uint32_t* const* d_cols, uint32_t log_rows, const uint32_t* eq_point,
               const uint32_t alpha[LM_EF_DIM], const uint32_t* logup_eq16, const uint32_t bus_beta[LM_EF_DIM], lm_air** out);
void lm_air_free(lm_ctx* ctx, lm_air* a);
/* Active prefix (AirSumcheckSession: unpadded_len / constraints_eval_at_padding, air_sumcheck.rs:194-200,236-240): rows
 * [n_active_rows, 2^log_rows) of every column are the table's padding row (identical rows, as pad_table writes them).  The
 * round kernels then evaluate only the pairs that contain an active row plus ONE padding pair, weighted with the sum of the
 * eq weights of all padding pairs (the reference's padding_eq_sum) — same sums as over the full table.  Call before the
 * first round; default = all rows active. */
int lm_air_set_active_rows(lm_air* a, uint64_t n_active_rows);
uint32_t lm_air_degree(const lm_air* a);
uint32_t lm_air_n_evals(const lm_air* a);
int lm_air_round(lm_ctx* ctx, lm_air* a, uint32_t* out_raw);
/* lm_air_round in two halves: _launch enqueues the round's kernels, _wait collects the result.  The sessions of one batched
 * round (prove_batched_air_sumcheck, air_sumcheck.rs:636-681) are independent until the shared challenge: launch them all,
 * then wait — one host round trip per batched round instead of one per table.  From the end of lm_air_new on a session
 * runs on its own HIP stream (one per table, forked from the context's stream with an event; its results carry their own
 * sequence flag), so the three chains of a batched round execute side by side; lm_air_free joins it.  The caller's columns
 * must stay untouched until lm_air_free. */
int lm_air_round_launch(lm_ctx* ctx, lm_air* a);
int lm_air_round_wait(lm_ctx* ctx, lm_air* a, uint32_t* out_raw);
int lm_air_bind(lm_ctx* ctx, lm_air* a, const uint32_t challenge[LM_EF_DIM]);
int lm_air_final_evals(lm_ctx* ctx, lm_air* a, uint32_t* out);
/* lm_air_final_evals in two halves, like lm_air_round: _begin enqueues the session's publication on its own stream, _end collects it
 * (the sessions of a batch: begin all, then end all). */
int lm_air_final_evals_begin(lm_ctx* ctx, lm_air* a);
int lm_air_final_evals_end(lm_ctx* ctx, lm_air* a, uint32_t* out);

/* ---- proof-of-work ------------------------------------------------------------------------------------------------
 * FSProver::pow_grinding (crates/backend/fiat-shamir/src/prover.rs:120-177): smallest canonical w such that
 * permute(capacity[0..8] || w || 0^7)[8], read canonically, has `bits` low zero bits.  *witness is Montgomery form.
 * (The reference takes whichever witness rayon finds first; every verifier accepts the smallest.) */
int lm_pow_grind(lm_ctx* ctx, const uint32_t capacity[8], uint32_t bits, uint32_t* witness);

#ifdef __cplusplus
}
#endif
#endif /* LEANMULTISIG_H */
