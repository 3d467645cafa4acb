// Internal to the library: what the VM runner (host, lm_vm.cpp) and its device half (lm_vm_device.hip) share.  Not part of the ABI.
//
// A parallel loop batch of the leanVM (Hint::ParallelBatchStart, crates/lean_vm/src/execution/runner.rs:369-482) consists of
// independent segments — one loop iteration each, with the SegmentMemory semantics of execution/memory.rs:118-189: the memory
// below the batch is read-only, the segment's own frame is writable (write-once), every other write is deferred.  On the device
// one wavefront interprets one segment: its frame lives in LDS, the shared prefix is read from HBM, Poseidon calls run on 16
// lanes (poseidon16_coop.h), and the segment's log (pc / fp per cycle, precompile call records, deref hints, deferred writes)
// goes straight into HBM, where the trace kernels of lm_logup.hip read it — nothing but counters comes back to the host.
#pragma once
#include <stdint.h>

#include "../kb.h"

struct lm_ctx;

namespace lmh {
using kb::u32;
using kb::u64;

// one decoded instruction (36 bytes): kind, operand modes (LM_VM_ARG_*), canonical and Montgomery operand forms, precompile data
enum : uint8_t { VM_K_ADD = 0, VM_K_MUL, VM_K_DEREF, VM_K_JUMP, VM_K_POSEIDON, VM_K_EXTOP };
struct VmInstr {
    uint8_t kind, ma, mb, mc;  // LM_VM_ARG_* of the three operands (nu_a, nu_b, nu_c)
    u32 a, b, c;               // canonical: offset, or the constant
    u32 am, bm, cm;            // the constant as a Montgomery word
    u32 x0, x1;                // poseidon: flags (1 permute, 2 half_output, 4 hardcoded_left), offset; extension op: mode flags, size
};
static_assert(sizeof(VmInstr) == 36, "VmInstr layout");
struct VmHintRec {
    u32 kind;
    u32 args[4];
    uint8_t mode[4];
};

// per-segment counters written by the segment kernel (VM_SEG_WORDS words each)
enum { VM_SEG_CYC = 0, VM_SEG_POS, VM_SEG_EXT, VM_SEG_PEND, VM_SEG_DEF, VM_SEG_ADD, VM_SEG_MUL, VM_SEG_DEREF, VM_SEG_JUMP, VM_SEG_ERR, VM_SEG_ERR_PC,
       VM_SEG_ERR_AUX, VM_SEG_WORDS = 16 };
// error codes of a segment (any non-zero code sends the batch to the host runner, which reports the reference's RunnerError)
enum {
    VM_E_OK = 0, VM_E_UNDEFINED_MEMORY, VM_E_MEMORY_ALREADY_SET, VM_E_NOT_EQUAL, VM_E_DIV_BY_ZERO, VM_E_NOT_A_POINTER, VM_E_JUMP_CONDITION,
    VM_E_PC_OUT_OF_BOUNDS, VM_E_REACHED_END, VM_E_NESTED_BATCH, VM_E_HINT, VM_E_EXTENSION_OP, VM_E_DEBUG_ASSERT, VM_E_LOG_CAPACITY, VM_E_UNSUPPORTED
};

static constexpr u32 VM_UNDEF = 0xFFFFFFFFu;  // "None" (values are < p < 2^31)
// A cell of the shared prefix whose defining Poseidon call the sequential runner has recorded but not yet executed (MemBuf, lm_vm.cpp):
// on the host it is None + an owner; a segment must not treat it as None — DEREF with an unknown result, ADD / MUL solving for an
// operand and the ExtensionOp solver all TOLERATE None and would go on with a memory the reference never has (round-4 advisor
// finding).  The image carries this poison word there instead, and any read of it ends the segment (the batch moves to the host pool).
static constexpr u32 VM_PENDING = 0xFFFFFFFEu;
static constexpr u32 VM_DEV_MAX_ARGS = 16;    // call-frame arguments of a batch
static constexpr u32 VM_DEV_MAX_NAMES = 64;   // named hint streams
static constexpr u32 VM_DEV_PREFIX_CACHE = 2048;  // words of the lowest addresses every workgroup keeps in LDS
static constexpr u32 VM_DEV_MAX_STRIDE = 14000;  // frame words kept in LDS (56 KB of the CU's 160 KB: two segments per CU stay resident)

struct VmSegArgs {  // kernel argument of k_vm_segments (by value)
    // program
    const VmInstr* code;
    const u32* hint_begin;
    const VmHintRec* hints;
    u32 n_instructions, ending_pc, n_hints;
    // hint streams (ExecutionWitness): entries of name k are [name_begin[k], name_begin[k + 1]); words of entry e are data[offset[e] .. offset[e + 1])
    const u32* wit_data;
    const u64* wit_entry_offset;
    const u64* wit_name_begin;
    u64 cur_index[VM_DEV_MAX_NAMES];  // cursor of every name when segment 0 starts
    u64 per_iter[VM_DEV_MAX_NAMES];   // entries an iteration consumes per name
    u32 n_names;
    // memory: image[0 .. init_len) is what the host had when the batch started (VM_UNDEF = None); the segment frames are written
    // back to image[split_at + i * stride ..)
    u32* image;
    u64 init_len, split_at, stride, batch_fp, frame_size;
    u32 batch_pc;
    u32 prefix_cache;  // image[0 .. prefix_cache) is also kept in LDS (<= split_at, <= VM_DEV_PREFIX_CACHE)
    u32 dbg;           // builds with -DLM_VM_DEBUG only (LM_VM_DBG, timing experiments: results are wrong): 1 skip the permutation, 2 skip the call records, 4 skip the pc / fp log; else 0
    // call frames (write_call_frame, runner.rs:353-367)
    u32 return_pc_m, saved_fp_m;  // Montgomery words
    u64 start_value;
    u32 n_args;
    u32 args_m[VM_DEV_MAX_ARGS];
    // per-segment log slots
    u32 cap_cyc, cap_pos, cap_ext, cap_pend, cap_def;
    u32 *pcs, *fps, *pos, *ext, *pend, *def, *counts;
    u32* summary;  // VM_SUMMARY_WORDS header words of the batch's summary block, zeroed by the segment kernel
    const u32* coop_tab;
    kb::EF frob[5];  // images of the basis under Frobenius (extension-field inverse)
};

static constexpr u32 VM_SUMMARY_WORDS = 32;      // header of a batch's summary block (k_vm_apply_deferred / k_vm_summary), the dirty list follows
static constexpr u32 VM_RESOLVE_INFO_WORDS = 16;  // [0] anomalies, [1 + r] entries resolved in round r

// ---- launchers (lm_vm_device.hip), all on the context's stream ---------------------------------------------------------------------
int vm_dev_segments(lm_ctx* ctx, const VmSegArgs& a, u64 n_par);
// Every deferred write of the segments against the image (write once: a different value already there counts as a conflict), the
// cells outside [lo, hi) that a write defined are listed (the host mirrors them into its arena); then the exclusive prefix sums of
// the per-segment counts (d_offsets: n_par x 4 u64: cycles, Poseidon calls, extension rows, pending derefs), totals and the first
// segment error — the summary block is all the host reads back.
int vm_dev_apply_deferred(lm_ctx* ctx, const VmSegArgs& a, u64 n_par, u64 image_cap, u64 lo, u64 hi, u64* d_offsets, u32* d_summary, u32 dirty_cap);
// Trace::merge: the segment slots into contiguous arrays at base[k] + the segment's offset
int vm_dev_splice(lm_ctx* ctx, const VmSegArgs& a, u64 n_par, const u64* d_offsets, const u64 base[4], u32* pcs, u32* fps, u32* pos, u32* ext, u32* pend);
// resolve_deref_hints (runner.rs:206-236) over n entries (target, src): n_rounds rounds numbered from first_round, then the zero
// fill of what is left — which acts only if the last of these rounds resolved nothing.  d_info: VM_RESOLVE_INFO_WORDS words.
int vm_dev_resolve(lm_ctx* ctx, u32* image, u64 image_len, const u32* pend, u64 n, uint8_t* status, u32* d_info, u32 first_round, u32 n_rounds);
// host pieces -> device destinations in one launch (src == nullptr: zero fill); the sources are copied before the call returns
static constexpr u32 VM_PLACE_MAX = 16;
struct VmPart {
    u32* dst;
    const u32* src;
    u64 n_words;
};
int vm_dev_place(lm_ctx* ctx, const VmPart* parts, u32 n_parts);
int vm_dev_fill(lm_ctx* ctx, u32* d, u32 word, u64 n);
int vm_dev_download(lm_ctx* ctx, void* dst, const void* d_src, size_t bytes);  // synchronises the stream
int vm_dev_upload(lm_ctx* ctx, void* d_dst, const void* src, size_t bytes);    // asynchronous: src must stay valid until the next synchronisation
// an event behind what has been enqueued so far / wait for it (the host may then overwrite the sources of the uploads in front of it)
int vm_dev_mark(lm_ctx* ctx);
int vm_dev_wait_mark(lm_ctx* ctx);
const u32* vm_dev_coop_table(lm_ctx* ctx);
// dst[i] = src[i] == VM_UNDEF ? 0 : src[i]; optional defined mask
int vm_dev_image_export(lm_ctx* ctx, u32* dst, const u32* src, u64 n, uint8_t* defined, u64 n_total = 0, const u32* tail24 = nullptr);  // dst[n .. n_total) = tail24, then zeros
}  // namespace lmh
