// Witness generation on the device and the whole-node entry point (SURVEY.md §8(f) rank 1 + 4):
//   lmh_get_execution_trace   get_execution_trace (crates/lean_prover/src/trace_gen.rs:14-168) from the runner's ExecutionResult
//   lmh_prove_execution_vm    prove_execution (crates/lean_prover/src/prove_execution.rs:20-274): VM run, trace, proof
// The VM log crosses PCIe once (pc / fp per cycle, the memory image, 9 words per Poseidon call, 24 per ExtensionOp row); every
// table column — execution 24, Poseidon 111 (the 84 permutation columns included), ExtensionOp 31, the padding rows — is
// produced by kernels from that log and the memory image.
// (the build compiles every source as HIP: this file is host-only, the device pass sees nothing)
#if !defined(__HIP_DEVICE_COMPILE__)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "lm_host_internal.h"
#include "lm_vm_internal.h"
#include "lm_vm_device.h"

using namespace lmh;

struct lmh_vm_trace {
    lm_execution_trace view;
    std::vector<u32*> owned;            // device buffers freed with the trace
    std::vector<u32*> cols[3];          // device column pointers per table (n_total)
    std::vector<u32> public_input;
    u32 bytecode_hash[8];
};

namespace {
// ---- pinned upload sources -----------------------------------------------------------------------------------------------------
// The runner's buffers (memory arena, logs) are recycled from run to run, so they are registered with the HIP runtime ONCE and
// uploaded from by plain DMA afterwards (an upload from pageable memory pins and unpins its pages every time, inside the copy
// call).  The runner reports every free / move of a buffer (vm_set_release_hook) and the registration goes with it.
// LM_VM_NO_PIN=1 switches this off.
struct PinRegistry {
    std::mutex mu;
    std::map<void*, size_t> pinned;  // base -> registered bytes
};
PinRegistry& pins() {
    static PinRegistry* r = new PinRegistry();  // never destroyed: buffers may be released from static destructors at exit
    return *r;
}
std::atomic<bool> g_exiting{false};  // set by an atexit handler: the HIP runtime may be gone, registrations die with the process
void unpin_hook(void* base) {
    PinRegistry& r = pins();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.pinned.find(base);
    if (it == r.pinned.end()) return;
    if (!g_exiting.load(std::memory_order_acquire)) (void)hipHostUnregister(base);
    r.pinned.erase(it);
}
bool pin_enabled() {
    static const bool on = getenv("LM_VM_NO_PIN") == nullptr;
    return on;
}
}  // namespace
namespace lmh {
// make [base, base + need) registered (need <= capacity); best effort: an upload from an unregistered buffer is still correct
void vm_ensure_pinned(const VmRegion& reg, size_t need) {
    // only whole pages that belong to the buffer alone (the VM's arena is a mapping of its own, its logs are page-aligned allocations
    // of whole pages: lm_vm.cpp, UVec): never a chunk of the malloc heap that shares a page with something else
    if (!pin_enabled() || !reg.base || need == 0 || need > reg.bytes || (reinterpret_cast<uintptr_t>(reg.base) & 4095) || (reg.bytes & 4095)) return;
    static std::once_flag hook_once;
    std::call_once(hook_once, [] {
        vm_set_release_hook(unpin_hook);
        std::atexit([] { g_exiting.store(true, std::memory_order_release); });
    });
    const size_t gran = 4u << 20;
    size_t want = (need + gran - 1) / gran * gran;
    if (want > reg.bytes) want = reg.bytes;
    PinRegistry& r = pins();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.pinned.find(reg.base);
    if (it != r.pinned.end()) {
        if (it->second >= need) return;
        (void)hipHostUnregister(reg.base);
        r.pinned.erase(it);
    }
    if (hipHostRegister(reg.base, want, hipHostRegisterDefault) == hipSuccess)
        r.pinned[reg.base] = want;
    else
        (void)hipGetLastError();
}
}  // namespace lmh
namespace {
void ensure_pinned(const VmRegion& reg, size_t need) { vm_ensure_pinned(reg, need); }

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" {

int lmh_get_execution_trace(lm_ctx* ctx, const lmh_bytecode* bc, const lmh_execution* e, const uint32_t* public_input, uint32_t n_public_input,
                            uint32_t log_inv_rate, lmh_vm_trace** out) {
    if (!ctx || !bc || !e || !out || (n_public_input && !public_input)) {
        lm_set_error("lmh_get_execution_trace: bad arguments");
        return LM_E_INVALID;
    }
    *out = nullptr;
    lm_vm_execution_view v;
    VmDeviceView dv;
    const bool resident = vm_execution_device(e, &dv);  // the run's batches executed on the device: log and image are already in HBM
    if (resident) {
        if (dv.ctx != ctx) {
            lm_set_error("lmh_get_execution_trace: the execution is resident on another context");
            return LM_E_INVALID;
        }
        memset(&v, 0, sizeof v);
        v.n_cycles = dv.n_cycles, v.memory_len = dv.memory_len, v.n_poseidon_calls = dv.n_poseidon_calls, v.n_extension_rows = dv.n_extension_rows;
        v.public_memory_size = dv.public_memory_size;
    } else
        lmh_execution_view(e, &v);
    if (!resident) {
        VmRegion reg[5];
        vm_execution_regions(e, reg);
        // the arena always (a private mapping); a log only when it is a mapping of its own too (allocations of >= 32 MB never come
        // from the malloc heap: M_MMAP_THRESHOLD's upper bound) — registered chunks of the heap are not safe, see vm_ensure_pinned
        const size_t own_mapping = 32u << 20;
        ensure_pinned(reg[0], (size_t)(v.memory_len + 24) * 4);
        for (int k = 1; k < 5; k++)
            if (reg[k].bytes >= own_mapping)
                ensure_pinned(reg[k], k < 3 ? (size_t)v.n_cycles * 4
                                            : (k == 3 ? (size_t)v.n_poseidon_calls * LM_VM_POSEIDON_CALL_WORDS * 4 : (size_t)v.n_extension_rows * LM_VM_EXTENSION_ROW_WORDS * 4));
    }
    lmh_vm_trace* t = new lmh_vm_trace();
    memset(&t->view, 0, sizeof t->view);
    int rc = LM_OK;
    // LM_VM_TIMES=1: wall clock of the steps below (each mark synchronises the stream: the total is pessimistic)
    const bool clk = getenv("LM_VM_TIMES") != nullptr;
    double t_mark = now_ms();
    auto mark = [&](const char* what) {
        if (!clk) return;
        (void)lm_sync(ctx);
        const double t1 = now_ms();
        fprintf(stderr, "#   trace: %-28s %.3f ms\n", what, t1 - t_mark);
        t_mark = t1;
    };
    mark("pin");
    auto fail = [&](int code) {
        lmh_vm_trace_free(ctx, t);
        return code;
    };
    auto dev = [&](u64 n_words, u32** p) {
        rc = lm_malloc(ctx, n_words, p);
        if (rc == LM_OK) t->owned.push_back(*p);
        return rc == LM_OK;
    };
    const u32 log_bytecode = lmh_bytecode_log_size(bc), ending_pc = lmh_bytecode_ending_pc(bc);
    // ---- memory_padded (trace_gen.rs:101-113) grown as prove_execution.rs:41-46 does -----------------------------------------------
    const u64 L = v.memory_len, zero_vec_ptr = L, null_hash_ptr = L + 16;
    u64 padded = 1ull << MIN_LOG_N_ROWS_PER_TABLE;
    while (padded < L + 24 || padded < v.n_cycles) padded <<= 1;
    while (padded < (1ull << MIN_LOG_MEMORY_SIZE) || padded < (1ull << log_bytecode)) padded <<= 1;
    if (padded > (1ull << MAX_LOG_MEMORY_SIZE)) {
        lm_set_error("lmh_get_execution_trace: memory of %llu words exceeds 2^%u", (unsigned long long)padded, MAX_LOG_MEMORY_SIZE);
        return fail(LM_E_INVALID);
    }
    const u32 log_memory = log2_ceil_u64(padded);
    // ---- the committed layout (stack_polynomials, stacked_pcs.rs:118-136): ONE buffer; memory, the access-counter slots and every
    // committed column are slices of it, so the prover commits it without the 2^n_vars-word copy (lm_execution_trace::d_stacked) ----
    const u64 n_rows[3] = {v.n_cycles, v.n_extension_rows, v.n_poseidon_calls};
    u32 log_rows[3];
    for (int tb = 0; tb < 3; tb++) {
        log_rows[tb] = lmh_table_log_rows(n_rows[tb]);
        if (log_rows[tb] > max_log_n_rows_per_table(tb)) {
            lm_set_error("TooBigTableError: table %d has 2^%u rows (limit 2^%u)", tb, log_rows[tb], max_log_n_rows_per_table(tb));
            return fail(LM_E_INVALID);
        }
    }
    if (log_rows[0] < log_rows[1] || log_rows[0] < log_rows[2] || log_memory < log_rows[0]) {
        lm_set_error("lmh_get_execution_trace: the execution table must be the tallest table and the memory at least as tall (stacked_pcs.rs:108-112)");
        return fail(LM_E_INVALID);
    }
    int order[3];
    sorted_tables(log_rows, order);
    u64 col_base[3];
    u64 total = 2 * padded + std::max(1ull << log_rows[order[0]], 1ull << log_bytecode);
    const u64 hole_from = 2 * padded + (1ull << log_bytecode), hole_to = total;
    for (int k = 0; k < 3; k++) {
        col_base[order[k]] = total;
        total += (u64)kVmTables[order[k]].n_columns << log_rows[order[k]];
    }
    const u32 stacked_n_vars = log2_ceil_u64(total);
    u32* d_stacked;
    if (!dev(1ull << stacked_n_vars, &d_stacked)) return fail(rc);
    if ((rc = lm_memset_zero(ctx, d_stacked + hole_from, hole_to - hole_from)) || (rc = lm_memset_zero(ctx, d_stacked + total, (1ull << stacked_n_vars) - total)))
        return fail(rc);
    u32* d_memory = d_stacked;
    if (L + 24 > (1ull << MAX_LOG_MEMORY_SIZE)) {
        lm_set_error("lmh_get_execution_trace: no room for the zero vector behind the memory");
        return fail(LM_E_INVALID);
    }
    if (resident) {  // None -> 0 on the way into the committed buffer; [0 x 16 | poseidon16(0)] behind it (trace_gen.rs:106-110)
        static const struct Tail {
            alignas(64) u32 w[24];
            Tail() {
                alignas(64) u32 st[16];
                memset(st, 0, sizeof st);
                memset(w, 0, sizeof w);
                host_compress(st);
                memcpy(w + 16, st, 32);
            }
        } tail;
        if ((rc = vm_dev_image_export(ctx, d_memory, dv.image, L, nullptr, padded, tail.w))) return fail(rc);  // (image | tail | zeros: one launch)
    } else if ((rc = lm_upload_async(ctx, d_memory, v.memory, L + 24)) || (rc = lm_memset_zero(ctx, d_memory + L + 24, padded - L - 24)))
        return fail(rc);  // image + [0 x 16 | poseidon16(0)] (written by the runner)
    mark("alloc + memory upload");
    // ---- bytecode table: device copy cached in the CONTEXT under the bytecode's unique id --------------------------------------
    u32* d_bytecode = (u32*)lm_ctx_cache_get(ctx, vm_bytecode_uid(bc) << 4);
    if (!d_bytecode) {
        if ((rc = lm_malloc(ctx, 16ull << log_bytecode, &d_bytecode))) return fail(rc);  // lives as long as the context's pool
        if ((rc = lm_upload(ctx, d_bytecode, lmh_bytecode_multilinear(bc), 16ull << log_bytecode))) return fail(rc);
        lm_ctx_cache_put(ctx, vm_bytecode_uid(bc) << 4, d_bytecode);
    }
    // ---- tables ----------------------------------------------------------------------------------------------------------------
    for (int tb = 0; tb < 3; tb++) {
        const u32 n_total = kVmTables[tb].n_total, n_com = kVmTables[tb].n_columns;
        t->cols[tb].resize(n_total);
        for (u32 c = 0; c < n_com; c++) t->cols[tb][c] = d_stacked + col_base[tb] + ((u64)c << log_rows[tb]);
        for (u32 c = n_com; c < n_total; c++)  // the virtual bus columns are not committed
            if (!dev(1ull << log_rows[tb], &t->cols[tb][c])) return fail(rc);
    }
    // execution table (trace_gen.rs:27-100) + its padding row (execution/mod.rs:59-74, the 4 temporary columns included)
    {
        u32 *d_pcs = const_cast<u32*>(dv.pcs), *d_fps = const_cast<u32*>(dv.fps);
        if (!resident) {
            if (!dev(v.n_cycles, &d_pcs) || !dev(v.n_cycles, &d_fps)) return fail(rc);
            if ((rc = lm_upload_async(ctx, d_pcs, v.pcs, v.n_cycles)) || (rc = lm_upload_async(ctx, d_fps, v.fps, v.n_cycles))) return fail(rc);
        }
        if ((rc = lm_execution_table_trace(ctx, d_pcs, d_fps, v.n_cycles, d_bytecode, 1ull << log_bytecode, d_memory, padded,
                                           t->cols[0].data())))
            return fail(rc);
        if ((rc = pad_table(ctx, 0, t->cols[0].data(), v.n_cycles, log_rows[0], (u32)zero_vec_ptr, (u32)null_hash_ptr, ending_pc, true))) return fail(rc);
    }
    mark("execution table");
    // Poseidon16 table: call records -> flag / index / input columns, padding rows, then the permutation columns of every row
    {
        const u64 n = v.n_poseidon_calls, rows = 1ull << log_rows[2];
        u32* d_calls = resident ? const_cast<u32*>(dv.poseidon_calls) : nullptr;
        if (n && !resident) {
            if (!dev(n * LM_VM_POSEIDON_CALL_WORDS, &d_calls)) return fail(rc);
            if ((rc = lm_upload_async(ctx, d_calls, v.poseidon_calls, n * LM_VM_POSEIDON_CALL_WORDS))) return fail(rc);
        }
        if ((rc = lm_poseidon_table_from_calls(ctx, d_calls, n, d_memory, padded, t->cols[2].data()))) return fail(rc);
        if ((rc = pad_table(ctx, 2, t->cols[2].data(), n, log_rows[2], (u32)zero_vec_ptr, (u32)null_hash_ptr, ending_pc, true))) return fail(rc);
        if ((rc = lm_poseidon_trace(ctx, t->cols[2].data(), rows))) return fail(rc);
        if ((rc = lm_poseidon_trace_outputs_from_memory(ctx, t->cols[2].data(), n, d_memory, padded))) return fail(rc);
    }
    mark("poseidon table");
    // ExtensionOp table
    {
        const u64 n = v.n_extension_rows, rows = 1ull << log_rows[1];
        u32* d_rows = resident ? const_cast<u32*>(dv.extension_rows) : nullptr;
        if (n && !resident) {
            if (!dev(n * LM_VM_EXTENSION_ROW_WORDS, &d_rows)) return fail(rc);
            if ((rc = lm_upload_async(ctx, d_rows, v.extension_rows, n * LM_VM_EXTENSION_ROW_WORDS))) return fail(rc);
        }
        if ((rc = lm_extension_table_from_rows(ctx, d_rows, n, t->cols[1].data()))) return fail(rc);
        if ((rc = lm_extension_op_trace(ctx, d_memory, padded, t->cols[1][6], t->cols[1].data() + 14, n))) return fail(rc);
        if ((rc = pad_table(ctx, 1, t->cols[1].data(), n, log_rows[1], (u32)zero_vec_ptr, (u32)null_hash_ptr, ending_pc, true))) return fail(rc);
    }
    mark("extension table");
    // the uploads above read the runner's host buffers asynchronously: they are consumed before this returns
    if ((rc = lm_sync(ctx))) return fail(rc);
    t->public_input.assign(public_input, public_input + n_public_input);
    lmh_bytecode_hash(bc, t->bytecode_hash);
    lm_execution_trace& w = t->view;
    w.log_inv_rate = log_inv_rate, w.log_memory = log_memory, w.log_bytecode = log_bytecode, w.ending_pc = ending_pc;
    w.public_memory_size = (u32)v.public_memory_size, w.n_public_input = n_public_input;
    w.public_input = t->public_input.data(), w.bytecode_hash = t->bytecode_hash;
    w.d_bytecode = d_bytecode, w.d_bytecode_acc = nullptr, w.d_memory = d_memory, w.d_memory_acc = nullptr;
    w.d_stacked = d_stacked;
    for (int tb = 0; tb < 3; tb++) {
        w.tables[tb].log_rows = log_rows[tb];
        w.tables[tb].non_padded_n_rows = (u32)n_rows[tb];
        w.tables[tb].d_cols = t->cols[tb].data();
    }
    *out = t;
    return LM_OK;
}
const lm_execution_trace* lmh_vm_trace_view(const lmh_vm_trace* t) { return &t->view; }
void lmh_vm_trace_free(lm_ctx* ctx, lmh_vm_trace* t) {
    if (!t) return;
    for (u32* p : t->owned) lm_free(ctx, p);
    delete t;
}

int lmh_prove_execution_vm(lm_ctx* ctx, lmh_prover* p, const lmh_bytecode* bc, const uint32_t* public_input, uint32_t n_public_input,
                           const lm_vm_witness* witness, const lm_whir_builder* builder, uint32_t n_threads, double times_ms[3]) {
    return lmh_prove_execution_vm_info(ctx, p, bc, public_input, n_public_input, witness, builder, n_threads, times_ms, nullptr);
}
int lmh_prove_execution_vm_info(lm_ctx* ctx, lmh_prover* p, const lmh_bytecode* bc, const uint32_t* public_input, uint32_t n_public_input,
                                const lm_vm_witness* witness, const lm_whir_builder* builder, uint32_t n_threads, double times_ms[3],
                                lm_vm_run_info* info) {
    if (info) memset(info, 0, sizeof *info);
    if (!ctx || !p || !bc || !builder) {
        lm_set_error("lmh_prove_execution_vm: bad arguments");
        return LM_E_INVALID;
    }
    if (!rate_ok(builder->starting_log_inv_rate)) {
        lm_set_error("lmh_prove_execution_vm: log_inv_rate outside [1, 4] (check_rate)");
        return LM_E_INVALID;
    }
    const double t0 = now_ms();
    lmh_execution* ex = nullptr;
    int rc = lmh_execute_bytecode_device(ctx, bc, public_input, n_public_input, witness, n_threads, &ex);  // (LM_VM_HOST=1: all on the host pool)
    if (rc) return rc;
    if (info) lmh_execution_info(ex, info);
    const double t1 = now_ms();
    lmh_vm_trace* tr = nullptr;
    rc = lmh_get_execution_trace(ctx, bc, ex, public_input, n_public_input, builder->starting_log_inv_rate, &tr);
    lmh_execution_free(ex);
    if (rc) return rc;
    const double t2 = now_ms();
    lm_whir_config cfg;
    rc = lmh_whir_config_new(builder, lmh_stacked_n_vars(&tr->view), &cfg);
    if (rc == LM_OK) rc = lmh_prove_execution(ctx, p, &tr->view, &cfg);
    lmh_vm_trace_free(ctx, tr);
    const double t3 = now_ms();
    if (times_ms) times_ms[0] = t1 - t0, times_ms[1] = t2 - t1, times_ms[2] = t3 - t2;
    return rc;
}

}  // extern "C"
#endif
