// Internal to the host layer: what lm_node.cpp needs from the VM runner (lm_vm.cpp).  Not part of the ABI.
#pragma once
#include <cstddef>
#include <functional>

#include "lm_host_internal.h"

namespace lmh {
// f(i) for i < n on the persistent host thread pool (the calling thread takes part); n_threads = 0: all hardware threads, <= 128
void vm_parallel_for(u64 n, u32 n_threads, const std::function<void(u64)>& f);
// process-unique id of a bytecode object: the key of its device copies in a context's cache (lm_ctx_cache_get / _put)
u64 vm_bytecode_uid(const lmh_bytecode* bc);
// The host buffers of an execution the device uploads from — [0] the memory arena, [1] pcs, [2] fps, [3] Poseidon call records,
// [4] ExtensionOp rows — with their CAPACITY in bytes (they are recycled from run to run, so a buffer pinned once stays useful).
struct VmRegion {
    void* base;
    size_t bytes;
};
void vm_execution_regions(const lmh_execution* e, VmRegion out[5]);
// make [base, base + need) registered with the HIP runtime (need <= capacity; best effort — lm_node.cpp): uploads from a
// registered buffer are plain DMA
void vm_ensure_pinned(const VmRegion& reg, size_t need);
// a run whose parallel batches executed on the device (lmh_execute_bytecode_device): the complete log and the memory image
// (VM_UNDEF = None, lm_vm_device.h) are resident on `ctx`
struct VmDeviceView {
    lm_ctx* ctx;
    const u32* image;
    u64 memory_len;
    const u32 *pcs, *fps;
    u64 n_cycles;
    const u32* poseidon_calls;
    u64 n_poseidon_calls;
    const u32* extension_rows;
    u64 n_extension_rows;
    u64 public_memory_size;
};
bool vm_execution_device(const lmh_execution* e, VmDeviceView* out);
// Words of a run's inputs that are still being computed when the run starts (lmh_aggregate_type_1: the hash of the public keys inside
// the `input_data` hint and the public input, a 0.65 ms hash chain that nothing in the program reads before its last instructions).
// The sequential runner treats them like the outputs of a deferred Poseidon call (MemBuf): the cells they are written to stay None
// with an owner, whatever depends on them is deferred behind them, and the first real access — normally the drain under the device
// batch — calls wait().  A run that does not defer (LM_VM_LAZY=0, no device, the repeated run after an error) waits before it starts.
struct VmLate {
    u32 n_ranges = 0;
    u64 first_word[4] = {0, 0, 0, 0};  // index into the witness's data words (a range lies inside one hint entry)
    u32 n_words[4] = {0, 0, 0, 0};     // 1 .. 64
    bool public_input = false;         // the public-input buffer handed to the run is late as a whole
    std::function<void()> wait;        // returns once every late word holds its final value (idempotent, called on the run's thread)
};
// applies to the NEXT run started on the calling thread (lmh_execute_bytecode{,_device}); cleared by that run
void vm_set_late(const VmLate* late);
// called with a buffer's base address right before the runner frees or moves it (lm_node.cpp unpins it there)
void vm_set_release_hook(void (*hook)(void* base));
}  // namespace lmh
