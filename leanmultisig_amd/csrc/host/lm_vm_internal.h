// Internal to the host layer: what lm_node.cpp needs from the VM runner (lm_vm.cpp).  Not part of the ABI.
#pragma once
#include <functional>

#include "lm_host_internal.h"

namespace lmh {
// f(i) for i < n on the persistent host thread pool (the calling thread takes part); n_threads = 0: all hardware threads, <= 128
void vm_parallel_for(u64 n, u32 n_threads, const std::function<void(u64)>& f);
// device copy of a bytecode's instructions_multilinear for context `ctx` (cached in the bytecode object, one per context):
// *slot is nullptr until lm_node.cpp fills it
u32** vm_bytecode_device_slot(const lmh_bytecode* bc, void* ctx);
}  // namespace lmh
