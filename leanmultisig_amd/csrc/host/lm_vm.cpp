// leanVM on the host: Bytecode object, the runner and its thread pool.
//
// Reference: crates/lean_vm/src/execution/runner.rs (execute_bytecode_helper, run_loop, handle_parallel_batch,
// resolve_deref_hints), execution/memory.rs (Memory, SegmentMemory), isa/instruction.rs (execute_instruction), isa/hint.rs
// (execute_hint, CustomHint::execute), tables/poseidon_16/mod.rs:209-289 and tables/extension_op/exec.rs (precompile
// execution).  Written for this machine rather than transcribed:
//   * instructions are decoded once from instructions_multilinear (the reference keeps an enum per pc next to it) into a flat
//     32-byte record with canonical AND Montgomery operand forms, hints in one array indexed by [begin, end) per pc;
//   * memory is a u32 array with a sentinel for "undefined" (the reference: Vec<Option<F>>);
//   * a parallel loop batch runs its segments on a persistent pool of host threads (the reference: rayon), every segment
//     logging into its own buffers which are spliced in iteration order afterwards; the precompile tables are kept as compact
//     call records (9 / 24 words) — the 109 / 31 table columns are built on the device from them and from the final memory
//     image (lm_node.cpp), so only ~30 MB cross PCIe for a 1550-signature run;
//   * Poseidon is the AVX-512 permutation of lm_poseidon_x86.cpp.
// (the build compiles every source as HIP: this file is host-only, the device pass sees nothing)
#if !defined(__HIP_DEVICE_COMPILE__)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "lm_host_internal.h"
#include "lm_vm_internal.h"
#include "lm_vm_device.h"

namespace lmh {
static thread_local const VmLate* g_vm_late = nullptr;
static thread_local unsigned long long g_vm_prof[4] = {0, 0, 0, 0};  // rdtsc ticks / calls: Poseidon16, ExtensionOp (LM_VM_TIMES)
static bool g_vm_prof_on = getenv("LM_VM_TIMES") != nullptr;
static void vm_prof_report() {
    static const double ticks_per_ms = [] {
        const auto t0 = std::chrono::steady_clock::now();
        const unsigned long long c0 = __builtin_ia32_rdtsc();
        while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
        }
        return (double)(__builtin_ia32_rdtsc() - c0) / std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }();
    fprintf(stderr, "[vm] this thread's precompile calls: Poseidon16 %llu in %.3f ms, ExtensionOp %llu in %.3f ms\n", g_vm_prof[1], g_vm_prof[0] / ticks_per_ms,
            g_vm_prof[3], g_vm_prof[2] / ticks_per_ms);
    g_vm_prof[0] = g_vm_prof[1] = g_vm_prof[2] = g_vm_prof[3] = 0;
}  // vm_set_late: inputs of the next run on this thread that are not final yet

// ---------------------------------------------------------------------------------------------------------------------
// thread pool: parallel_for(n, f) runs f(i) for i < n on the calling thread + the workers, dynamic scheduling
// ---------------------------------------------------------------------------------------------------------------------
class Pool {
   public:
    Pool() {}
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // the segment logs of this pool's worker threads (registered by the worker on its first segment) and the deferred-write list
    // of the batch in progress: a pool serves ONE run at a time (PoolSet), so they belong to that run's batch
    std::vector<void*> worker_logs;  // ThreadLog* (defined with the batch code below)
    std::mutex logs_mu;
    void* deferred = nullptr;  // UVec<std::pair<u64, u32>>*, owned (created on first use by handle_parallel_batch)
    // Inside a session a worker that has finished a job keeps polling for the next one for ~100 us before it goes back to sleep on
    // the condition variable: a VM run issues its parallel_for's in bursts (resize -> segments -> merge, resolve -> mask) a few
    // microseconds apart, and waking 127 sleeping threads costs ~0.7 ms each time on the 2 x 64-core host.  The poll is BOUNDED:
    // the GPU box runs under a CPU quota (cgroup cpu.max = 16 CPUs per 100 ms), and 127 threads spinning through the sequential
    // parts of a run exhausted it — the whole process, prover thread included, was throttled for 20-60 ms every few proofs.
    static u32 spin_limit() {  // x pause (~25 ns); LM_VM_SPIN overrides (A/B measurements)
        static const u32 v = [] {
            const char* e = getenv("LM_VM_SPIN");
            return e ? (u32)strtoul(e, nullptr, 10) : 4000u;
        }();
        return v;
    }
    void begin_session(u32 n_threads) {
        std::lock_guard<std::mutex> user(user_mu_);
        ensure(resolve(n_threads) - 1);
        spin_.fetch_add(1, std::memory_order_release);
    }
    void end_session() { spin_.fetch_sub(1, std::memory_order_release); }
    void parallel_for(u64 n, u32 n_threads, const std::function<void(u64)>& f) {
        if (n == 0) return;
        u32 want = resolve(n_threads);
        if (want > n) want = (u32)n;
        if (want <= 1) {
            for (u64 i = 0; i < n; i++) f(i);
            return;
        }
        std::lock_guard<std::mutex> user(user_mu_);  // one batch at a time
        ensure(want - 1);
        job_ = &f;
        total_ = n;
        next_.store(0, std::memory_order_relaxed);
        pending_.store(want - 1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu_);  // (orders the generation change with sleepers that are about to wait)
            // generation and the number of workers that take part travel in ONE word: a worker that is not part of generation g
            // may read it late, when g + 1 (with more participants) is already being set up — it must not pick up g + 1's count
            // with g's number and run (and check out) twice
            gen_.store(((gen_.load(std::memory_order_relaxed) >> 16) + 1) << 16 | (want - 1), std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire)) cv_.notify_all();
        work();
        while (pending_.load(std::memory_order_acquire) != 0) cpu_relax();
        job_ = nullptr;
    }
    size_t n_workers() const { return workers_.size(); }  // (read by PoolSet while the pool is free: nobody is growing it)
    static u32 workers_for(u32 n_threads) { return resolve(n_threads) - 1; }
    static Pool*& tl_worker_pool() {  // the pool a worker thread belongs to (nullptr on caller threads)
        static thread_local Pool* p = nullptr;
        return p;
    }
    // CPUs the cgroup grants this process (cpu.max = "<quota> <period>"), 0 = unlimited / unknown
    static u32 cgroup_cpus() {
        static const u32 v = [] {
            FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
            if (!f) return 0u;
            char q[32];
            unsigned long long per = 0;
            u32 r = 0;
            if (fscanf(f, "%31s %llu", q, &per) == 2 && strcmp(q, "max") != 0 && per) {
                const unsigned long long quota = strtoull(q, nullptr, 10);
                r = (u32)std::max<unsigned long long>(1, quota / per);
            }
            fclose(f);
            return r;
        }();
        return v;
    }
    // Default width of a run: hardware threads, at most 128 (2 x 64-core EPYC host of the GPU box: 64 threads 2.9 ms, 128 threads
    // 1.7 ms for the 1549 segments), and at most 4 x the cgroup's CPU quota: a run is a burst of a few milliseconds, which the
    // quota (CPU time per 100 ms period) tolerates well above its average, but 128 threads every 30 ms on a 16-CPU quota is more
    // CPU time than the period holds and the whole process is throttled (round 3: VM 4.5 ms alone, 7.4 ms in steady state).
    // LM_VM_THREADS overrides.
    static u32 default_threads() {
        static const u32 v = [] {
            if (const char* e = getenv("LM_VM_THREADS")) {
                const u32 x = (u32)strtoul(e, nullptr, 10);
                if (x) return x;
            }
            u32 hw = std::thread::hardware_concurrency();
            if (hw == 0) hw = 1;
            if (hw > 128) hw = 128;
            const u32 q = cgroup_cpus();
            if (q && 4 * q < hw) hw = 4 * q;
            return hw;
        }();
        return v;
    }

   private:
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    static u32 resolve(u32 n_threads) {
        u32 hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 1;
        const u32 want = n_threads ? n_threads : default_threads();
        return want > hw ? hw : want;  // never more threads than hardware threads: spinning workers must not compete for cores
    }
    void ensure(u32 n) {
        while (workers_.size() < n) {
            const u32 id = (u32)workers_.size();
            workers_.emplace_back([this, id] { loop(id); });
        }
    }
    void work() {
        const std::function<void(u64)>& f = *job_;
        for (;;) {
            const u64 i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= total_) break;
            f(i);
        }
    }
    void loop(u32 id) {
        tl_worker_pool() = this;
        u64 seen = 0;
        for (;;) {
            u64 g;
            u32 spins = 0;
            for (;;) {
                g = gen_.load(std::memory_order_acquire);
                if (g != seen) break;
                if (stop_.load(std::memory_order_acquire)) return;
                if (spin_.load(std::memory_order_acquire) && spins < spin_limit()) {  // bounded: see begin_session
                    spins++;
                    cpu_relax();
                    continue;
                }
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1, std::memory_order_relaxed);
                cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen || stop_.load(std::memory_order_acquire); });
                sleepers_.fetch_sub(1, std::memory_order_relaxed);
            }
            seen = g;
            if (id < (u32)(g & 0xffff)) {  // (job_ and total_ were written before the word was released and stay until this worker checks out)
                work();
                pending_.fetch_sub(1, std::memory_order_release);
            }
        }
    }
    std::mutex mu_, user_mu_;
    std::condition_variable cv_;
    std::vector<std::thread> workers_;
    const std::function<void(u64)>* job_ = nullptr;
    std::atomic<u64> next_{0}, gen_{0};
    std::atomic<u32> pending_{0}, spin_{0}, sleepers_{0};
    std::atomic<bool> stop_{false};
    u64 total_ = 0;
};

// Pools are leased to ONE run at a time: several provers of a process (leaves in flight) run their VMs side by side, each on its
// own pool — with a single shared pool their parallel batches took turns, and ten leaves waited for each other's segments.
// A caller that asks for n threads gets a pool with n - 1 workers (grown on demand); pools are kept for the life of the process.
class PoolSet {
   public:
    static PoolSet& get() {
        static PoolSet* s = new PoolSet();  // never destroyed (worker threads live as long as the process)
        return *s;
    }
    // `workers`: what the run will use.  Best fit: the free pool with the fewest workers that has enough of them (every job wakes
    // all sleeping workers of its pool, so a run on 8 threads should not sit on the 127-worker pool of an earlier run); else a new
    // pool; else (MAX_POOLS reached) the largest free one, grown on demand; else wait.
    Pool* acquire(u32 workers) {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            int best = -1, largest = -1;
            for (size_t i = 0; i < pools_.size(); i++) {
                if (busy_[i]) continue;
                const size_t n = pools_[i]->n_workers();
                if (n >= workers && (best < 0 || n < pools_[best]->n_workers())) best = (int)i;
                if (largest < 0 || n > pools_[largest]->n_workers()) largest = (int)i;
            }
            if (best < 0 && pools_.size() < MAX_POOLS) {
                pools_.push_back(new Pool());
                busy_.push_back(true);
                return pools_.back();
            }
            if (best < 0) best = largest;
            if (best >= 0) {
                busy_[best] = true;
                return pools_[best];
            }
            cv_.wait(lk);
        }
    }
    void release(Pool* p) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < pools_.size(); i++)
                if (pools_[i] == p) busy_[i] = false;
        }
        cv_.notify_one();
    }

   private:
    static constexpr size_t MAX_POOLS = 64;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Pool*> pools_;
    std::vector<bool> busy_;
};
Pool*& tl_run_pool() {  // the pool leased by the run in progress on this (caller) thread
    static thread_local Pool* p = nullptr;
    return p;
}
void vm_parallel_for(u64 n, u32 n_threads, const std::function<void(u64)>& f) {
    if (Pool* p = tl_run_pool()) {
        p->parallel_for(n, n_threads, f);
        return;
    }
    Pool* p = PoolSet::get().acquire(Pool::workers_for(n_threads));  // outside a run (lmh_poseidon16_compress_many): a lease for this one job
    p->parallel_for(n, n_threads, f);
    PoolSet::get().release(p);
}
struct PoolSession {  // a run's lease of a pool; LM_VM_NO_SPIN=1: its workers sleep between the jobs (for A/B measurements)
    bool spin;
    Pool* pool;
    explicit PoolSession(u32 n_threads) : spin(getenv("LM_VM_NO_SPIN") == nullptr), pool(PoolSet::get().acquire(Pool::workers_for(n_threads))) {
        tl_run_pool() = pool;
        if (spin) pool->begin_session(n_threads);
    }
    ~PoolSession() {
        if (spin) pool->end_session();
        tl_run_pool() = nullptr;
        PoolSet::get().release(pool);
    }
    PoolSession(const PoolSession&) = delete;
    PoolSession& operator=(const PoolSession&) = delete;
};

namespace {
typedef uint8_t u8;
inline double vm_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline bool vm_times() {
    static const bool on = getenv("LM_VM_TIMES") != nullptr;
    return on;
}

// lm_node.cpp pins the buffers it uploads from; it must hear about every free / move (vm_set_release_hook)
std::atomic<void (*)(void*)> g_release_hook{nullptr};
inline void notify_release(void* base) {
    if (!base) return;
    if (auto h = g_release_hook.load(std::memory_order_acquire)) h(base);
}

constexpr u32 UNDEF = 0xFFFFFFFFu;  // not a field element (values are < p < 2^31)
constexpr u64 MAX_MEMORY = 1ull << MAX_LOG_MEMORY_SIZE;

// The VM memory: 2^26 words of address space reserved up front (untouched pages cost nothing), so that growing never moves
// the image and the segments of a parallel batch can fill and first-touch their own frames concurrently.
template <class T>
struct UVec;
// extension_op/exec.rs: one element of an ExtensionOp (mode flags of the instruction: 8 add, 16 mul, 32 poly_eq)
enum { VM_OP_ADD = 8, VM_OP_MUL = 16, VM_OP_POLY_EQ = 32 };
inline EF vm_compute_elem(const EF& a, const EF& b, u32 op) {
    if (op == VM_OP_ADD) return kb::ef_add(a, b);
    const EF ab = kb::ef_mul(a, b);
    if (op == VM_OP_MUL) return ab;
    return kb::ef_add_base(kb::ef_sub(kb::ef_sub(kb::ef_dbl(ab), a), b), kb::ONE);  // 2ab - a - b + 1
}

struct MemBuf {
    u32* p = nullptr;
    u64 len = 0;
    // [dev_lo, dev_hi): cells whose current value lives in the DEVICE image only (the frames of a batch that ran on the device,
    // lm_vm_device.hip).  The sequential runner checks every access against the window; the first touch brings the cells back
    // (on_touch) and closes the window.  Empty (0, 0) on the host-only path.
    u64 dev_lo = 0, dev_hi = 0;
    std::function<bool()> on_touch;
    bool touch_failed = false;
    inline void guard(u64 i) {
        if (__builtin_expect(i - dev_lo < dev_hi - dev_lo, 0)) touch();
    }
    void touch() {
        if (on_touch && !on_touch()) touch_failed = true;
        dev_lo = dev_hi = 0;
    }
    // ---- deferred Poseidon16 calls of the sequential runner -------------------------------------------------------------------------
    // A program's sequential parts hold long hash chains (the aggregation program hashes the 1550 public keys before its parallel
    // loop: 1750 dependent permutations, 0.7 ms of one host core) whose results nothing reads until much later.  With `lazy_on` the
    // sequential runner records such a call instead of executing it: the output cells stay None in the arena, `owner` (a shadow of
    // the arena) names the call that will define them, and the first access to one of them (peek / set of MainMem) executes the call
    // and whatever pending calls its inputs hang on.  What is still pending when a batch goes to the device is executed while the
    // segment kernel runs (device_batch) — the segments see None in those cells, and one that reads them fails and sends the batch to
    // the host, as any other irregularity does.  Deferring changes no result and no error: a call is deferred only when all its inputs
    // are defined or pending and all its output cells are fresh (anything else takes the eager path, which raises what the reference
    // raises, at the same cycle), so executing it later cannot fail, and every later write to a pending cell executes the call first.
    // The same holds for an ExtensionOp whose operands and result are ALL defined or pending: it writes nothing, it is a check
    // (copy_8(computed_hash, expected_hash) behind a hash chain is the case that matters: executed at once it would pull the whole
    // chain in front of the parallel loop) plus the values of its table rows.  Rows are reserved in the log when the instruction is
    // met and filled when the check is executed.  A deferred check CAN fail, and an error met while checks are pending may not be the
    // first one of the program: in both cases the run is repeated with everything executed at once (execute_impl), which reports
    // what the reference reports.
    struct LazyCall {
        u64 left_first, left_second, arg_b, res;  // ExtensionOp: left_first = a, arg_b = b, res = result pointer
        u64 size, row_at;                         // ExtensionOp: elements, first word of its rows in the log
        u32 op;
        u8 kind, permute, n_out, done, is_be;     // kind 0 Poseidon16, 1 ExtensionOp check
    };
    UVec<u32>* lazy_rows = nullptr;  // the run's ExtensionOp log
    bool lazy_failed = false;        // a deferred check did not hold
    u64 lazy_checks = 0;             // checks ever deferred in this run
    bool lazy_on = false;
    u64 lazy_open = 0;              // calls recorded and not yet executed
    u64 lazy_total = 0;             // calls ever deferred in this run (statistics)
    std::vector<LazyCall> lazy;     // in program order
    size_t lazy_drained = 0;        // every call before this index is done
    u32* owner = nullptr;           // owner[i] = 1 + index of the pending call that defines cell i, 0 otherwise (allocated on first use)
    bool lazy_alloc() {
        if (owner) return true;
        void* q = mmap(nullptr, (1ull << MAX_LOG_MEMORY_SIZE) * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (q == MAP_FAILED) return false;
        owner = (u32*)q;
        return true;
    }
    std::function<void()> late_wait;  // kind 2 (late input words, VmLate): blocks until the words at left_first (a host pointer) are final
    void lazy_execute_check(LazyCall& c);
    void lazy_execute(LazyCall& c) {  // all inputs are defined by now
        if (c.kind == 2) {
            if (late_wait) {
                const double t0 = vm_now_ms();
                late_wait();
                if (vm_times() && vm_now_ms() - t0 > 0.01)
                    fprintf(stderr, "[vm] late input words (cells %llu..): the run waited %.2f ms for them\n", (unsigned long long)c.res, vm_now_ms() - t0);
            }
            memcpy(p + c.res, reinterpret_cast<const u32*>((uintptr_t)c.left_first), 4u * c.n_out);
            memset(owner + c.res, 0, 4u * c.n_out);
            c.done = 1;
            lazy_open--;
            return;
        }
        if (c.kind == 1) {
            lazy_execute_check(c);
            c.done = 1;
            lazy_open--;
            return;
        }
        alignas(64) u32 st[16];
        memcpy(st, p + c.left_first, 16), memcpy(st + 4, p + c.left_second, 16), memcpy(st + 8, p + c.arg_b, 32);
        if (c.permute)
            host_permute(st);
        else
            host_compress(st);
        memcpy(p + c.res, st, 4u * c.n_out);
        memset(owner + c.res, 0, 4u * c.n_out);
        c.done = 1;
        lazy_open--;
    }
    void lazy_force(u32 k) {  // call k and, first, the pending calls its inputs hang on (an explicit stack: a chain can be long)
        if (lazy[k].done) return;
        std::vector<u32> stack(1, k);
        while (!stack.empty()) {
            LazyCall& c = lazy[stack.back()];
            if (c.done) {
                stack.pop_back();
                continue;
            }
            u32 dep = 0;
            auto scan = [&](u64 at, u64 n) {
                for (u64 j = 0; j < n && !dep; j++) dep = owner[at + j];
            };
            if (c.kind == 0)
                scan(c.left_first, 4), scan(c.left_second, 4), scan(c.arg_b, 8);
            else if (c.kind == 1)
                scan(c.left_first, c.is_be ? c.size : 5 * c.size), scan(c.arg_b, 5 * c.size), scan(c.res, 5);
            // (kind 2, late input words: no inputs in memory)
            if (dep)
                stack.push_back(dep - 1);
            else {
                lazy_execute(c);
                stack.pop_back();
            }
        }
    }
    u32 lazy_cell(u64 i) {  // value of cell i after executing the call that defines it (UNDEF: no call does)
        const u32 o = owner[i];
        if (!o) return 0xFFFFFFFFu;
        lazy_force(o - 1);
        return p[i];
    }
    void lazy_drain() {  // everything pending, in program order (dependencies point backwards)
        for (; lazy_drained < lazy.size(); lazy_drained++)
            if (!lazy[lazy_drained].done) lazy_force((u32)lazy_drained);
        lazy.clear();
        lazy_drained = 0;
    }
    void lazy_reset() {  // an aborted run: forget the pending calls
        for (LazyCall& c : lazy)
            if (!c.done && c.kind != 1) memset(owner + c.res, 0, 4u * c.n_out);
        lazy.clear();
        lazy_drained = 0, lazy_open = 0;
    }
    // cells [at, at + n) will hold the n words at src once late_wait() has returned: pending, owned by a call of kind 2.  false: not
    // possible (deferral off, a cell already defined or owned, n > 64) — the caller waits and writes the words as usual.
    bool lazy_late(u64 at, const u32* src, u32 n) {
        if (!lazy_on || !owner || n == 0 || n > 64 || at + n > MAX_MEMORY) return false;
        guard(at), guard(at + n - 1);
        for (u32 j = 0; j < n; j++)
            if (at + j < len && (p[at + j] != 0xFFFFFFFFu || owner[at + j])) return false;
        if (at + n > len) grow(at + n);
        if (!lazy_open && !lazy.empty()) lazy.clear(), lazy_drained = 0;
        LazyCall c;
        memset(&c, 0, sizeof c);
        c.kind = 2, c.left_first = (u64)(uintptr_t)src, c.res = at, c.n_out = (u8)n;
        lazy.push_back(c);
        const u32 tag = (u32)lazy.size();
        for (u32 j = 0; j < n; j++) owner[at + j] = tag;
        lazy_open++, lazy_total++;
        return true;
    }
    MemBuf() {
        {  // an arena released by an earlier run: its pages are already resident
            std::lock_guard<std::mutex> lk(cache_mu());
            auto& c = cache();
            if (!c.empty()) {
                p = c.back().first, owner = c.back().second;
                c.pop_back();
                return;
            }
        }
        void* q = mmap(nullptr, MAX_MEMORY * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (q == MAP_FAILED) throw std::bad_alloc();
        p = (u32*)q;
    }
    ~MemBuf() {
        if (!p) return;
        if (owner) lazy_reset();
        {
            std::lock_guard<std::mutex> lk(cache_mu());
            if (cache().size() < 4) {
                cache().push_back(std::make_pair(p, owner));
                return;
            }
        }
        notify_release(p);
        munmap(p, MAX_MEMORY * 4);
        if (owner) munmap(owner, MAX_MEMORY * 4);
    }
    MemBuf(const MemBuf&) = delete;
    MemBuf& operator=(const MemBuf&) = delete;
    u64 size() const { return len; }
    u32* data() { return p; }
    const u32* data() const { return p; }
    void grow(u64 n) {  // [len, n) becomes undefined
        for (u64 i = len; i < n; i++) p[i] = UNDEF;
        if (n > len) len = n;
    }
    static std::mutex& cache_mu() {
        static std::mutex m;
        return m;
    }
    static std::vector<std::pair<u32*, u32*>>& cache() {
        static std::vector<std::pair<u32*, u32*>> c;
        return c;
    }
};
// growable array without value-initialisation (the logs are written exactly once)
template <class T>
struct UVec {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    UVec() {}
    ~UVec() {
        notify_release(p);
        free(p);
    }
    UVec(const UVec&) = delete;
    UVec& operator=(const UVec&) = delete;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    const T* begin() const { return p; }
    const T* end() const { return p + n; }
    // Whole pages of its own: the buffers that hold a run's log are registered with the HIP runtime for the uploads (vm_ensure_pinned),
    // and a registered range must not share a page with other allocations — the runtime pins pageable copy targets on the fly, page
    // by page, and releases them again (a registered chunk of the malloc heap next to such a target lost its device mapping now and
    // then: "Memory access fault by GPU" on a heap address, once in a few runs of the test suite).
    static constexpr size_t PAGE = 4096;
    void reserve(size_t c) {
        if (c <= cap) return;
        size_t nc = cap ? cap : 256;
        while (nc < c) nc *= 2;
        const size_t bytes = (nc * sizeof(T) + PAGE - 1) / PAGE * PAGE;
        void* q = nullptr;
        if (posix_memalign(&q, PAGE, bytes) != 0 || !q) throw std::bad_alloc();
        if (n) memcpy(q, p, n * sizeof(T));
        notify_release(p);
        free(p);
        p = (T*)q, cap = bytes / sizeof(T);
    }
    void push_back(const T& v) {
        if (n == cap) reserve(n + 1);
        p[n++] = v;
    }
    void release() {
        notify_release(p);
        free(p);
        p = nullptr;
        n = cap = 0;
    }
    void swap(UVec& o) {
        std::swap(p, o.p), std::swap(n, o.n), std::swap(cap, o.cap);
    }
    T* extend(size_t k) {  // k uninitialised elements at the end
        reserve(n + k);
        T* r = p + n;
        n += k;
        return r;
    }
};

// a deferred ExtensionOp check (exec_multi_row, exec.rs:106-190, with every operand defined): the values of its rows, result == c
inline void MemBuf::lazy_execute_check(LazyCall& c) {
    const u64 size = c.size, a_stride = c.is_be ? 1 : 5;
    static thread_local std::vector<EF> elems, vbs, comp;  // (size <= 256 for a deferred check)
    elems.resize(size), vbs.resize(size), comp.resize(size);
    for (u64 i = 0; i < size; i++) {
        EF a, b;
        if (c.is_be)
            a = kb::ef_from_base(p[c.left_first + i]);
        else
            memcpy(a.v, p + c.left_first + i * a_stride, 20);
        memcpy(b.v, p + c.arg_b + i * 5, 20);
        elems[i] = vm_compute_elem(a, b, c.op);
        vbs[i] = b;
    }
    comp[size - 1] = elems[size - 1];
    for (u64 i = size - 1; i-- > 0;) comp[i] = c.op == VM_OP_POLY_EQ ? kb::ef_mul(elems[i], comp[i + 1]) : kb::ef_add(elems[i], comp[i + 1]);
    if (memcmp(comp[0].v, p + c.res, 20) != 0) lazy_failed = true;
    u32* rows = lazy_rows->data() + c.row_at;
    for (u64 i = 0; i < size; i++) {
        u32* r = rows + i * LM_VM_EXTENSION_ROW_WORDS;
        memcpy(r + 9, vbs[i].v, 20);
        memcpy(r + 14, comp[0].v, 20);
        memcpy(r + 19, comp[i].v, 20);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// errors (lean_vm/src/diagnostics/error.rs)
// ---------------------------------------------------------------------------------------------------------------------
struct Err {
    bool set = false;
    std::string msg;
    void raise(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
        if (set) return;
        char buf[256];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        msg = buf;
        set = true;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// memory (execution/memory.rs)
// ---------------------------------------------------------------------------------------------------------------------
struct MainMem {  // Memory: grows on write, write-once cells
    MemBuf& m;
    u32 peek(u64 i) const {
        m.guard(i);
        if (i >= m.len) return UNDEF;
        const u32 v = m.p[i];
        if (__builtin_expect(v == UNDEF && m.lazy_open, 0)) return m.lazy_cell(i);  // (a deferred Poseidon call defines it: MemBuf)
        return v;
    }
    // the pointee of a DEREF whose result is unknown: a cell that holds a LATE input word (MemBuf kind 2) reads as None — the
    // instruction then does nothing and resolve_deref_hints fills the result at the end of the run, exactly as for a cell that is not
    // defined yet (a range check on a small value lands on the public input: it must not wait for it).  Counted as a deferred check:
    // an error met afterwards repeats the run with everything executed at once.
    u32 peek_pointee(u64 i) const {
        m.guard(i);
        if (i >= m.len) return UNDEF;
        const u32 v = m.p[i];
        if (__builtin_expect(v == UNDEF && m.lazy_open, 0)) {
            const u32 o = m.owner[i];
            if (o && m.lazy[o - 1].kind == 2 && !m.lazy[o - 1].done) {
                m.lazy_checks++;
                return UNDEF;
            }
            return m.lazy_cell(i);
        }
        return v;
    }
    bool set(u64 i, u32 v, Err& e) {
        m.guard(i);
        if (i >= m.len) {
            if (i >= MAX_MEMORY) {
                e.raise("OutOfMemory");
                return false;
            }
            m.grow(i + 1);
        }
        u32& c = m.p[i];
        if (__builtin_expect(c == UNDEF && m.lazy_open, 0)) (void)m.lazy_cell(i);  // the deferred call wrote first
        if (c == UNDEF)
            c = v;
        else if (c != v) {
            e.raise("MemoryAlreadySet { address: %llu, prev_value: %u, new_value: %u }", (unsigned long long)i, kb::from_monty(c),
                    kb::from_monty(v));
            return false;
        }
        return true;
    }
};
// late words of a hint entry (Witness::late): MainMem keeps the cells pending, anything else waits for the words
inline bool mem_set_late(MainMem& mm, u64 at, const u32* src, u32 n) { return mm.m.lazy_late(at, src, n); }
struct SegMem;
inline bool mem_set_late(SegMem&, u64, const u32*, u32) { return false; }
struct SegMem {  // SegmentMemory (memory.rs:118-189): shared prefix read-only, own slice writable, other writes deferred
    const u32* shared;
    u64 shared_len;
    u32* seg;
    u64 seg_start, seg_len;
    UVec<std::pair<u64, u32>>* deferred;  // the running thread's log (ThreadLog)
    u32 peek(u64 i) const {
        if (i < seg_start) return i < shared_len ? shared[i] : UNDEF;
        const u64 o = i - seg_start;
        return o < seg_len ? seg[o] : UNDEF;
    }
    u32 peek_pointee(u64 i) const { return peek(i); }
    bool set(u64 i, u32 v, Err& e) {
        if (i < seg_start || i - seg_start >= seg_len) {
            deferred->push_back(std::pair<u64, u32>(i, v));
            return true;
        }
        u32& c = seg[i - seg_start];
        if (c == UNDEF)
            c = v;
        else if (c != v) {
            e.raise("MemoryAlreadySet { address: %llu, prev_value: %u, new_value: %u }", (unsigned long long)i, kb::from_monty(c),
                    kb::from_monty(v));
            return false;
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// decoded program
// ---------------------------------------------------------------------------------------------------------------------
enum : u8 { K_ADD = VM_K_ADD, K_MUL = VM_K_MUL, K_DEREF = VM_K_DEREF, K_JUMP = VM_K_JUMP, K_POSEIDON = VM_K_POSEIDON, K_EXTOP = VM_K_EXTOP };
typedef VmInstr Instr;      // (lm_vm_device.h: the device interpreter reads the same records)
typedef VmHintRec HintRec;

struct Trace {  // runner.rs:70-76
    UVec<u32> pcs, fps;
    UVec<u32> pos;  // LM_VM_POSEIDON_CALL_WORDS per call
    UVec<u32> ext;  // LM_VM_EXTENSION_ROW_WORDS per row
    UVec<std::pair<u64, u64>> pending;  // (target_addr, src_addr)
    u64 n_add = 0, n_mul = 0, n_deref = 0, n_jump = 0;
};

struct Cursors {  // named hint cursors
    std::vector<u64> index;
};

}  // namespace
}  // namespace lmh

using namespace lmh;

struct lmh_bytecode {
    std::vector<u32> multilinear;
    u32 log_size = 0, ending_pc = 0, starting_frame_memory = 0, n_names = 0;
    u64 n_instructions = 0;
    std::vector<Instr> code;
    std::vector<u32> hint_begin;  // n_instructions + 1
    std::vector<HintRec> hints;
    std::vector<std::string> names;  // names of the HintWitness streams by id (lmh_bytecode_set_hint_names; empty when the caller never gave them)
    mutable std::mutex hash_mu;
    mutable bool hash_done = false;
    mutable u32 hash[8];
    // process-unique id: the key of this object's device copies in a context's cache (lm_ctx_cache_get: the copies belong to the
    // context and die with it; an id is never reused, so a context never finds another bytecode's tables under it)
    const u64 uid = next_uid();
    static u64 next_uid() {
        static std::atomic<u64> n{1};
        return n.fetch_add(1, std::memory_order_relaxed);
    }
};
namespace lmh {
u64 vm_bytecode_uid(const lmh_bytecode* bc) { return bc->uid; }
}  // namespace lmh

// the log buffers of the last released execution: their pages are resident, the next run writes into them without faulting
namespace lmh {
namespace {
struct LogCache {
    std::mutex mu;
    bool full = false;
    UVec<u32> pcs, fps, pos, ext;
    UVec<uint8_t> defined;
};
LogCache& log_cache() {
    static LogCache* c = new LogCache();  // never destroyed (its buffers may be pinned by lm_node.cpp: nothing to unpin at exit)
    return *c;
}
}  // namespace
}  // namespace lmh

namespace lmh {
struct DevRun;
void dev_run_free(DevRun* d);
}  // namespace lmh

struct lmh_execution {
    lmh_execution() {
        LogCache& c = log_cache();
        std::lock_guard<std::mutex> lk(c.mu);
        if (!c.full) return;
        tr.pcs.swap(c.pcs), tr.fps.swap(c.fps), tr.pos.swap(c.pos), tr.ext.swap(c.ext), defined.swap(c.defined);
        tr.pcs.n = tr.fps.n = tr.pos.n = tr.ext.n = defined.n = 0;
        c.full = false;
    }
    ~lmh_execution() {
        lmh::dev_run_free(dev);
        LogCache& c = log_cache();
        std::lock_guard<std::mutex> lk(c.mu);
        if (c.full) return;
        tr.pcs.swap(c.pcs), tr.fps.swap(c.fps), tr.pos.swap(c.pos), tr.ext.swap(c.ext), defined.swap(c.defined);
        c.full = true;
    }
    Trace tr;
    MemBuf memory;                 // UNDEF -> 0 after the run
    UVec<uint8_t> defined;
    u64 public_memory_size = 0, runtime_memory_size = 0;
    lmh::DevRun* dev = nullptr;    // a run whose parallel batches executed on the device (lmh_execute_bytecode_device): the logs and the
                                   // memory image are resident in HBM; the host view is materialised on demand (lmh_execution_view)
    // where the parallel batches of this run executed (lmh_execution_info): a batch the device hands back is not an error, but a node
    // that expects the device path wants to know (bench.py fails when the default workload falls back)
    u32 n_device_batches = 0, n_host_batches = 0, run_repeated = 0;
    std::string host_batch_reason;
};

namespace lmh {
namespace {
typedef uint8_t u8;

struct Witness {
    u32 preamble_memory_len;
    const u64* name_begin;
    const u64* entry_offset;
    const u32* data;
    const VmLate* late = nullptr;  // words of `data` that are not final yet (the sequential runner only: MainMem::set_late)
};

template <class Mem>
struct Machine {
    const lmh_bytecode& bc;
    const Witness& w;
    Mem& mem;
    Trace& tr;
    Cursors& cur;
    u64 pc, fp, ap;
    Err err;

    Machine(const lmh_bytecode& b, const Witness& wi, Mem& m, Trace& t, Cursors& c) : bc(b), w(wi), mem(m), tr(t), cur(c), pc(0), fp(0), ap(0) {}

    // MemOrConstant / MemOrFpOrConstant::read_value: UNDEF when the value is unknown
    u32 read(u8 mode, u32 canon, u32 monty) const {
        if (mode == LM_VM_ARG_CONST) return monty;
        if (mode == LM_VM_ARG_MEM) return mem.peek(fp + canon);
        return kb::to_monty((u32)((fp + canon) % kb::P));  // F::from_usize(fp + offset)
    }
    u32 need(u8 mode, u32 canon, u32 monty) {  // read_value(..)?
        const u32 v = read(mode, canon, monty);
        if (v == UNDEF) err.raise("UndefinedMemory(%llu)", (unsigned long long)(fp + canon));
        return v;
    }
    u32 need_mem(u64 addr) {
        const u32 v = mem.peek(addr);
        if (v == UNDEF) err.raise("UndefinedMemory(%llu)", (unsigned long long)addr);
        return v;
    }
    static u64 usize(u32 monty) { return kb::from_monty(monty); }

    u32 hint_arg(const HintRec& h, int k) {  // operand of a hint: value (Montgomery)
        const u8 mode = h.mode[k];
        if (mode == LM_VM_ARG_CONST) return kb::to_monty(h.args[k]);
        if (mode == LM_VM_ARG_MEM) return need_mem(fp + h.args[k]);
        return kb::to_monty((u32)((fp + h.args[k]) % kb::P));
    }
    bool hint_arg_address(const HintRec& h, int k, u64& addr) {  // memory_address(fp)?
        if (h.mode[k] != LM_VM_ARG_MEM) {
            err.raise("NotAPointer");
            return false;
        }
        addr = fp + h.args[k];
        return true;
    }

    // ---- hints (isa/hint.rs:270-386, CustomHint::execute :137-203).  ParallelBatchStart is the run loop's business. -----
    void run_hint(const HintRec& h) {
        switch (h.kind) {
            case LM_VM_HINT_REQUEST_MEMORY: {
                const u32 size = hint_arg(h, 1);
                if (err.set) return;
                mem.set(fp + h.args[0], kb::to_monty((u32)(ap % kb::P)), err);
                ap += usize(size);
                break;
            }
            case LM_VM_HINT_INVERSE: {
                const u32 v = hint_arg(h, 0);
                if (err.set) return;
                mem.set(fp + h.args[1], v ? kb::inv(v) : 0u, err);  // try_inverse().unwrap_or(ZERO)
                break;
            }
            case LM_VM_HINT_DEREF:
                tr.pending.push_back(std::pair<u64, u64>(fp + h.args[1], fp + h.args[0]));  // (target_addr, src_addr)
                break;
            case LM_VM_HINT_DECOMPOSE_BITS_XMSS: {
                const u32 dp = hint_arg(h, 0), sp = hint_arg(h, 1), nn = hint_arg(h, 2), cs = hint_arg(h, 3);
                if (err.set) return;
                const u64 chunk = usize(cs);
                if (chunk == 0 || 24 % chunk) {
                    err.raise("Panic: hint_decompose_bits_xmss: 24 is not a multiple of the chunk size");
                    return;
                }
                u64 out = usize(dp);
                const u64 src = usize(sp), num = usize(nn);
                for (u64 i = 0; i < num && !err.set; i++) {
                    const u32 v = need_mem(src + i);
                    if (err.set) return;
                    const u64 x = usize(v);
                    for (u64 j = 0; j < 24 / chunk; j++)
                        if (!mem.set(out++, kb::to_monty((u32)((x >> (chunk * j)) & ((1ull << chunk) - 1))), err)) return;
                }
                break;
            }
            case LM_VM_HINT_DECOMPOSE_BITS_MERKLE_WHIR: {
                const u32 dp = hint_arg(h, 0), vv = hint_arg(h, 1), cs = hint_arg(h, 2);
                if (err.set) return;
                const u64 chunk = usize(cs), x = usize(vv);
                if (chunk == 0 || 24 % chunk) {
                    err.raise("Panic: hint_decompose_bits_merkle_whir: 24 is not a multiple of the chunk size");
                    return;
                }
                u64 out = usize(dp);
                for (u64 j = 0; j < 24 / chunk; j++)
                    if (!mem.set(out++, kb::to_monty((u32)((x >> (chunk * j)) & ((1ull << chunk) - 1))), err)) return;
                break;
            }
            case LM_VM_HINT_DECOMPOSE_BITS: {  // to_big_endian_in_field(to_decompose, num_bits)
                const u32 vv = hint_arg(h, 0), mi = hint_arg(h, 1), nb = hint_arg(h, 2);
                if (err.set) return;
                const u64 x = usize(vv), at = usize(mi), bits = usize(nb);
                if (bits > 31) {
                    err.raise("Panic: hint_decompose_bits: num_bits > F::bits()");
                    return;
                }
                for (u64 j = 0; j < bits; j++)
                    if (!mem.set(at + j, ((x >> (bits - 1 - j)) & 1) ? kb::ONE : 0u, err)) return;
                break;
            }
            case LM_VM_HINT_LESS_THAN: {
                const u32 a = hint_arg(h, 0), b = hint_arg(h, 1);
                u64 at;
                if (err.set || !hint_arg_address(h, 2, at)) return;
                mem.set(at, usize(a) < usize(b) ? kb::ONE : 0u, err);
                break;
            }
            case LM_VM_HINT_LOG2_CEIL: {
                const u32 n = hint_arg(h, 0);
                u64 at;
                if (err.set || !hint_arg_address(h, 1, at)) return;
                mem.set(at, kb::to_monty(log2_ceil_u64(usize(n))), err);
                break;
            }
            case LM_VM_HINT_WITNESS_INLINE:
            case LM_VM_HINT_WITNESS_INDIRECT: {
                const u32 name = h.args[0];
                const u64 e = w.name_begin[name] + cur.index[name];
                if (e >= w.name_begin[name + 1]) {
                    err.raise("Panic: hint_witness: exhausted entries for name %u (index=%llu)", name, (unsigned long long)cur.index[name]);
                    return;
                }
                cur.index[name]++;
                u64 dest;
                if (h.kind == LM_VM_HINT_WITNESS_INLINE)
                    dest = fp + h.args[1];
                else {
                    const u32 p = need_mem(fp + h.args[1]);
                    if (err.set) return;
                    dest = usize(p);
                }
                for (u64 k = w.entry_offset[e]; k < w.entry_offset[e + 1]; k++) {
                    if (__builtin_expect(w.late != nullptr, 0)) {  // words still being computed: their cells stay pending (VmLate)
                        u32 r = 0;
                        while (r < w.late->n_ranges && w.late->first_word[r] != k) r++;
                        if (r < w.late->n_ranges && k + w.late->n_words[r] <= w.entry_offset[e + 1]) {
                            const u32 n = w.late->n_words[r];
                            if (mem_set_late(mem, dest, w.data + k, n)) {
                                dest += n, k += n - 1;
                                continue;
                            }
                            if (vm_times()) fprintf(stderr, "[vm] late hint words written to cell %llu could not stay pending: waiting\n", (unsigned long long)dest);
                            w.late->wait();  // (not deferrable here: the words are needed now)
                        } else {
                            for (u32 q = 0; q < w.late->n_ranges; q++)
                                if (k > w.late->first_word[q] && k < w.late->first_word[q] + w.late->n_words[q]) w.late->wait();
                        }
                    }
                    if (!mem.set(dest++, w.data[k], err)) return;
                }
                break;
            }
            case LM_VM_HINT_DEBUG_ASSERT: {
                const u32 l = hint_arg(h, 0), r = hint_arg(h, 1);
                if (err.set) return;
                const u64 lv = usize(l), rv = usize(r);
                if (h.args[3] && rv >= (1ull << MIN_LOG_MEMORY_SIZE)) {
                    err.raise("RangeCheckWithTooBigRange { range: %llu }", (unsigned long long)rv);
                    return;
                }
                bool ok = true;
                switch (h.args[2]) {
                    case 0: ok = lv == rv; break;
                    case 1: ok = lv != rv; break;
                    case 2: ok = lv < rv; break;
                    default: ok = lv <= rv; break;
                }
                if (!ok) err.raise("DebugAssertFailed(%llu ? %llu, kind %u)", (unsigned long long)lv, (unsigned long long)rv, h.args[2]);
                break;
            }
            default:
                break;
        }
    }

    // ---- precompiles -----------------------------------------------------------------------------------------------------
    bool get_slice(u64 at, u32 n, u32* out) {
        for (u32 i = 0; i < n; i++) {
            out[i] = need_mem(at + i);
            if (err.set) return false;
        }
        return true;
    }
    bool set_slice(u64 at, u32 n, const u32* v) {
        for (u32 i = 0; i < n; i++)
            if (!mem.set(at + i, v[i], err)) return false;
        return true;
    }
    // The call is recorded instead of executed (MemBuf: deferred Poseidon16 calls) when that cannot change anything: every input cell
    // is defined or the output of a pending call, every output cell is fresh.  false: the caller executes it now.
    bool lazy_poseidon(u64 left_first, u64 left_second, u64 arg_b, u64 res, bool permute, u32 n_out) {
        if constexpr (std::is_same<Mem, MainMem>::value) {
            MemBuf& m = mem.m;
            if (!m.lazy_on || res + n_out > MAX_MEMORY || m.lazy.size() >= 0x7FFFFFF0u) return false;
            auto known = [&](u64 at, u32 n) {
                for (u32 j = 0; j < n; j++) {
                    m.guard(at + j);
                    if (at + j >= m.len || (m.p[at + j] == UNDEF && !m.owner[at + j])) return false;
                }
                return true;
            };
            if (!known(left_first, 4) || !known(left_second, 4) || !known(arg_b, 8)) return false;
            for (u32 j = 0; j < n_out; j++) {
                m.guard(res + j);
                if (res + j < m.len && (m.p[res + j] != UNDEF || m.owner[res + j])) return false;
            }
            if (res + n_out > m.len) m.grow(res + n_out);
            if (!m.lazy_open && !m.lazy.empty()) m.lazy.clear(), m.lazy_drained = 0;  // (everything recorded so far has been executed)
            MemBuf::LazyCall c;
            memset(&c, 0, sizeof c);
            c.left_first = left_first, c.left_second = left_second, c.arg_b = arg_b, c.res = res, c.permute = permute, c.n_out = (u8)n_out;
            m.lazy.push_back(c);
            const u32 tag = (u32)m.lazy.size();
            for (u32 j = 0; j < n_out; j++) m.owner[res + j] = tag;
            m.lazy_open++, m.lazy_total++;
            return true;
        } else
            return false;
    }
    void poseidon(const Instr& in, u32 va, u32 vb, u32 vc) {  // Poseidon16Precompile::execute (poseidon_16/mod.rs:209-289)
        const bool permute = in.x0 & 1, half = in.x0 & 2, hard = in.x0 & 4;
        const u64 arg_a = usize(va), arg_b = usize(vb), res = usize(vc);
        const u64 left_first = hard ? in.x1 : arg_a;
        const u64 left_second = hard ? arg_a : arg_a + 4;
        if (!lazy_poseidon(left_first, left_second, arg_b, res, permute, permute ? 16u : (half ? 4u : 8u))) {
            alignas(64) u32 st[16];
            if (!get_slice(left_first, 4, st) || !get_slice(left_second, 4, st + 4) || !get_slice(arg_b, 8, st + 8)) return;
            if (permute) {
                host_permute(st);
                if (!set_slice(res, 16, st)) return;
            } else {
                host_compress(st);
                if (!set_slice(res, half ? 4 : 8, st)) return;
            }
        }
        const u32 rec[LM_VM_POSEIDON_CALL_WORDS] = {(u32)arg_a, (u32)arg_b, (u32)res, half ? 1u : 0u, hard ? 1u : 0u, hard ? in.x1 : 0u,
                                                    (u32)left_first, (u32)left_second, permute ? 1u : 0u};
        memcpy(tr.pos.extend(LM_VM_POSEIDON_CALL_WORDS), rec, sizeof rec);
    }

    // extension_op/exec.rs
    enum { OP_ADD = VM_OP_ADD, OP_MUL = VM_OP_MUL, OP_POLY_EQ = VM_OP_POLY_EQ };
    static EF compute_elem(const EF& a, const EF& b, u32 op) { return vm_compute_elem(a, b, op); }
    // An ExtensionOp over operands and a result that are all defined or pending, at least one of them pending, is recorded instead of
    // executed (MemBuf: it is a check).  false: the caller executes it now.
    bool lazy_extension_check(u64 pa, u64 pb, u64 pr, bool is_be, u32 op, u64 size) {
        if constexpr (std::is_same<Mem, MainMem>::value) {
            MemBuf& m = mem.m;
            if (!m.lazy_on || !m.lazy_open || !m.lazy_rows || size == 0 || size > 256 || m.lazy.size() >= 0x7FFFFFF0u) return false;
            bool pending = false;
            auto known = [&](u64 at, u64 n) {
                for (u64 j = 0; j < n; j++) {
                    m.guard(at + j);
                    if (at + j >= m.len) return false;
                    if (m.p[at + j] == UNDEF) {
                        if (!m.owner[at + j]) return false;
                        pending = true;
                    }
                }
                return true;
            };
            if (!known(pa, is_be ? size : 5 * size) || !known(pb, 5 * size) || !known(pr, 5) || !pending) return false;
            MemBuf::LazyCall c;
            memset(&c, 0, sizeof c);
            c.left_first = pa, c.arg_b = pb, c.res = pr, c.size = size, c.op = op, c.kind = 1, c.is_be = is_be;
            c.row_at = tr.ext.size();
            const u64 a_stride = is_be ? 1 : 5;
            u32* rows = tr.ext.extend(size * LM_VM_EXTENSION_ROW_WORDS);
            for (u64 i = 0; i < size; i++) {
                u32* r = rows + i * LM_VM_EXTENSION_ROW_WORDS;
                memset(r, 0, 4 * LM_VM_EXTENSION_ROW_WORDS);
                r[0] = is_be, r[1] = i == 0, r[2] = op == OP_ADD, r[3] = op == OP_MUL, r[4] = op == OP_POLY_EQ, r[5] = (u32)(size - i);
                r[6] = (u32)(pa + i * a_stride), r[7] = (u32)(pb + i * 5), r[8] = (u32)pr;
            }
            m.lazy.push_back(c);
            m.lazy_open++, m.lazy_checks++;
            return true;
        } else
            return false;
    }
    bool peek_ef(u64 at, EF& out) const {
        for (int k = 0; k < 5; k++) {
            const u32 v = mem.peek(at + k);
            if (v == UNDEF) return false;
            out.v[k] = v;
        }
        return true;
    }
    bool need_ef(u64 at, EF& out) {
        for (int k = 0; k < 5; k++) {
            out.v[k] = need_mem(at + k);
            if (err.set) return false;
        }
        return true;
    }
    bool set_ef(u64 at, const EF& v) { return set_slice(at, 5, v.v); }
    bool make_slices_equal_and_defined(u64 p0, u64 p1, u32 len) {  // memory.rs:41-66
        for (u32 i = 0; i < len; i++) {
            const u32 v0 = mem.peek(p0 + i), v1 = mem.peek(p1 + i);
            if (v0 != UNDEF && v1 != UNDEF) {
                if (v0 != v1) {
                    err.raise("NotEqual(%u, %u)", kb::from_monty(v0), kb::from_monty(v1));
                    return false;
                }
            } else if (v0 != UNDEF) {
                if (!mem.set(p1 + i, v0, err)) return false;
            } else if (v1 != UNDEF) {
                if (!mem.set(p0 + i, v1, err)) return false;
            } else {
                if (!mem.set(p0 + i, 0, err) || !mem.set(p1 + i, 0, err)) return false;
            }
        }
        return true;
    }
    bool solve_unknowns(u64 pa, u64 pb, u64 pr, bool is_be, u32 op) {  // exec.rs:29-104
        EF a, b, c;
        bool ka, kb_, kc;
        if (is_be) {
            const u32 v = mem.peek(pa);
            ka = v != UNDEF;
            a = kb::ef_from_base(ka ? v : 0);
        } else
            ka = peek_ef(pa, a);
        kb_ = peek_ef(pb, b);
        kc = peek_ef(pr, c);
        if (op == OP_MUL && !is_be) {  // "copy_5"
            if (kb_ && kb::ef_eq(b, kb::ef_one())) return make_slices_equal_and_defined(pa, pr, 5);
            if (ka && kb::ef_eq(a, kb::ef_one())) return make_slices_equal_and_defined(pb, pr, 5);
        }
        if (ka && kb_ && kc) {
            if (!kb::ef_eq(compute_elem(a, b, op), c)) {
                err.raise("InvalidExtensionOp");
                return false;
            }
        } else if (ka && kb_ && !kc) {
        } else if (!ka && kb_ && kc) {
            const EF x = op == OP_ADD ? kb::ef_sub(c, b) : kb::ef_mul(c, kb::ef_inv(b));
            if (is_be) {
                if (x.v[1] | x.v[2] | x.v[3] | x.v[4]) {
                    err.raise("Panic: solved A not in base field");
                    return false;
                }
                return mem.set(pa, x.v[0], err);
            }
            return set_ef(pa, x);
        } else if (ka && !kb_ && kc) {
            const EF x = op == OP_ADD ? kb::ef_sub(c, a) : kb::ef_mul(c, kb::ef_inv(a));
            return set_ef(pb, x);
        } else {
            err.raise("InvalidExtensionOp");
            return false;
        }
        return true;
    }
    void extension_op(const Instr& in, u32 va, u32 vb, u32 vc) {  // exec_multi_row (exec.rs:106-190)
        const u32 op = in.x0 & (OP_ADD | OP_MUL | OP_POLY_EQ);
        const bool is_be = in.x0 & 4;
        const u64 size = in.x1, pa = usize(va), pb = usize(vb), pr = usize(vc);
        if (lazy_extension_check(pa, pb, pr, is_be, op, size)) return;
        if (size == 1 && op != OP_POLY_EQ && !solve_unknowns(pa, pb, pr, is_be, op)) return;
        const u64 a_stride = is_be ? 1 : 5;
        // `size` comes from the bytecode (< 2^25): the operand vectors grow as the operands are read, so a size that runs off the
        // defined memory fails with UndefinedMemory before anything of its order is allocated
        // (scratch kept per thread: the recursion program runs tens of thousands of short calls on the host, three heap
        // allocations each were a third of their cost)
        static thread_local std::vector<EF> elems, vbs, comp;
        if (elems.capacity() > (1u << 16)) elems = std::vector<EF>(), vbs = std::vector<EF>(), comp = std::vector<EF>();  // (after a rare long call)
        elems.clear(), vbs.clear();
        elems.reserve(std::min<u64>(size, 1u << 12)), vbs.reserve(std::min<u64>(size, 1u << 12));
        for (u64 i = 0; i < size; i++) {
            EF a, b;
            if (is_be) {
                const u32 v = need_mem(pa + i);
                if (err.set) return;
                a = kb::ef_from_base(v);
            } else if (!need_ef(pa + i * a_stride, a))
                return;
            if (!need_ef(pb + i * 5, b)) return;
            elems.push_back(compute_elem(a, b, op));
            vbs.push_back(b);
        }
        comp.resize(size);
        comp[size - 1] = elems[size - 1];
        for (u64 i = size - 1; i-- > 0;) comp[i] = op == OP_POLY_EQ ? kb::ef_mul(elems[i], comp[i + 1]) : kb::ef_add(elems[i], comp[i + 1]);
        if (!set_ef(pr, comp[0])) return;
        u32* rows = tr.ext.extend(size * LM_VM_EXTENSION_ROW_WORDS);
        for (u64 i = 0; i < size; i++) {
            u32* r = rows + i * LM_VM_EXTENSION_ROW_WORDS;
            r[0] = is_be;
            r[1] = i == 0;
            r[2] = op == OP_ADD;
            r[3] = op == OP_MUL;
            r[4] = op == OP_POLY_EQ;
            r[5] = (u32)(size - i);
            r[6] = (u32)(pa + i * a_stride);
            r[7] = (u32)(pb + i * 5);
            r[8] = (u32)pr;
            memcpy(r + 9, vbs[i].v, 20);
            memcpy(r + 14, comp[0].v, 20);
            memcpy(r + 19, comp[i].v, 20);
        }
    }

    // ---- one instruction (isa/instruction.rs:146-246) -----------------------------------------------------------------------
    void step(const Instr& in) {
        switch (in.kind) {
            case K_ADD:
            case K_MUL: {  // nu_a (arg_a) op nu_c (arg_c) = nu_b (res)
                const bool mul = in.kind == K_MUL;
                const u32 r = read(in.mb, in.b, in.bm);
                if (r == UNDEF) {
                    const u32 a = need(in.ma, in.a, in.am);
                    if (err.set) return;
                    const u32 c = need(in.mc, in.c, in.cm);
                    if (err.set) return;
                    mem.set(fp + in.b, mul ? kb::mul(a, c) : kb::add(a, c), err);
                } else {
                    const u32 a = read(in.ma, in.a, in.am);
                    if (a == UNDEF) {  // a = res inv_op c
                        const u32 c = need(in.mc, in.c, in.cm);
                        if (err.set) return;
                        if (mul && c == 0) {
                            err.raise("DivByZero");
                            return;
                        }
                        mem.set(fp + in.a, mul ? kb::mul(r, kb::inv(c)) : kb::sub(r, c), err);
                    } else {
                        const u32 c = read(in.mc, in.c, in.cm);
                        if (c == UNDEF) {
                            if (in.mc != LM_VM_ARG_MEM) {
                                err.raise("NotAPointer");
                                return;
                            }
                            if (mul && a == 0) {
                                err.raise("DivByZero");
                                return;
                            }
                            mem.set(fp + in.c, mul ? kb::mul(r, kb::inv(a)) : kb::sub(r, a), err);
                        } else {
                            const u32 v = mul ? kb::mul(a, c) : kb::add(a, c);
                            if (v != r) {
                                err.raise("NotEqual(%u, %u)", kb::from_monty(v), kb::from_monty(r));
                                return;
                            }
                        }
                    }
                }
                if (mul)
                    tr.n_mul++;
                else
                    tr.n_add++;
                pc++;
                break;
            }
            case K_DEREF: {  // res = m[m[fp + shift_0] + shift_1]
                const u32 r = read(in.mc, in.c, in.cm);
                if (r == UNDEF) {
                    if (in.mc != LM_VM_ARG_MEM) {
                        err.raise("NotAPointer");
                        return;
                    }
                    const u32 p = need_mem(fp + in.a);
                    if (err.set) return;
                    const u32 v = mem.peek_pointee(usize(p) + in.b);
                    if (v != UNDEF && !mem.set(fp + in.c, v, err)) return;
                    // else: a range check, resolved by resolve_deref_hints
                } else {
                    const u32 p = need_mem(fp + in.a);
                    if (err.set) return;
                    if (!mem.set(usize(p) + in.b, r, err)) return;
                }
                tr.n_deref++;
                pc++;
                break;
            }
            case K_JUMP: {
                const u32 cond = need(in.ma, in.a, in.am);
                if (err.set) return;
                if (cond == 0)
                    pc++;
                else if (cond == kb::ONE) {
                    const u32 d = need(in.mb, in.b, in.bm);
                    if (err.set) return;
                    const u32 f = need(in.mc, in.c, in.cm);
                    if (err.set) return;
                    pc = usize(d);
                    fp = usize(f);
                } else {
                    err.raise("Panic: jump condition %u is not boolean", kb::from_monty(cond));
                    return;
                }
                tr.n_jump++;
                break;
            }
            default: {
                const u32 a = need(in.ma, in.a, in.am);
                if (err.set) return;
                const u32 b = need(in.mb, in.b, in.bm);
                if (err.set) return;
                const u32 c = need(in.mc, in.c, in.cm);
                if (err.set) return;
                if (__builtin_expect(g_vm_prof_on, 0)) {  // LM_VM_TIMES: where the sequential runner's time goes (the calling thread's totals)
                    const unsigned long long t0 = __builtin_ia32_rdtsc();
                    if (in.kind == K_POSEIDON)
                        poseidon(in, a, b, c), g_vm_prof[0] += __builtin_ia32_rdtsc() - t0, g_vm_prof[1]++;
                    else
                        extension_op(in, a, b, c), g_vm_prof[2] += __builtin_ia32_rdtsc() - t0, g_vm_prof[3]++;
                } else if (in.kind == K_POSEIDON)
                    poseidon(in, a, b, c);
                else
                    extension_op(in, a, b, c);
                if (err.set) return;
                pc++;
                break;
            }
        }
    }

    // run_loop (runner.rs:121-204).  Returns 0 Halted, 1 LoopBack, 2 ParallelBatch, -1 error.
    struct Batch {
        u64 batch_pc = 0, batch_fp = 0, frame_size = 0;
        u32 n_args = 0, end_mode = 0, end_value = 0;
        std::vector<u64> hint_indices_at_start;
        size_t cyc_at_arm = 0, pos_at_arm = 0, ext_at_arm = 0, pend_at_arm = 0;  // log sizes when iteration 0 started (its footprint sizes the device slots)
        bool armed = false;
    };
    // skip_arm_pc: the pc of the parallel loop whose batch was just handled.  run_loop (runner.rs:120-198) restarts ON that loop's header —
    // the last iteration, i == end — and its ParallelBatchStart hint arms a batch that can never fire (the loop returns), which makes the
    // reference ignore every later ParallelBatchStart of the program: only the FIRST parallel loop of a run is ever batched there.  A
    // batch is an execution strategy — the ExecutionResult equals the sequential one (tests: every run against the sequential oracle
    // VM) — so this runner does not arm on that first instruction and later parallel loops are batched too (the recursion program's
    // per-round query loops, programs/whir_verify.py).  ~0: the reference's arming, literally (error reports, LM_VM_REARM=0).
    int run(bool has_stop, u64 stop_pc, Batch& batch, u64 skip_arm_pc = ~0ull) {
        batch.armed = false;
        bool first = true;
        for (;;) {
            if (pc == bc.ending_pc) return 0;
            if (pc >= bc.n_instructions) {
                err.raise("PCOutOfBounds");
                return -1;
            }
            tr.pcs.push_back((u32)pc);
            tr.fps.push_back((u32)fp);
            for (u32 h = bc.hint_begin[pc]; h < bc.hint_begin[pc + 1]; h++) {
                const HintRec& hr = bc.hints[h];
                if (hr.kind == LM_VM_HINT_PARALLEL_BATCH_START) {
                    if (!batch.armed && !(first && pc == skip_arm_pc)) {
                        batch.armed = true;
                        batch.batch_pc = pc;
                        batch.batch_fp = fp;
                        batch.frame_size = ap - fp;
                        batch.n_args = hr.args[0];
                        batch.end_mode = hr.mode[1];
                        batch.end_value = hr.args[1];
                        batch.hint_indices_at_start = cur.index;
                        batch.cyc_at_arm = tr.pcs.size() - 1, batch.pos_at_arm = tr.pos.size() / LM_VM_POSEIDON_CALL_WORDS;
                        batch.ext_at_arm = tr.ext.size() / LM_VM_EXTENSION_ROW_WORDS, batch.pend_at_arm = tr.pending.size();
                    }
                    continue;
                }
                run_hint(hr);
                if (err.set) return -1;
            }
            step(bc.code[pc]);
            first = false;
            if (err.set) return -1;
            if (has_stop && pc == stop_pc) return 1;
            if (batch.armed && pc == batch.batch_pc) return 2;
        }
    }
};

// resolve_deref_hints (runner.rs:206-236): memory[target] = memory[memory[src]] for every recorded deref, repeated until no more
// progress, the rest zero-filled.  Almost every entry is a no-op (the DEREF instruction itself found its value): those are
// recognised by a read-only pass on the pool; the reference's sequential loop then runs over what is left, with its `resolved`
// set (targets already written are skipped) seeded by the no-op entries that share a target with a remaining one.
bool resolve_deref_hints(MainMem& mem, const UVec<std::pair<u64, u64>>& pending, u32 n_threads, Err& err) {
    const u64 n = pending.size();
    if (n == 0) return true;
    std::vector<uint8_t> noop(n, 0);
    const u64 chunk = 4096;
    vm_parallel_for((n + chunk - 1) / chunk, n_threads, [&](u64 c) {
        const u64 e = std::min(n, (c + 1) * chunk);
        for (u64 i = c * chunk; i < e; i++) {
            const u32 a = mem.peek(pending[i].second);
            if (a == UNDEF) continue;
            const u32 v = mem.peek(kb::from_monty(a));
            noop[i] = v != UNDEF && mem.peek(pending[i].first) == v;
        }
    });
    std::vector<u64> rest;
    for (u64 i = 0; i < n; i++)
        if (!noop[i]) rest.push_back(i);
    if (rest.empty()) return true;
    std::set<u64> rest_targets, resolved;
    for (u64 i : rest) rest_targets.insert(pending[i].first);
    for (u64 i = 0; i < n; i++)
        if (noop[i] && rest_targets.count(pending[i].first)) resolved.insert(pending[i].first);
    for (;;) {
        bool progress = false;
        for (u64 i : rest) {
            const auto& [target, src] = pending[i];
            if (resolved.count(target)) continue;
            const u32 a = mem.peek(src);
            if (a == UNDEF) {
                err.raise("Panic: deref hint source %llu is undefined", (unsigned long long)src);
                return false;
            }
            const u32 v = mem.peek(kb::from_monty(a));
            if (v == UNDEF) continue;
            if (!mem.set(target, v, err)) return false;
            resolved.insert(target);
            progress = true;
        }
        if (!progress) break;
    }
    for (u64 i : rest)
        if (!resolved.count(pending[i].first) && !mem.set(pending[i].first, 0, err)) return false;
    return true;
}

// per pool thread: the logs of the segments it has run in the current batch (persistent, grow-only)
struct ThreadLog {
    Trace tr;
    UVec<std::pair<u64, u32>> deferred;
};
ThreadLog& thread_log() {
    static thread_local ThreadLog* mine = nullptr;
    if (!mine) {
        mine = new ThreadLog();  // lives as long as the process (pool threads do)
        if (Pool* p = Pool::tl_worker_pool()) {
            std::lock_guard<std::mutex> lk(p->logs_mu);
            p->worker_logs.push_back(mine);
        }
    }
    return *mine;
}
void reset_log(ThreadLog* l) {
    l->tr.pcs.n = l->tr.fps.n = l->tr.pos.n = l->tr.ext.n = l->tr.pending.n = l->deferred.n = 0;
    l->tr.n_add = l->tr.n_mul = l->tr.n_deref = l->tr.n_jump = 0;
}
// start of a batch: the logs of the run's pool and of the calling thread are empty (the pool is leased to this run alone)
void thread_logs_reset(Pool* p) {
    reset_log(&thread_log());
    std::lock_guard<std::mutex> lk(p->logs_mu);
    for (void* l : p->worker_logs) reset_log(static_cast<ThreadLog*>(l));
}
UVec<std::pair<u64, u32>>& batch_deferred(Pool* p) {
    if (!p->deferred) p->deferred = new UVec<std::pair<u64, u32>>();
    return *static_cast<UVec<std::pair<u64, u32>>*>(p->deferred);
}

// The per-thread logs and the deferred-write list belong to the run's pool (leased to this run alone, PoolSet): concurrent runs
// of several provers in one process do not share them.
// the length a sequential run leaves behind its last call frame: one past the highest cell anything SET (the frame's arguments, or a cell
// an iteration defined in the next frame), not the end of the frame
static void trim_to_defined(MemBuf& memory, u64 floor, u64 hi) {
    while (hi > floor && memory.p[hi - 1] == UNDEF) hi--;
    memory.len = hi;
}

// trim_last_frame: this batch is one the reference would have run sequentially (Machine::run, skip_arm_pc): its run leaves the memory
// as long as the last cell it SET, not resized to the end of the last call frame — the length is put back to that frame's arguments.
bool handle_parallel_batch(const lmh_bytecode& bc, const Witness& w, MemBuf& memory, Trace& trace, Cursors& cur, u64& pc, u64& fp,
                           u64& ap, const Machine<MainMem>::Batch& batch, u32 n_threads, Err& err, bool trim_last_frame = false) {
    memory.lazy_drain();  // the segments read the arena directly
    if (memory.lazy_failed) {
        err.raise("a deferred check failed");  // (execute_impl repeats the run)
        return false;
    }
    Pool* const pool = tl_run_pool();
    if (!pool) {
        err.raise("handle_parallel_batch outside a run");
        return false;
    }
    MainMem mm{memory};
    const double tp0 = vm_now_ms();
    auto get = [&](u64 at) -> u32 {
        const u32 v = mm.peek(at);
        if (v == UNDEF) err.raise("UndefinedMemory(%llu)", (unsigned long long)at);
        return v;
    };
    const u32 sv = get(batch.batch_fp + 2);
    if (err.set) return false;
    const u64 start_value = kb::from_monty(sv);
    u64 end_value;
    if (batch.end_mode == LM_VM_ARG_CONST)
        end_value = batch.end_value;
    else {
        const u32 ev = get(batch.batch_fp + batch.end_value);
        if (err.set) return false;
        end_value = kb::from_monty(ev);
    }
    if (end_value < start_value) {
        err.raise("Panic: parallel batch: end value below the start value");
        return false;
    }
    const u64 n_iters = end_value - start_value;
    if (n_iters == 1) return true;
    if (n_iters == 0) {
        err.raise("Panic: parallel batch with zero iterations ran one");
        return false;
    }
    const u64 stride = fp - batch.batch_fp;
    const u32 return_pc = get(fp), saved_fp = get(fp + 1);
    if (err.set) return false;
    std::vector<u32> args(batch.n_args);
    for (u32 i = 0; i < batch.n_args; i++) {
        args[i] = get(batch.batch_fp + 2 + i);
        if (err.set) return false;
    }
    std::vector<u64> per_iter(cur.index.size());
    for (size_t k = 0; k < per_iter.size(); k++) per_iter[k] = cur.index[k] - batch.hint_indices_at_start[k];
    const u64 max_addr = batch.batch_fp + (n_iters + 1) * stride;
    if (max_addr > MAX_MEMORY) {
        err.raise("OutOfMemory");
        return false;
    }
    // memory.0.resize(max_addr, None), done BEFORE the call frames are written (the reference resizes after; Memory::set grows on demand
    // either way, so the final length is the same): the new cells are filled — and first touched — by the pool instead of one by one
    const bool grown = max_addr > memory.len;
    if (max_addr > memory.len) {
        const u64 from = memory.len, chunk = 1u << 15;
        u32* mp = memory.p;
        vm_parallel_for((max_addr - from + chunk - 1) / chunk, n_threads, [&](u64 c) {
            const u64 e = std::min(max_addr, from + (c + 1) * chunk);
            for (u64 i = from + c * chunk; i < e; i++) mp[i] = UNDEF;
        });
        memory.len = max_addr;
    }
    const double tp1 = vm_now_ms();
    for (u64 i = 1; i <= n_iters; i++) {  // write_call_frame
        const u64 f = batch.batch_fp + i * stride;
        const u64 iter_val = i < n_iters ? start_value + i : end_value;
        if (!mm.set(f, return_pc, err) || !mm.set(f + 1, saved_fp, err) || !mm.set(f + 2, kb::to_monty((u32)(iter_val % kb::P)), err)) return false;
        for (u32 j = 1; j < batch.n_args; j++)
            if (!mm.set(f + 2 + j, args[j], err)) return false;
    }
    const u64 n_par = n_iters - 1;
    const u64 split_at = batch.batch_fp + stride;

    // Every pool thread appends the logs of the segments it runs to its own persistent buffers (ThreadLog: grow-only, reused by the
    // next run); a segment remembers where its part lies.  No allocation per segment: 10^4 malloc / free pairs per run (and the
    // page trimming they trigger under 128 threads) were the source of sporadic 10-30 ms stalls.
    struct Seg {
        ThreadLog* log = nullptr;
        size_t o_cyc = 0, n_cyc = 0, o_pos = 0, n_pos = 0, o_ext = 0, n_ext = 0, o_pend = 0, n_pend = 0, o_def = 0, n_def = 0;
        u64 n_add = 0, n_mul = 0, n_deref = 0, n_jump = 0;
        Err err;
    };
    const double tb0 = vm_now_ms();
    thread_logs_reset(pool);
    std::vector<Seg> segs(n_par);
    u32* base = memory.data();
    vm_parallel_for(n_par, n_threads, [&](u64 i) {
        Seg& s = segs[i];
        ThreadLog& L = thread_log();
        Trace& t = L.tr;
        s.log = &L;
        s.o_cyc = t.pcs.size(), s.o_pos = t.pos.size(), s.o_ext = t.ext.size(), s.o_pend = t.pending.size(), s.o_def = L.deferred.size();
        const u64 a0 = t.n_add, m0 = t.n_mul, d0 = t.n_deref, j0 = t.n_jump;
        const u64 seg_start = split_at + i * stride;
        SegMem sm{base, split_at, base + seg_start, seg_start, stride, &L.deferred};
        Cursors c = cur;
        for (size_t k = 0; k < c.index.size(); k++) c.index[k] += i * per_iter[k];
        Machine<SegMem> m(bc, w, sm, t, c);
        m.pc = batch.batch_pc;
        m.fp = batch.batch_fp + (i + 1) * stride;
        m.ap = m.fp + batch.frame_size;
        Machine<SegMem>::Batch inner;
        try {  // (a pool thread has no caller to unwind to: an allocation failure becomes the segment's error)
            const int rc = m.run(true, batch.batch_pc, inner);
            if (rc != 1 && !m.err.set) m.err.raise(rc == 0 ? "Panic: a parallel segment reached the end of the program" : "Panic: nested parallel batch");
        } catch (const std::bad_alloc&) {
            m.err.set = false;
            m.err.raise("OutOfMemory (host allocation)");
        }
        s.err = m.err;
        s.n_cyc = t.pcs.size() - s.o_cyc, s.n_pos = t.pos.size() - s.o_pos, s.n_ext = t.ext.size() - s.o_ext;
        s.n_pend = t.pending.size() - s.o_pend, s.n_def = L.deferred.size() - s.o_def;
        s.n_add = t.n_add - a0, s.n_mul = t.n_mul - m0, s.n_deref = t.n_deref - d0, s.n_jump = t.n_jump - j0;
    });
    for (u64 i = 0; i < n_par; i++)
        if (segs[i].err.set) {
            err.raise("ParallelSegmentFailed(%llu, %s)", (unsigned long long)(i + 1), segs[i].err.msg.c_str());
            return false;
        }
    const double tb1 = vm_now_ms();
    // Trace::merge in iteration order (a parallel copy out of the thread logs), then the deferred writes
    size_t n_cyc = trace.pcs.size(), n_pos = trace.pos.size(), n_ext = trace.ext.size(), n_pend = trace.pending.size(), n_def = 0;
    std::vector<size_t> o_cyc(n_par), o_pos(n_par), o_ext(n_par), o_pend(n_par), o_def(n_par);
    for (u64 i = 0; i < n_par; i++) {
        o_cyc[i] = n_cyc, o_pos[i] = n_pos, o_ext[i] = n_ext, o_pend[i] = n_pend, o_def[i] = n_def;
        n_cyc += segs[i].n_cyc, n_pos += segs[i].n_pos, n_ext += segs[i].n_ext, n_pend += segs[i].n_pend, n_def += segs[i].n_def;
        trace.n_add += segs[i].n_add, trace.n_mul += segs[i].n_mul, trace.n_deref += segs[i].n_deref, trace.n_jump += segs[i].n_jump;
    }
    trace.pcs.extend(n_cyc - trace.pcs.size()), trace.fps.extend(n_cyc - trace.fps.size()), trace.pos.extend(n_pos - trace.pos.size());
    trace.ext.extend(n_ext - trace.ext.size()), trace.pending.extend(n_pend - trace.pending.size());
    UVec<std::pair<u64, u32>>& all_def = batch_deferred(pool);
    all_def.n = 0;
    all_def.extend(n_def);
    vm_parallel_for(n_par, n_threads, [&](u64 i) {
        const Seg& sg = segs[i];
        const Trace& t = sg.log->tr;
        if (sg.n_cyc) memcpy(&trace.pcs[o_cyc[i]], t.pcs.data() + sg.o_cyc, sg.n_cyc * 4), memcpy(&trace.fps[o_cyc[i]], t.fps.data() + sg.o_cyc, sg.n_cyc * 4);
        if (sg.n_pos) memcpy(&trace.pos[o_pos[i]], t.pos.data() + sg.o_pos, sg.n_pos * 4);
        if (sg.n_ext) memcpy(&trace.ext[o_ext[i]], t.ext.data() + sg.o_ext, sg.n_ext * 4);
        if (sg.n_pend) memcpy((void*)&trace.pending[o_pend[i]], t.pending.data() + sg.o_pend, sg.n_pend * sizeof(std::pair<u64, u64>));
        if (sg.n_def) memcpy((void*)&all_def[o_def[i]], sg.log->deferred.data() + sg.o_def, sg.n_def * sizeof(std::pair<u64, u32>));
    });
    for (size_t k = 0; k < n_def; k++)
        if (!mm.set(all_def[k].first, all_def[k].second, err)) return false;
    for (size_t k = 0; k < cur.index.size(); k++) cur.index[k] += n_par * per_iter[k];
    pc = batch.batch_pc;
    fp = batch.batch_fp + n_iters * stride;
    ap = fp + batch.frame_size;
    if (trim_last_frame && grown && memory.len == max_addr) trim_to_defined(memory, fp + 2 + batch.n_args, max_addr);
    if (vm_times())
        fprintf(stderr, "[vm] batch of %llu segments: resize %.2f ms, call frames %.2f ms, run %.2f ms, merge + deferred writes %.2f ms\n",
                (unsigned long long)n_par, tp1 - tp0, tb0 - tp1, tb1 - tb0, vm_now_ms() - tb1);
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// parallel batches on the device (lm_vm_device.hip): what the host keeps of a run whose segments were interpreted by wavefronts
// ---------------------------------------------------------------------------------------------------------------------
struct DevBatch {
    VmSegArgs a;  // the device slots and their capacities
    u64 n_par = 0;
    u64* d_offsets = nullptr;  // per segment: exclusive prefix sums of (cycles, Poseidon calls, extension rows, pending derefs)
    u64 tot[4] = {0, 0, 0, 0};
    size_t host_cyc_at = 0, host_pos_at = 0, host_ext_at = 0, host_pend_at = 0;  // where the batch sits in the host logs (entries)
    u64 win_lo = 0, win_hi = 0;                                                    // the segment frames: valid in the device image only
    std::vector<u32*> owned;
};
struct DevRun {
    lm_ctx* ctx = nullptr;
    u64 ctx_uid = 0;  // the context may be destroyed before the execution is released (a garbage collector): its pool went with it
    const VmInstr* d_code = nullptr;  // cached in the context under the bytecode's id
    const u32* d_hint_begin = nullptr;
    const VmHintRec* d_hints = nullptr;
    u32* d_wit_data = nullptr;  // this run's hint streams
    u64 *d_wit_off = nullptr, *d_wit_names = nullptr;
    u32* d_image = nullptr;  // the memory image, VM_UNDEF = None
    u64 image_cap = 0;
    std::vector<DevBatch> batches;
    bool windows_open = false;
    // after finalisation: the complete log, resident
    u32 *d_pcs = nullptr, *d_fps = nullptr, *d_pos = nullptr, *d_ext = nullptr;
    u64 n_cycles = 0, n_pos = 0, n_ext = 0, image_len = 0;
    u64 n_add = 0, n_mul = 0, n_deref = 0, n_jump = 0;  // of the segments
    bool finalized = false, host_valid = false;
    std::vector<u32*> owned;
    std::vector<std::vector<u64>> keep64;  // host staging that asynchronous uploads read: kept until the next synchronisation
    std::vector<std::vector<u32>> keep32;
};
void dev_run_free(DevRun* d) {
    if (!d) return;
    // (context gone: its device pool went with it.)  May run on any thread (a garbage collector): the registry lock is held across the
    // frees, so the context cannot be destroyed under them, and the pool has its own lock (round-4 advisor finding)
    lm_ctx_with_live(d->ctx_uid, d->ctx, [](lm_ctx* ctx, void* arg) {
        DevRun* r = (DevRun*)arg;
        for (DevBatch& b : r->batches)
            for (u32* p : b.owned) lm_free(ctx, p);
        for (u32* p : r->owned) lm_free(ctx, p);
    }, d);
    delete d;
}

namespace {
bool vm_device_enabled() {
    static const bool on = getenv("LM_VM_HOST") == nullptr;  // LM_VM_HOST=1: parallel batches on the host thread pool (A/B measurements)
    return on;
}
template <class T>
bool dev_alloc(DevRun& D, std::vector<u32*>& owner, u64 n_elems, T** out) {
    u32* p = nullptr;
    const u64 words = (n_elems * sizeof(T) + 3) / 4;
    if (lm_malloc(D.ctx, words ? words : 1, &p) != LM_OK) return false;
    owner.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return true;
}
// device copies of the decoded program: one per (bytecode, context), owned by the context
bool dev_program(DevRun& D, const lmh_bytecode& bc) {
    const u64 key = bc.uid << 4;
    D.d_code = (const VmInstr*)lm_ctx_cache_get(D.ctx, key | 1);
    D.d_hint_begin = (const u32*)lm_ctx_cache_get(D.ctx, key | 2);
    D.d_hints = (const VmHintRec*)lm_ctx_cache_get(D.ctx, key | 3);
    if (D.d_code && D.d_hint_begin && D.d_hints) return true;
    u32 *c = nullptr, *hb = nullptr, *h = nullptr;
    const u64 wc = (bc.code.size() * sizeof(VmInstr) + 3) / 4, wh = (bc.hints.size() * sizeof(VmHintRec) + 3) / 4;
    if (lm_malloc(D.ctx, wc ? wc : 1, &c) || lm_malloc(D.ctx, bc.hint_begin.size(), &hb) || lm_malloc(D.ctx, wh ? wh : 1, &h) ||
        vm_dev_upload(D.ctx, c, bc.code.data(), bc.code.size() * sizeof(VmInstr)) || vm_dev_upload(D.ctx, hb, bc.hint_begin.data(), bc.hint_begin.size() * 4) ||
        vm_dev_upload(D.ctx, h, bc.hints.data(), bc.hints.size() * sizeof(VmHintRec)) || lm_sync(D.ctx)) {
        lm_free(D.ctx, c), lm_free(D.ctx, hb), lm_free(D.ctx, h);  // (lm_free(nullptr) is a no-op)
        return false;
    }
    lm_ctx_cache_put(D.ctx, key | 1, c), lm_ctx_cache_put(D.ctx, key | 2, hb), lm_ctx_cache_put(D.ctx, key | 3, h);
    D.d_code = (const VmInstr*)c, D.d_hint_begin = hb, D.d_hints = (const VmHintRec*)h;
    return true;
}
bool dev_witness(DevRun& D, const lm_vm_witness* w) {
    if (D.d_wit_names) return true;
    const u64 n_entries = w->n_names ? w->name_entry_begin[w->n_names] : 0;
    const u64 n_words = n_entries ? w->entry_offset[n_entries] : 0;
    if (!dev_alloc(D, D.owned, n_words, &D.d_wit_data) || !dev_alloc(D, D.owned, n_entries + 1, &D.d_wit_off) ||
        !dev_alloc(D, D.owned, (u64)w->n_names + 1, &D.d_wit_names))
        return false;
    static const u64 zero = 0;
    return vm_dev_upload(D.ctx, D.d_wit_data, w->data, n_words * 4) == LM_OK &&
           vm_dev_upload(D.ctx, D.d_wit_off, n_entries ? w->entry_offset : &zero, (n_entries + 1) * 8) == LM_OK &&
           vm_dev_upload(D.ctx, D.d_wit_names, w->n_names ? w->name_entry_begin : &zero, ((u64)w->n_names + 1) * 8) == LM_OK;
}
// [lo, hi) of the arena -> the device image, skipping the frames of earlier device batches (current in the image only)
bool dev_upload_host_owned(DevRun& D, MemBuf& memory, u64 lo, u64 hi) {
    u64 at = lo;
    auto up = [&](u64 a, u64 b) { return a >= b || vm_dev_upload(D.ctx, D.d_image + a, memory.p + a, (b - a) * 4) == LM_OK; };
    if (D.windows_open)
        for (const DevBatch& b : D.batches) {  // (ascending: memory grows upwards)
            if (b.win_hi <= at || b.win_lo >= hi) continue;
            if (!up(at, std::min(b.win_lo, hi))) return false;
            at = std::max(at, b.win_hi);
        }
    return up(at, hi);
}
bool dev_image_reserve(DevRun& D, u64 need) {
    if (need <= D.image_cap) return true;
    const u64 cap = need + (1u << 16);
    u32* n = nullptr;
    if (lm_malloc(D.ctx, cap, &n) != LM_OK) return false;
    if ((D.d_image && lm_copy_d2d(D.ctx, n, D.d_image, D.image_cap) != LM_OK) || vm_dev_fill(D.ctx, n + D.image_cap, VM_UNDEF, cap - D.image_cap) != LM_OK) {
        lm_free(D.ctx, n);
        return false;
    }
    if (D.d_image) {
        for (u32*& p : D.owned)
            if (p == D.d_image) p = n;
        lm_free(D.ctx, D.d_image);  // (stream-ordered pool: the copy above is queued before any reuse)
    } else
        D.owned.push_back(n);
    D.d_image = n, D.image_cap = cap;
    return true;
}
// the frames of every device batch back into the arena (a sequential part of the program, or a host batch, reads them)
bool dev_close_windows(DevRun& D, MemBuf& memory) {
    if (!D.windows_open) return true;
    D.windows_open = false;
    for (const DevBatch& b : D.batches)
        if (vm_dev_download(D.ctx, memory.p + b.win_lo, D.d_image + b.win_lo, (b.win_hi - b.win_lo) * 4) != LM_OK) return false;
    return true;
}

enum { DEV_DONE = 0, DEV_FALLBACK = 1, DEV_ERROR = 2 };
// handle_parallel_batch with the segments on the device.  DEV_FALLBACK: nothing of the run's state has changed in a way the host
// batch would notice — the caller runs handle_parallel_batch (which also reports every RunnerError: the device never does).
int device_batch(const lmh_bytecode& bc, const lm_vm_witness* witness, MemBuf& memory, Trace& trace, Cursors& cur, u64& pc, u64& fp, u64& ap,
                 const Machine<MainMem>::Batch& batch, u32 n_threads, DevRun& D, std::string& why, bool trim_last_frame = false,
                 const VmLate* late = nullptr) {
    MainMem mm{memory};
    Err scratch;
    const double t0 = vm_now_ms();
    auto get = [&](u64 at) -> u32 {
        const u32 v = mm.peek(at);
        if (v == UNDEF) scratch.raise("undefined");
        return v;
    };
    const u32 sv = get(batch.batch_fp + 2);
    const u32 ev = batch.end_mode == LM_VM_ARG_CONST ? 0 : get(batch.batch_fp + batch.end_value);
    why = "an undefined loop bound / call frame cell";
    if (scratch.set) return DEV_FALLBACK;
    const u64 start_value = kb::from_monty(sv), end_value = batch.end_mode == LM_VM_ARG_CONST ? batch.end_value : kb::from_monty(ev);
    why = "fewer than two iterations";
    if (end_value <= start_value + 1) return DEV_FALLBACK;
    const u64 n_iters = end_value - start_value, n_par = n_iters - 1, stride = fp - batch.batch_fp;
    const u32 return_pc = get(fp), saved_fp = get(fp + 1);
    {
        char buf[200];
        snprintf(buf, sizeof buf, "batch shape outside the device limits (%llu segments [32, 2^24), frame %llu words (<= %u), %u call-frame arguments (1..%u), %u hint names (<= %u))",
                 (unsigned long long)n_par, (unsigned long long)stride, VM_DEV_MAX_STRIDE, batch.n_args, VM_DEV_MAX_ARGS, bc.n_names, VM_DEV_MAX_NAMES);
        why = buf;
    }
    if (scratch.set || batch.n_args > VM_DEV_MAX_ARGS || batch.n_args == 0 || bc.n_names > VM_DEV_MAX_NAMES || stride == 0 || stride > VM_DEV_MAX_STRIDE ||
        n_par < 32 || n_par >= (1u << 24) || fp <= batch.batch_fp)
        return DEV_FALLBACK;
    u32 args[VM_DEV_MAX_ARGS];
    for (u32 i = 0; i < batch.n_args; i++) args[i] = get(batch.batch_fp + 2 + i);
    why = "an undefined call-frame argument";
    if (scratch.set) return DEV_FALLBACK;
    const u64 max_addr = batch.batch_fp + (n_iters + 1) * stride, split_at = batch.batch_fp + stride, frames_end = batch.batch_fp + n_iters * stride;
    why = "the batch's frames exceed the memory limit";
    if (max_addr > MAX_MEMORY) return DEV_FALLBACK;
    why = "the call frame of the last iteration conflicts with memory / a deferred check failed";
    std::vector<u64> per_iter(cur.index.size());
    for (size_t k = 0; k < per_iter.size(); k++) per_iter[k] = cur.index[k] - batch.hint_indices_at_start[k];
    if (memory.touch_failed) return DEV_ERROR;
    if (!dev_program(D, bc) || !dev_witness(D, witness)) return DEV_ERROR;
    // Late input words (VmLate): the device copy of the hint streams went up while a helper thread may still have been writing them
    // (round-5 advisor finding: only the sequential runner honoured the late ranges).  A batch whose segments CONSUME entries of a stream
    // that holds a late range must read the final words: wait for the helper and send the range again, stream-ordered in front of the
    // segment kernels.  (The aggregation program reads its late words in its sequential head: per_iter is 0 for that stream.)
    if (late && late->n_ranges) {
        bool waited = false;
        for (u32 r = 0; r < late->n_ranges; r++) {
            const u64 w0 = late->first_word[r];
            for (u32 k = 0; k < witness->n_names; k++) {
                if (!per_iter[k]) continue;
                const u64 e0 = witness->name_entry_begin[k], e1 = witness->name_entry_begin[k + 1];
                if (e0 == e1 || w0 < witness->entry_offset[e0] || w0 >= witness->entry_offset[e1]) continue;
                if (!waited && late->wait) late->wait();
                waited = true;
                if (vm_dev_upload(D.ctx, D.d_wit_data + w0, witness->data + w0, 4ull * late->n_words[r]) != LM_OK) return DEV_ERROR;
            }
        }
    }

    // ---- host side of the batch: the memory grows to max_addr, the frame the sequential runner continues in gets its call frame.
    // The frames of the segments are NOT touched here: they exist in the device image only (window) until something reads them.
    const u64 old_len = memory.len;
    auto fill_host = [&](u64 a, u64 b) {  // [a, b) <- None, on the pool
        if (a >= b) return;
        const u64 chunk = 1u << 15;
        u32* mp = memory.p;
        vm_parallel_for((b - a + chunk - 1) / chunk, n_threads, [&](u64 c) {
            const u64 e = std::min(b, a + (c + 1) * chunk);
            for (u64 i = a + c * chunk; i < e; i++) mp[i] = UNDEF;
        });
    };
    bool host_grown = false;
    auto fallback = [&](DevBatch* b) {
        if (b) {
            for (u32* p : b->owned) lm_free(D.ctx, p);
            // the abandoned batch's deferred writes may have defined image cells above the memory's length, where the uploads of a later
            // batch (which cover [0, memory.len)) do not reach: back to None
            if (max_addr < D.image_cap && vm_dev_fill(D.ctx, D.d_image + max_addr, VM_UNDEF, D.image_cap - max_addr) != LM_OK) return (int)DEV_ERROR;
        }
        // the host batch grows the memory itself (and fills what it adds): back to the length this batch found, so that it also knows
        // whether IT was the one that grew the memory (trim_last_frame)
        if (host_grown) memory.len = old_len;
        if (!dev_close_windows(D, memory)) return (int)DEV_ERROR;
        memory.dev_lo = memory.dev_hi = 0;
        return (int)DEV_FALLBACK;
    };
    fill_host(std::max(old_len, frames_end), max_addr);
    const bool grown = max_addr > memory.len;
    if (max_addr > memory.len) memory.len = max_addr;
    host_grown = true;
    {  // write_call_frame for the last iteration (the loop comes back to the sequential runner in it)
        Err e;
        const u64 f = frames_end;
        const u64 iter_val = end_value;
        bool ok = mm.set(f, return_pc, e) && mm.set(f + 1, saved_fp, e) && mm.set(f + 2, kb::to_monty((u32)(iter_val % kb::P)), e);
        for (u32 j = 1; j < batch.n_args && ok; j++) ok = mm.set(f + 2 + j, args[j], e);
        if (!ok) return fallback(nullptr);
    }
    // ---- device image: what the host holds, minus the not yet existing frames
    {
        VmRegion reg[5];
        reg[0] = {(void*)memory.p, (size_t)MAX_MEMORY * 4};
        vm_ensure_pinned(reg[0], (size_t)(memory.len + 24) * 4);
    }
    // a pending call whose output cells lie at or above split_at (iteration 0 hashed into the next frame) would define them in the arena
    // behind the window, where the end-of-run upload does not look: such calls are executed before the image goes up
    for (const MemBuf::LazyCall& c : memory.lazy)
        if (!c.done && c.kind == 0 && c.res + c.n_out > split_at) {
            memory.lazy_drain();
            break;
        }
    if (memory.lazy_failed) return fallback(nullptr);
    if (!dev_image_reserve(D, memory.len)) return DEV_ERROR;
    // the output cells of the calls still pending go up POISONED (VM_PENDING), not as None: a segment that touches one in any way ends
    // there (several reads tolerate None: a DEREF with an unknown result, the ADD / MUL / ExtensionOp solvers).  The arena cells hold
    // the poison only until lazy_drain below, which writes every digest; no host read of the arena lies in between.
    if (memory.lazy_open)
        for (const MemBuf::LazyCall& c : memory.lazy)
            if (!c.done && c.kind == 0)
                for (u32 j = 0; j < c.n_out; j++) memory.p[c.res + j] = VM_PENDING;
    if (!dev_upload_host_owned(D, memory, 0, std::max(split_at, std::min(old_len, frames_end))) || !dev_upload_host_owned(D, memory, frames_end, memory.len))
        return DEV_ERROR;
    // ---- slots sized from what iteration 0 logged on the host
    DevBatch B;
    B.n_par = n_par;
    B.host_cyc_at = trace.pcs.size(), B.host_pos_at = trace.pos.size() / LM_VM_POSEIDON_CALL_WORDS;
    B.host_ext_at = trace.ext.size() / LM_VM_EXTENSION_ROW_WORDS, B.host_pend_at = trace.pending.size();
    B.win_lo = split_at, B.win_hi = frames_end;
    VmSegArgs& a = B.a;
    memset(&a, 0, sizeof a);
    a.cap_cyc = (u32)std::min<u64>(2 * (B.host_cyc_at - batch.cyc_at_arm) + 64, 1u << 22);
    a.cap_pos = (u32)std::min<u64>(2 * (B.host_pos_at - batch.pos_at_arm) + 16, 1u << 20);
    a.cap_ext = (u32)std::min<u64>(2 * (B.host_ext_at - batch.ext_at_arm) + 16, 1u << 20);
    a.cap_pend = (u32)std::min<u64>(2 * (B.host_pend_at - batch.pend_at_arm) + 16, 1u << 20);
    a.cap_def = 64 + 2 * batch.n_args;
    // cells outside the frames a segment may define (the XMSS loop: 1; a query of the recursion program: its fold, 5, and its circle value)
    const u32 dirty_cap = (u32)(12 * n_par + 64);
    u32* d_summary = nullptr;
    if (!dev_alloc(D, B.owned, n_par * a.cap_cyc, &a.pcs) || !dev_alloc(D, B.owned, n_par * a.cap_cyc, &a.fps) ||
        !dev_alloc(D, B.owned, n_par * a.cap_pos * LM_VM_POSEIDON_CALL_WORDS, &a.pos) ||
        !dev_alloc(D, B.owned, n_par * a.cap_ext * LM_VM_EXTENSION_ROW_WORDS, &a.ext) || !dev_alloc(D, B.owned, n_par * a.cap_pend * 2, &a.pend) ||
        !dev_alloc(D, B.owned, n_par * a.cap_def * 2, &a.def) || !dev_alloc(D, B.owned, n_par * VM_SEG_WORDS, &a.counts) ||
        !dev_alloc(D, B.owned, n_par * 4, &B.d_offsets) || !dev_alloc(D, B.owned, (u64)VM_SUMMARY_WORDS + 2 * dirty_cap, &d_summary)) {
        for (u32* p : B.owned) lm_free(D.ctx, p);
        return DEV_ERROR;
    }
    a.code = D.d_code, a.hint_begin = D.d_hint_begin, a.hints = D.d_hints;
    for (size_t k = 0; k < cur.index.size(); k++) a.cur_index[k] = cur.index[k], a.per_iter[k] = per_iter[k];  // (kernel arguments: <= 64 names)
    a.n_instructions = (u32)bc.n_instructions, a.ending_pc = bc.ending_pc, a.n_hints = (u32)bc.hints.size();
    a.prefix_cache = (u32)std::min<u64>(split_at, VM_DEV_PREFIX_CACHE);
#ifdef LM_VM_DEBUG  // timing experiments (tools/vm_device_probe.py): the knobs make results WRONG, so they exist in debug builds only
    if (const char* e = getenv("LM_VM_DBG")) a.dbg = (u32)strtoul(e, nullptr, 10);
#endif
    a.wit_data = D.d_wit_data, a.wit_entry_offset = D.d_wit_off, a.wit_name_begin = D.d_wit_names;
    a.n_names = bc.n_names;
    a.image = D.d_image, a.init_len = old_len, a.split_at = split_at, a.stride = stride, a.batch_fp = batch.batch_fp, a.frame_size = batch.frame_size;
    a.batch_pc = (u32)batch.batch_pc, a.return_pc_m = return_pc, a.saved_fp_m = saved_fp, a.start_value = start_value, a.n_args = batch.n_args;
    memcpy(a.args_m, args, sizeof(u32) * batch.n_args);
    a.coop_tab = vm_dev_coop_table(D.ctx);
    {
        const kb::FrobeniusTable& ft = kb::frobenius_table();
        for (int i = 0; i < 5; i++) a.frob[i] = ft.img[i];
    }
    std::vector<u32> summary((size_t)VM_SUMMARY_WORDS + 2 * dirty_cap);
    const double t1 = vm_now_ms();
    a.summary = d_summary;  // (zeroed by the segment kernel's first workgroup: no fill command in front of it)
    bool launched = (!memory.lazy_open || vm_dev_mark(D.ctx) == LM_OK) && vm_dev_segments(D.ctx, a, n_par) == LM_OK &&
                    vm_dev_apply_deferred(D.ctx, a, n_par, D.image_cap, split_at, frames_end, B.d_offsets, d_summary, dirty_cap) == LM_OK;
    // the deferred Poseidon calls and checks of the sequential part (MemBuf) are executed while the segments run: their results reach the
    // image with the host-owned cells at the end of the run (device_finalize), the segments saw None there
    const double t_lazy0 = vm_now_ms();
    const u64 n_lazy = memory.lazy_open;
    if (launched && n_lazy) launched = vm_dev_wait_mark(D.ctx) == LM_OK;  // (the image upload has read the arena: the segments see None, not a race)
    memory.lazy_drain();
    const double t_lazy1 = vm_now_ms();
    if (!launched || vm_dev_download(D.ctx, summary.data(), d_summary, summary.size() * 4)) {
        for (u32* p : B.owned) lm_free(D.ctx, p);
        return DEV_ERROR;
    }
    const double t2 = vm_now_ms();
    if (summary[0] || summary[2] || summary[3] || summary[1] > dirty_cap) {
        char buf[256];
        snprintf(buf, sizeof buf, "device batch of %llu segments handed back: %u conflicting deferred writes, %u beyond the image, first failed segment %u "
                                  "(code %u, pc %u, aux %u), %u dirty cells", (unsigned long long)n_par, summary[0], summary[2], summary[3], summary[4], summary[5],
                 summary[6], summary[1]);
        why = buf;
        if (vm_times()) fprintf(stderr, "[vm] %s\n", buf);
        return fallback(&B);
    }
    // ---- success: mirror the cells outside the frames that deferred writes defined, then commit
    for (u32 k = 0; k < summary[1]; k++) {
        Err e;
        if (!mm.set(summary[VM_SUMMARY_WORDS + 2 * k], summary[VM_SUMMARY_WORDS + 2 * k + 1], e)) return fallback(&B);  // (cannot happen: the device image agreed)
    }
    const u64* tot = reinterpret_cast<const u64*>(summary.data() + 8);
    for (int k = 0; k < 4; k++) B.tot[k] = tot[k];
    D.n_add += tot[4], D.n_mul += tot[5], D.n_deref += tot[6], D.n_jump += tot[7];
    D.batches.push_back(B);
    D.windows_open = true;
    memory.dev_lo = D.batches.front().win_lo, memory.dev_hi = frames_end;  // hull of every open window
    for (size_t k = 0; k < cur.index.size(); k++) cur.index[k] += n_par * per_iter[k];
    pc = batch.batch_pc;
    fp = batch.batch_fp + n_iters * stride;
    ap = fp + batch.frame_size;
    if (trim_last_frame && grown && memory.len == max_addr) trim_to_defined(memory, fp + 2 + batch.n_args, max_addr);  // (see handle_parallel_batch)
    if (vm_times())
        fprintf(stderr, "[vm] device batch of %llu segments: host preparation + uploads %.2f ms, segments + deferred writes + summary %.2f ms, commit %.2f ms "
                        "(%llu cycles, %llu Poseidon calls, %u dirty cells; %llu deferred host Poseidon calls executed meanwhile in %.2f ms)\n",
                (unsigned long long)n_par, t1 - t0, t2 - t1, vm_now_ms() - t2, (unsigned long long)tot[0], (unsigned long long)tot[1], summary[1],
                (unsigned long long)n_lazy, t_lazy1 - t_lazy0);
    return DEV_DONE;
}

// End of a run with device batches: the complete log and the memory image are assembled in HBM — host parts uploaded around the
// spliced segment logs —, then resolve_deref_hints runs there.  false + anomaly: the caller repeats the run on the host.
bool device_finalize(lmh_execution* ex, DevRun& D, bool& anomaly) {
    anomaly = false;
    Trace& tr = ex->tr;
    MemBuf& memory = ex->memory;
    const u64 L = memory.len;
    if (memory.touch_failed) return false;
    if (!dev_image_reserve(D, L)) return false;
    {
        VmRegion reg{(void*)memory.p, (size_t)MAX_MEMORY * 4};
        vm_ensure_pinned(reg, (size_t)(L + 24) * 4);
    }
    if (!dev_upload_host_owned(D, memory, 0, L)) return false;
    u64 n_cyc = tr.pcs.size(), n_pos = tr.pos.size() / LM_VM_POSEIDON_CALL_WORDS, n_ext = tr.ext.size() / LM_VM_EXTENSION_ROW_WORDS, n_pend = tr.pending.size();
    for (const DevBatch& b : D.batches) n_cyc += b.tot[0], n_pos += b.tot[1], n_ext += b.tot[2], n_pend += b.tot[3];
    u32* d_pend = nullptr;
    uint8_t* d_status = nullptr;
    u32* d_info = nullptr;
    std::vector<u32*> tmp;
    auto drop_tmp = [&] {
        for (u32* p : tmp) lm_free(D.ctx, p);
    };
    if (!dev_alloc(D, D.owned, n_cyc, &D.d_pcs) || !dev_alloc(D, D.owned, n_cyc, &D.d_fps) || !dev_alloc(D, D.owned, n_pos * LM_VM_POSEIDON_CALL_WORDS, &D.d_pos) ||
        !dev_alloc(D, D.owned, n_ext * LM_VM_EXTENSION_ROW_WORDS, &D.d_ext) || !dev_alloc(D, tmp, n_pend * 2, &d_pend) || !dev_alloc(D, tmp, n_pend, &d_status) ||
        !dev_alloc(D, tmp, (u64)VM_RESOLVE_INFO_WORDS, &d_info)) {
        drop_tmp();
        return false;
    }
    // the host's pending list as (target, src) words
    D.keep32.emplace_back(tr.pending.size() * 2);
    std::vector<u32>& hp = D.keep32.back();
    for (size_t i = 0; i < tr.pending.size(); i++) hp[2 * i] = (u32)tr.pending[i].first, hp[2 * i + 1] = (u32)tr.pending[i].second;
    // (the host parts of the log — a few thousand cycles of the sequential head and tail — are uploaded from pageable memory: registering
    // such small buffers costs more than it saves, and registered chunks of the malloc heap were behind an intermittent GPU memory
    // access fault in the test suite; only the arena, a private mapping that lives across runs, is registered)
    bool ok = true;
    u64 h_cyc = 0, h_pos = 0, h_ext = 0, h_pend = 0;  // host entries consumed so far
    u64 o_cyc = 0, o_pos = 0, o_ext = 0, o_pend = 0;  // positions in the final arrays
    std::vector<VmPart> parts;  // every host piece and the resolver's zeroed scratch: one launch (vm_dev_place)
    auto host_part = [&](u64 cyc_to, u64 pos_to, u64 ext_to, u64 pend_to) {
        parts.push_back({D.d_pcs + o_cyc, tr.pcs.data() + h_cyc, cyc_to - h_cyc});
        parts.push_back({D.d_fps + o_cyc, tr.fps.data() + h_cyc, cyc_to - h_cyc});
        parts.push_back({D.d_pos + o_pos * LM_VM_POSEIDON_CALL_WORDS, tr.pos.data() + h_pos * LM_VM_POSEIDON_CALL_WORDS, (pos_to - h_pos) * LM_VM_POSEIDON_CALL_WORDS});
        parts.push_back({D.d_ext + o_ext * LM_VM_EXTENSION_ROW_WORDS, tr.ext.data() + h_ext * LM_VM_EXTENSION_ROW_WORDS, (ext_to - h_ext) * LM_VM_EXTENSION_ROW_WORDS});
        parts.push_back({d_pend + o_pend * 2, hp.data() + h_pend * 2, (pend_to - h_pend) * 2});
        o_cyc += cyc_to - h_cyc, o_pos += pos_to - h_pos, o_ext += ext_to - h_ext, o_pend += pend_to - h_pend;
        h_cyc = cyc_to, h_pos = pos_to, h_ext = ext_to, h_pend = pend_to;
    };
    for (const DevBatch& b : D.batches) {
        host_part(b.host_cyc_at, b.host_pos_at, b.host_ext_at, b.host_pend_at);
        const u64 base[4] = {o_cyc, o_pos, o_ext, o_pend};
        ok = ok && vm_dev_splice(D.ctx, b.a, b.n_par, b.d_offsets, base, D.d_pcs, D.d_fps, D.d_pos, D.d_ext, d_pend) == LM_OK;
        o_cyc += b.tot[0], o_pos += b.tot[1], o_ext += b.tot[2], o_pend += b.tot[3];
    }
    host_part(tr.pcs.size(), tr.pos.size() / LM_VM_POSEIDON_CALL_WORDS, tr.ext.size() / LM_VM_EXTENSION_ROW_WORDS, tr.pending.size());
    // resolve_deref_hints on the assembled image
    parts.push_back({(u32*)d_status, nullptr, (n_pend + 3) / 4 ? (n_pend + 3) / 4 : 1});
    parts.push_back({d_info, nullptr, (u64)VM_RESOLVE_INFO_WORDS});
    ok = ok && vm_dev_place(D.ctx, parts.data(), (u32)parts.size()) == LM_OK;
    u32 info[VM_RESOLVE_INFO_WORDS] = {0};
    for (u32 first = 0; ok && n_pend;) {
        const u32 rounds = 3;
        if (first + rounds >= VM_RESOLVE_INFO_WORDS - 1) {  // a dependency chain deeper than the counters: the host's sequential loop takes it
            anomaly = true;
            break;
        }
        ok = vm_dev_resolve(D.ctx, D.d_image, L, d_pend, n_pend, d_status, d_info, first, rounds) == LM_OK &&
             vm_dev_download(D.ctx, info, d_info, sizeof info) == LM_OK;
        if (!ok) break;
        if (info[0]) anomaly = true;
        if (anomaly || info[1 + first + rounds - 1] == 0) break;
        first += rounds;
    }
    if (ok && !n_pend) ok = lm_sync(D.ctx) == LM_OK;
    D.keep32.clear();
    drop_tmp();
    for (DevBatch& b : D.batches) {
        for (u32* p : b.owned) lm_free(D.ctx, p);
        b.owned.clear();
    }
    if (!ok || anomaly) return false;
    D.n_cycles = n_cyc, D.n_pos = n_pos, D.n_ext = n_ext, D.image_len = L;
    D.finalized = true;
    return true;
}

// the host view of a device run (lmh_execution_view): one download of the image and of the logs
bool device_materialize(lmh_execution* ex) {
    DevRun& D = *ex->dev;
    if (D.host_valid) return true;
    if (lm_ctx_by_uid(D.ctx_uid) != D.ctx) {
        lm_set_error("lmh_execution_view: the context this execution is resident on has been destroyed");
        return false;
    }
    Trace& tr = ex->tr;
    const u64 L = D.image_len;
    tr.pcs.n = tr.fps.n = tr.pos.n = tr.ext.n = 0;
    tr.pcs.extend(D.n_cycles), tr.fps.extend(D.n_cycles), tr.pos.extend(D.n_pos * LM_VM_POSEIDON_CALL_WORDS), tr.ext.extend(D.n_ext * LM_VM_EXTENSION_ROW_WORDS);
    if (vm_dev_download(D.ctx, ex->memory.p, D.d_image, L * 4) || vm_dev_download(D.ctx, tr.pcs.data(), D.d_pcs, D.n_cycles * 4) ||
        vm_dev_download(D.ctx, tr.fps.data(), D.d_fps, D.n_cycles * 4) ||
        (D.n_pos && vm_dev_download(D.ctx, tr.pos.data(), D.d_pos, D.n_pos * LM_VM_POSEIDON_CALL_WORDS * 4)) ||
        (D.n_ext && vm_dev_download(D.ctx, tr.ext.data(), D.d_ext, D.n_ext * LM_VM_EXTENSION_ROW_WORDS * 4)))
        return false;
    ex->defined.n = 0;
    ex->defined.extend(L);
    for (u64 i = 0; i < L; i++) {
        const bool d = ex->memory.p[i] != UNDEF;
        ex->defined[i] = d;
        if (!d) ex->memory.p[i] = 0;
    }
    D.host_valid = true;
    return true;
}

bool decode_instruction(const u32* row, Instr& in, std::string& why) {
    u32 c[12];
    for (int k = 0; k < 12; k++) c[k] = kb::from_monty(row[k]);
    const u32 fa = c[3], fb = c[4], fc = c[5], fcfp = c[6], fabfp = c[7], mul = c[8], jump = c[9], aux = c[10], pd = c[11];
    if (fa > 1 || fb > 1 || fc > 1 || fcfp > 1 || fabfp > 1 || mul > 1 || jump > 1 || aux > 2 || (fc && fcfp) || (fabfp && (fa || fb))) {
        why = "flag columns out of range";
        return false;
    }
    memset(&in, 0, sizeof in);
    in.a = c[0], in.b = c[1], in.c = c[2];
    in.am = row[0], in.bm = row[1], in.cm = row[2];
    in.ma = fa ? LM_VM_ARG_CONST : (fabfp ? LM_VM_ARG_FP : LM_VM_ARG_MEM);
    in.mb = fb ? LM_VM_ARG_CONST : (fabfp ? LM_VM_ARG_FP : LM_VM_ARG_MEM);
    in.mc = fc ? LM_VM_ARG_CONST : (fcfp ? LM_VM_ARG_FP : LM_VM_ARG_MEM);
    const int n_kinds = (pd != 0) + (jump != 0) + (mul != 0) + (aux != 0);
    if (n_kinds != 1) {
        why = "not exactly one of precompile_data / jump / mul / aux";
        return false;
    }
    if (pd) {
        if (pd & 1) {  // POSEIDON_PRECOMPILE_DATA + 2 permute + 4 half_output + 8 hardcoded_left + 16 offset
            in.kind = K_POSEIDON;
            in.x0 = ((pd >> 1) & 1) | (((pd >> 2) & 1) << 1) | (((pd >> 3) & 1) << 2);
            in.x1 = pd >> 4;
            if ((in.x0 & 1) && (in.x0 & 6)) {
                why = "poseidon16: permute excludes half_output / hardcoded_left";
                return false;
            }
            if (!(in.x0 & 4) && in.x1) {
                why = "poseidon16: offset without the hardcoded_left flag";
                return false;
            }
        } else {  // mode flags (4 is_be, 8 add, 16 mul, 32 poly_eq) + 64 size
            in.kind = K_EXTOP;
            in.x0 = pd & 63;
            in.x1 = pd >> 6;
            const u32 op = in.x0 & 56;
            if ((in.x0 & 3) || (op != 8 && op != 16 && op != 32) || in.x1 == 0) {
                why = "extension_op: bad mode / size";
                return false;
            }
        }
    } else if (jump)
        in.kind = K_JUMP;
    else if (mul)
        in.kind = K_MUL;
    else if (aux == 1)
        in.kind = K_ADD;
    else {
        in.kind = K_DEREF;
        if (in.ma != LM_VM_ARG_MEM || in.mb != LM_VM_ARG_CONST) {
            why = "deref: operand a must be m[fp + shift_0] and operand b the constant shift_1";
            return false;
        }
    }
    if ((in.kind == K_ADD || in.kind == K_MUL || in.kind == K_JUMP || in.kind == K_DEREF) && fabfp) {
        why = "flag_ab_fp outside a precompile";
        return false;
    }
    return true;
}

}  // namespace
}  // namespace lmh

namespace lmh {
void vm_set_late(const VmLate* late) { g_vm_late = late; }
void vm_execution_regions(const lmh_execution* e, VmRegion out[5]) {
    out[0] = {(void*)e->memory.data(), (size_t)MAX_MEMORY * 4};
    out[1] = {(void*)e->tr.pcs.data(), e->tr.pcs.cap * sizeof(u32)};
    out[2] = {(void*)e->tr.fps.data(), e->tr.fps.cap * sizeof(u32)};
    out[3] = {(void*)e->tr.pos.data(), e->tr.pos.cap * sizeof(u32)};
    out[4] = {(void*)e->tr.ext.data(), e->tr.ext.cap * sizeof(u32)};
}
void vm_set_release_hook(void (*hook)(void*)) { g_release_hook.store(hook, std::memory_order_release); }
bool vm_execution_device(const lmh_execution* e, VmDeviceView* out) {
    if (!e->dev || !e->dev->finalized || lm_ctx_by_uid(e->dev->ctx_uid) != e->dev->ctx) return false;
    const DevRun& D = *e->dev;
    out->ctx = D.ctx, out->image = D.d_image, out->memory_len = D.image_len, out->pcs = D.d_pcs, out->fps = D.d_fps, out->n_cycles = D.n_cycles;
    out->poseidon_calls = D.d_pos, out->n_poseidon_calls = D.n_pos, out->extension_rows = D.d_ext, out->n_extension_rows = D.n_ext;
    out->public_memory_size = e->public_memory_size;
    return true;
}
}  // namespace lmh

extern "C" {

lmh_bytecode* lmh_bytecode_new(const uint32_t* ml, uint32_t log_size, uint64_t n_instructions, uint32_t ending_pc, uint32_t starting_frame_memory,
                               const lm_vm_hint* hints, uint64_t n_hints, uint32_t n_hint_names) {
    if (!ml || log_size > 24 || n_instructions > (1ull << log_size) || ending_pc >= n_instructions || (n_hints && !hints)) {
        lm_set_error("lmh_bytecode_new: bad arguments");
        return nullptr;
    }
    lmh_bytecode* bc = nullptr;
    try {
        bc = new lmh_bytecode();
        bc->log_size = log_size, bc->ending_pc = ending_pc, bc->starting_frame_memory = starting_frame_memory;
        bc->n_instructions = n_instructions, bc->n_names = n_hint_names;
        bc->multilinear.assign(ml, ml + (16ull << log_size));
        bc->code.resize(n_instructions);
        for (u64 pc = 0; pc < n_instructions; pc++) {
            std::string why;
            if (!decode_instruction(ml + 16 * pc, bc->code[pc], why)) {
                lm_set_error("lmh_bytecode_new: instruction %llu: %s", (unsigned long long)pc, why.c_str());
                delete bc;
                return nullptr;
            }
        }
        bc->hint_begin.assign(n_instructions + 1, 0);
        bc->hints.resize(n_hints);
        u64 prev_pc = 0;
        for (u64 h = 0; h < n_hints; h++) {
            const lm_vm_hint& s = hints[h];
            if (s.pc >= n_instructions || s.pc < prev_pc || s.kind == 0 || s.kind > LM_VM_HINT_DEBUG_ASSERT ||
                ((s.kind == LM_VM_HINT_WITNESS_INLINE || s.kind == LM_VM_HINT_WITNESS_INDIRECT) && s.args[0] >= n_hint_names)) {
                lm_set_error("lmh_bytecode_new: hint %llu is malformed (pcs must ascend, names < n_hint_names)", (unsigned long long)h);
                delete bc;
                return nullptr;
            }
            prev_pc = s.pc;
            HintRec& r = bc->hints[h];
            r.kind = s.kind;
            memcpy(r.args, s.args, sizeof r.args);
            memcpy(r.mode, s.mode, sizeof r.mode);
            bc->hint_begin[s.pc + 1]++;
        }
        for (u64 pc = 0; pc < n_instructions; pc++) bc->hint_begin[pc + 1] += bc->hint_begin[pc];
    } catch (...) {
        delete bc;
        lm_set_error("lmh_bytecode_new: out of memory");
        return nullptr;
    }
    return bc;
}
void lmh_bytecode_free(lmh_bytecode* bc) { delete bc; }
uint32_t lmh_bytecode_log_size(const lmh_bytecode* bc) { return bc->log_size; }
uint32_t lmh_bytecode_ending_pc(const lmh_bytecode* bc) { return bc->ending_pc; }
const uint32_t* lmh_bytecode_multilinear(const lmh_bytecode* bc) { return bc->multilinear.data(); }
void lmh_bytecode_hash(const lmh_bytecode* bc, uint32_t out[8]) {
    std::lock_guard<std::mutex> lk(bc->hash_mu);
    if (!bc->hash_done) {  // poseidon_compress_slice(.., use_iv = true): hash = compress(hash || chunk) from the zero IV
        alignas(64) u32 st[16];
        memset(st, 0, sizeof st);
        const u32* d = bc->multilinear.data();
        const u64 n = bc->multilinear.size();
        for (u64 off = 0; off < n; off += 8) {
            memcpy(st + 8, d + off, 32);
            host_compress(st);
        }
        memcpy(bc->hash, st, 32);
        bc->hash_done = true;
    }
    memcpy(out, bc->hash, 32);
}

// ctx != nullptr: parallel batches run on that context's device (lm_vm_device.hip) when they qualify, else on the host pool
static int execute_impl(lm_ctx* ctx, const lmh_bytecode* bc, const uint32_t* public_input, uint32_t n_public_input, const lm_vm_witness* witness,
                        uint32_t n_threads, lmh_execution** out, bool allow_deferred = true) {
    if (!bc || !out || (n_public_input && !public_input) || !witness || witness->n_names != bc->n_names ||
        (bc->n_names && (!witness->name_entry_begin || !witness->entry_offset))) {
        lm_set_error("lmh_execute_bytecode: bad arguments (the witness must carry one hint stream per name of the bytecode)");
        return LM_E_INVALID;
    }
    *out = nullptr;
    lmh_execution* ex = nullptr;
    try {
        ex = new lmh_execution();
        PoolSession session(n_threads);  // the pool stays hot from here to the end of the run
        Witness w{witness->preamble_memory_len, witness->name_entry_begin, witness->entry_offset, witness->data};
        const VmLate* late = g_vm_late;  // (vm_set_late: this run's inputs are not all final yet)
        g_vm_late = nullptr;
        struct LateGuard {  // whatever way the run ends, its caller reads the inputs afterwards: they are final by then
            const VmLate* l;
            ~LateGuard() {
                if (l && l->wait) l->wait();
            }
        } late_guard{late};
        // execute_bytecode_helper (runner.rs:238-343)
        u64 pub = 1;
        while (pub < n_public_input) pub <<= 1;  // padd_with_zero_to_next_power_of_two (0usize.next_power_of_two() == 1: one zero word)
        MemBuf& memory = ex->memory;
        memory.len = pub;
        memset(memory.p, 0, 4 * pub);
        if (n_public_input) memcpy(memory.p, public_input, 4ull * n_public_input);
        u64 fp = pub + w.preamble_memory_len;
        fp = (fp + 4) / 5 * 5;  // next_multiple_of(DIMENSION)
        const u64 initial_ap = fp + bc->starting_frame_memory;
        Cursors cur;
        cur.index.assign(bc->n_names, 0);
        MainMem mm{memory};
        Machine<MainMem> m(*bc, w, mm, ex->tr, cur);
        m.fp = fp, m.ap = initial_ap, m.pc = 0;
        const double t_start = vm_now_ms();
        double t_batches = 0;
        DevRun* D = nullptr;
        if (ctx && vm_device_enabled()) {
            D = new DevRun();
            D->ctx = ctx;
            D->ctx_uid = lm_ctx_uid(ctx);
            ex->dev = D;
            memory.on_touch = [D, &memory] { return dev_close_windows(*D, memory); };
            const double tw0 = vm_now_ms();
            if (!dev_witness(*D, witness)) {  // the hint streams travel while the sequential head of the program runs
                delete ex;
                return LM_E_DEVICE;
            }
            if (vm_times()) fprintf(stderr, "[vm] hint streams: upload enqueued in %.3f ms\n", vm_now_ms() - tw0);
        }
        // deferred Poseidon calls (MemBuf) pay off when a batch runs on the device; LM_VM_LAZY=1 / 0 forces them on (host runs too) / off
        if (const char* e = getenv("LM_VM_LAZY"))
            memory.lazy_on = e[0] == '1';
        else
            memory.lazy_on = D != nullptr;
        if (!allow_deferred || (memory.lazy_on && !memory.lazy_alloc())) memory.lazy_on = false;
        memory.lazy_rows = &ex->tr.ext;
        if (late) {
            if (!memory.lazy_on || !late->wait) {  // nothing is deferred in this run: the inputs are awaited here
                if (late->wait) late->wait();
                if (n_public_input) memcpy(memory.p, public_input, 4ull * n_public_input);
            } else {
                memory.late_wait = late->wait;
                w.late = late;
                if (late->public_input && n_public_input) {
                    for (u32 i = 0; i < n_public_input; i++) memory.p[i] = UNDEF;
                    bool ok = true;
                    for (u32 at = 0; at < n_public_input && ok; at += 64) ok = memory.lazy_late(at, public_input + at, std::min<u32>(64, n_public_input - at));
                    if (!ok) {
                        late->wait();
                        memory.lazy_drain();
                        memcpy(memory.p, public_input, 4ull * n_public_input);
                    }
                }
            }
        }
        bool device_failed = false;
        // INVARIANT of re-arming (a batch per parallel loop where the reference arms one per run): the ExecutionResult is the SEQUENTIAL
        // run's — pcs, fps, every defined cell, and the memory LENGTH, which feeds log_memory and the public memory and so the proof.  A
        // batch resizes the memory to the end of its last call frame (runner.rs:404-407); a sequential run of the same loop ends at the
        // last cell it SET.  For the one batch the reference runs, the resize is the reference's own behaviour; for every later one the
        // length is put back (trim_to_defined, only when the batch grew the memory to exactly its frame end — a batch that did not grow it
        // changed no length).  Nested batches, a batch inside a segment, or an error with more than one batch behind it repeat the run
        // with the reference's arming, literally (below).  tests/test_vm.py::test_consecutive_parallel_batches_leave_the_sequential_
        // memory_length and tools/vm_fuzz.py compare length, cells and logs with the sequential oracle VM.
        static const bool rearm_env = !(getenv("LM_VM_REARM") && getenv("LM_VM_REARM")[0] == '0');
        const bool rearm = allow_deferred && rearm_env;  // (the repeated run that reports an error is the reference's, literally)
        u64 skip_arm_pc = ~0ull;
        u32 n_batches = 0;
        for (;;) {
            Machine<MainMem>::Batch batch;
            const int rc = m.run(false, 0, batch, skip_arm_pc);
            if (rc == 0) break;
            if (rc < 0) break;
            const double tb = vm_now_ms();
            if (vm_times() && n_batches == 0) fprintf(stderr, "[vm] sequential head: %.3f ms until the first batch\n", tb - t_start);
            int how = DEV_FALLBACK;
            std::string why = ctx ? "LM_VM_HOST is set" : "no device context (lmh_execute_bytecode)";
            const bool extra = n_batches > 0;  // a batch the reference runs sequentially (Machine::run, skip_arm_pc)
            if (D) how = device_batch(*bc, witness, memory, ex->tr, cur, m.pc, m.fp, m.ap, batch, n_threads, *D, why, extra, late);
            if (how == DEV_DONE)
                ex->n_device_batches++;
            else if (how == DEV_FALLBACK) {
                if (!ex->n_host_batches) ex->host_batch_reason = why;
                ex->n_host_batches++;
            }
            if (how == DEV_ERROR) {
                device_failed = true;
                break;
            }
            if (how == DEV_FALLBACK && D && D->windows_open) {
                // a host batch behind a device batch: its segments read the arena directly (SegMem, no MemBuf::guard), so the frames of the
                // earlier device batches — current in the device image only — come back first (round-4 advisor finding: device_batch's early
                // returns did not do this, only its fallback() lambda)
                if (!dev_close_windows(*D, memory)) {
                    device_failed = true;
                    break;
                }
                memory.dev_lo = memory.dev_hi = 0;
            }
            const bool ok = how == DEV_DONE || handle_parallel_batch(*bc, w, memory, ex->tr, cur, m.pc, m.fp, m.ap, batch, n_threads, m.err, extra);
            t_batches += vm_now_ms() - tb;
            n_batches++;
            if (rearm) skip_arm_pc = batch.batch_pc;
            if (!ok || memory.lazy_failed) break;
        }
        if (!m.err.set && !device_failed) memory.lazy_drain();
        if (!device_failed && allow_deferred && (memory.lazy_failed || (m.err.set && (memory.lazy_checks || n_batches > 1)))) {
            // a deferred check failed, or an error was met with checks deferred: the first error of the program is what a run
            // without deferred work reports
            if (vm_times()) fprintf(stderr, "[vm] deferred checks: the run is repeated with every instruction executed at once\n");
            delete ex;
            if (late && late->wait) late->wait();  // (the repeated run reads the inputs as they are)
            return execute_impl(ctx, bc, public_input, n_public_input, witness, n_threads, out, false);
        }
        if (memory.touch_failed) device_failed = true;
        if (device_failed) {
            delete ex;
            return LM_E_DEVICE;  // (lm_last_error holds the device call that failed)
        }
        const double t_loop = vm_now_ms();
        const bool on_device = D && !D->batches.empty();
        if (!m.err.set && !on_device) resolve_deref_hints(mm, ex->tr.pending, n_threads, m.err);
        const double t_resolve = vm_now_ms();
        if (!m.err.set)
            for (u32 k = 0; k < bc->n_names; k++)
                if (cur.index[k] != witness->name_entry_begin[k + 1] - witness->name_entry_begin[k]) {
                    m.err.raise("Panic: not all entries of named hint %u were consumed (%llu of %llu used)", k, (unsigned long long)cur.index[k],
                                (unsigned long long)(witness->name_entry_begin[k + 1] - witness->name_entry_begin[k]));
                    break;
                }
        if (m.err.set) {
            lm_set_error("lmh_execute_bytecode: pc %llu: %s", (unsigned long long)m.pc, m.err.msg.c_str());
            delete ex;
            return LM_E_INVALID;
        }
        ex->tr.pcs.push_back((u32)m.pc);
        ex->tr.fps.push_back((u32)m.fp);
        ex->public_memory_size = pub;
        ex->runtime_memory_size = m.ap - initial_ap;
        if (on_device) {
            // the log and the image are assembled in HBM and resolve_deref_hints runs there; anything the device cannot decide (a
            // conflict, an undefined source: a RunnerError or a panic of the reference) repeats the run on the host, which reports it
            bool anomaly = false;
            const bool ok = device_finalize(ex, *D, anomaly);
            if (vm_times())
                vm_prof_report(), fprintf(stderr, "[vm] sequential parts %.2f ms, batches %.2f ms, device assembly + resolve_deref_hints %.2f ms%s\n", t_loop - t_start - t_batches,
                        t_batches, vm_now_ms() - t_resolve, anomaly ? " (anomaly: the run is repeated on the host)" : "");
            if (!ok) {
                delete ex;
                if (!anomaly) return LM_E_DEVICE;
                if (late && late->wait) late->wait();
                const int rc = execute_impl(nullptr, bc, public_input, n_public_input, witness, n_threads, out);
                if (rc == LM_OK && *out) (*out)->run_repeated = 1, (*out)->host_batch_reason = "resolve_deref_hints met a conflict / an undefined source on the device: the run was repeated on the host";
                return rc;
            }
            *out = ex;
            return LM_OK;
        }
        if (D) {  // no batch ran on the device: a plain host run
            ex->dev = nullptr;
            dev_run_free(D);
            memory.on_touch = nullptr;
        }
        const u64 n = memory.size();
        ex->defined.extend(n);
        uint8_t* def = ex->defined.data();
        u32* mem = memory.data();
        const u64 chunk = 1u << 16;
        vm_parallel_for((n + chunk - 1) / chunk, n_threads, [&](u64 c) {
            const u64 e = std::min(n, (c + 1) * chunk);
            for (u64 i = c * chunk; i < e; i++) {
                const bool d = mem[i] != UNDEF;
                def[i] = d;
                if (!d) mem[i] = 0;
            }
        });
        // get_execution_trace appends [0 x 16 | poseidon16_compress(0 x 16)] behind the memory (trace_gen.rs:106-110): written behind
        // memory_len in the arena, so that the trace builder uploads image and tail with one copy
        if (n + 24 <= MAX_MEMORY) {
            alignas(64) u32 st[16];
            memset(st, 0, sizeof st);
            memset(mem + n, 0, 16 * 4);
            host_compress(st);
            memcpy(mem + n + 16, st, 32);
        }
        if (vm_times())
            vm_prof_report(), fprintf(stderr, "[vm] sequential parts %.2f ms, batches %.2f ms, resolve_deref_hints %.2f ms, defined mask %.2f ms\n",
                    t_loop - t_start - t_batches, t_batches, t_resolve - t_loop, vm_now_ms() - t_resolve);
    } catch (const std::bad_alloc&) {
        delete ex;
        lm_set_error("lmh_execute_bytecode: out of memory");
        return LM_E_NOMEM;
    }
    *out = ex;
    return LM_OK;
}
int lmh_execute_bytecode(const lmh_bytecode* bc, const uint32_t* public_input, uint32_t n_public_input, const lm_vm_witness* witness,
                         uint32_t n_threads, lmh_execution** out) {
    return execute_impl(nullptr, bc, public_input, n_public_input, witness, n_threads, out);
}
int lmh_execute_bytecode_device(lm_ctx* ctx, const lmh_bytecode* bc, const uint32_t* public_input, uint32_t n_public_input, const lm_vm_witness* witness,
                                uint32_t n_threads, lmh_execution** out) {
    if (!ctx) {
        lm_set_error("lmh_execute_bytecode_device: no context");
        return LM_E_INVALID;
    }
    return execute_impl(ctx, bc, public_input, n_public_input, witness, n_threads, out);
}
int lmh_execution_on_device(const lmh_execution* e) { return e && e->dev && e->dev->finalized; }
void lmh_execution_info(const lmh_execution* e, lm_vm_run_info* out) {
    memset(out, 0, sizeof *out);
    if (!e) return;
    out->on_device = e->dev && e->dev->finalized;
    out->n_device_batches = e->n_device_batches, out->n_host_batches = e->n_host_batches, out->run_repeated = e->run_repeated;
    snprintf(out->host_batch_reason, sizeof out->host_batch_reason, "%s", e->host_batch_reason.c_str());
}
int lmh_bytecode_set_hint_names(lmh_bytecode* bc, const char* const* names, uint32_t n_names) {
    if (!bc || n_names != bc->n_names || (n_names && !names)) {
        lm_set_error("lmh_bytecode_set_hint_names: one name per hint stream of the bytecode");
        return LM_E_INVALID;
    }
    bc->names.assign(names, names + n_names);
    return LM_OK;
}
int lmh_bytecode_hint_name_id(const lmh_bytecode* bc, const char* name) {
    for (size_t k = 0; k < bc->names.size(); k++)
        if (bc->names[k] == name) return (int)k;
    return -1;
}
uint32_t lmh_bytecode_n_hint_names(const lmh_bytecode* bc) { return bc->n_names; }
void lmh_execution_free(lmh_execution* e) { delete e; }
void lmh_execution_view(const lmh_execution* e, lm_vm_execution_view* v) {
    memset(v, 0, sizeof *v);
    if (e->dev && e->dev->finalized) {  // the log lives in HBM: the host view is one download, made on the first request
        lmh_execution* me = const_cast<lmh_execution*>(e);
        if (!device_materialize(me)) return;  // (v stays empty: n_cycles = 0)
        const u64 n = me->memory.size();
        if (n + 24 <= MAX_MEMORY) {
            alignas(64) u32 st[16];
            memset(st, 0, sizeof st);
            memset(me->memory.p + n, 0, 16 * 4);
            host_compress(st);
            memcpy(me->memory.p + n + 16, st, 32);
        }
    }
    v->n_cycles = e->tr.pcs.size();
    v->pcs = e->tr.pcs.data();
    v->fps = e->tr.fps.data();
    v->memory_len = e->memory.size();
    v->memory = e->memory.data();
    v->memory_defined = e->defined.data();
    v->public_memory_size = e->public_memory_size;
    v->runtime_memory_size = e->runtime_memory_size;
    v->n_poseidon_calls = e->tr.pos.size() / LM_VM_POSEIDON_CALL_WORDS;
    v->poseidon_calls = e->tr.pos.data();
    v->n_extension_rows = e->tr.ext.size() / LM_VM_EXTENSION_ROW_WORDS;
    v->extension_rows = e->tr.ext.data();
    v->n_add = e->tr.n_add, v->n_mul = e->tr.n_mul, v->n_deref = e->tr.n_deref, v->n_jump = e->tr.n_jump;
    if (e->dev && e->dev->finalized) v->n_add += e->dev->n_add, v->n_mul += e->dev->n_mul, v->n_deref += e->dev->n_deref, v->n_jump += e->dev->n_jump;
}

void lmh_poseidon16_compress_many(uint32_t* states, uint64_t n, uint32_t n_threads) {
    const u64 chunk = 256;
    vm_parallel_for((n + chunk - 1) / chunk, n_threads, [&](u64 c) {
        const u64 e = std::min(n, (c + 1) * chunk);
        for (u64 i = c * chunk; i < e; i++) {
            alignas(64) u32 st[16];
            memcpy(st, states + 16 * i, 64);
            host_compress(st);
            memcpy(states + 16 * i, st, 64);
        }
    });
}

}  // extern "C"
#endif
