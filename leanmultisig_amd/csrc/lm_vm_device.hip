// leanVM on the device: the segments of a parallel loop batch, one wavefront each (round 4).
//
// Reference: crates/lean_vm/src/execution/runner.rs:369-482 (handle_parallel_batch: independent per-iteration segments),
// execution/memory.rs:118-189 (SegmentMemory), isa/instruction.rs:146-246 (execute_instruction), isa/hint.rs:137-386 (hints),
// tables/poseidon_16/mod.rs:209-289 and tables/extension_op/exec.rs (precompile execution), runner.rs:206-236 (resolve_deref_hints).
// The host runner (csrc/host/lm_vm.cpp) executes the sequential parts of a program and hands a batch to k_vm_segments; what the
// segments log stays in HBM for the trace kernels (lm_logup.hip).  Design for gfx950:
//   * one segment = one wave64.  The interpreter state (pc, fp, ap, operands) is wave-uniform; every lane runs the same scalar
//     stream, so there is no divergence between lanes and no cross-wave synchronisation at all;
//   * the segment's frame (its only writable memory) is an LDS array — a leanVM instruction is 2-3 dependent memory accesses, LDS
//     answers in ~100 cycles where HBM / L2 would take ~1 us each; the shared prefix below the batch is read-only and comes from
//     HBM through the caches; the frame is written back with coalesced stores when the segment ends;
//   * Poseidon16 = the 16-lane cooperative permutation of poseidon16_coop.h (a ~1.1 k instruction chain instead of ~7 k): lane l
//     gathers state word l, the 4 rows of the wave compute the same permutation (no divergence), results are stored per lane;
//   * writes outside the frame are appended to the segment's deferred list in program order (ballot + prefix count), exactly the
//     list SegmentMemory::into_deferred_writes returns;
//   * ANY irregularity (a RunnerError, a log slot that is too small, a nested batch) ends the segment with an error code: the host
//     then discards the device batch and runs it on its thread pool, which reports the reference's error.  The device path never
//     has to format an error or decide an order between failures.
#include "lm_common.h"
#include "poseidon16_coop.h"
#include "host/lm_vm_device.h"
#include "../../include/leanmultisig_host.h"

using namespace kb;
using namespace lmh;

namespace {
constexpr u32 UNDEF = VM_UNDEF;
constexpr u32 R2 = 0x17f7efe4u;  // 2^64 mod p: mul(x, R2) = x * 2^32 mod p
static_assert((u32)(((u64)ONE * ONE) % P) == R2, "R2 = (2^32 mod p)^2 mod p");

constexpr u32 VM_WIN = 48;   // instruction window (records)
constexpr u32 VM_HWIN = 32;  // hint window (records)
__host__ __device__ constexpr u32 vm_wave_lds_words(u32 stride_pad) {  // frame | cursors | instruction window | hint range | hint window
    return stride_pad + 2 * VM_DEV_MAX_NAMES + VM_WIN * 9 + (VM_WIN + 2) + VM_HWIN * 6;
}

__device__ __forceinline__ u32 to_monty_d(u32 x) { return mul(x % P, R2); }
__device__ __forceinline__ u32 rfl(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ u64 rfl64(u64 x) { return ((u64)rfl((u32)(x >> 32)) << 32) | rfl((u32)x); }
__device__ __forceinline__ u32 lanes_below(u64 mask) {  // set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0));
}

struct Machine {
    const VmSegArgs& A;
    const u32* coop_tab;  // LDS copy of the 16-lane Poseidon table (poseidon16_coop.h): loaded into registers per call, not held across the loop
    const u32* pcache;    // LDS copy of image[0 .. k_prefix_cache): the lowest addresses (constants, tables) are what every segment reads
    u32* frame;    // LDS: the segment's frame, k_stride words
    u64* cursors;  // LDS: named hint cursors
    // instruction window (LDS): records [win_base, win_base + VM_WIN) and their hint ranges, refilled with one coalesced load when
    // the pc leaves it — a leanVM program runs ~8 instructions between jumps, a global load per instruction would cost ~1 us each
    u32* wcode;    // VM_WIN x 9 words
    u32* whb;      // VM_WIN + 1 words: hint_begin[win_base ..]
    u32* whint;    // VM_HWIN x 6 words: hints [hwin_base, hwin_base + VM_HWIN)
    u32 win_base = 0xFFFFFFFFu, win_n = 0, hwin_base = 0xFFFFFFFFu, hwin_n = 0;
    // Addresses are 32-bit here: every address the interpreter forms is the sum of two values below 2^31 (an fp or a pointer read from
    // memory — field elements — plus an operand offset / a small index); hint operands are checked on entry (run_hint).
    u32 seg_start;
    u32 lane;
    u32 pc;
    u32 fp;
    u64 ap;
    u32 n_cyc = 0, n_pos = 0, n_ext = 0, n_pend = 0, n_def = 0, n_add = 0, n_mul = 0, n_deref = 0, n_jump = 0;
    mutable u32 err = 0, err_aux = 0;  // (mutable: reading a poisoned cell fails the segment from the const read paths)
    u32 *pcs, *fps, *pos, *ext, *pend, *def;  // this segment's slots
    // hot kernel arguments as plain members behind an opaque copy (hot_args): left in the kernarg segment the compiler re-loads them
    // with scalar loads at every use under register pressure (373 s_load in the first version, a wait on each)
    u32 k_split_at, k_stride;
    u32 k_prefix_cache, k_cap_cyc, k_cap_pos, k_cap_ext, k_cap_pend, k_cap_def, k_batch_pc, k_ending_pc, k_n_instructions, k_n_hints, k_dbg;
    u32* k_image;
    const VmInstr* k_code;
    const u32* k_hint_begin;
    const VmHintRec* k_hints;
    __device__ __forceinline__ void hot_args() {
        k_split_at = (u32)A.split_at, k_stride = (u32)A.stride, k_prefix_cache = A.prefix_cache, k_cap_cyc = A.cap_cyc, k_cap_pos = A.cap_pos, k_cap_ext = A.cap_ext;
        k_cap_pend = A.cap_pend, k_cap_def = A.cap_def, k_batch_pc = A.batch_pc, k_ending_pc = A.ending_pc, k_n_instructions = A.n_instructions;
        k_n_hints = A.n_hints, k_dbg = A.dbg, k_image = A.image, k_code = A.code, k_hint_begin = A.hint_begin, k_hints = A.hints;
        asm volatile("" : "+s"(k_split_at), "+s"(k_stride), "+s"(k_prefix_cache), "+s"(k_cap_cyc), "+s"(k_cap_pos), "+s"(k_cap_ext), "+s"(k_cap_pend),
                     "+s"(k_cap_def));
        asm volatile("" : "+s"(k_batch_pc), "+s"(k_ending_pc), "+s"(k_n_instructions), "+s"(k_n_hints), "+s"(k_dbg), "+s"(k_image), "+s"(k_code),
                     "+s"(k_hint_begin), "+s"(k_hints));
    }

    __device__ __forceinline__ Machine(const VmSegArgs& a) : A(a) {}
    __device__ __forceinline__ bool dbg_bit(u32 b) const {  // the timing knobs exist in -DLM_VM_DEBUG builds only (round-4 advisor finding)
#ifdef LM_VM_DEBUG
        return k_dbg & b;
#else
        return false;
#endif
    }

    __device__ __forceinline__ void fail(u32 code, u64 aux) const {
        if (!err) err = code, err_aux = (u32)aux;
    }
    // ---- SegmentMemory (memory.rs:118-189) ------------------------------------------------------------------------------------------
    __device__ __forceinline__ u32 peek(u32 a) const {  // any lane, any address
        if (a < k_prefix_cache) return pcache[a];
        if (a < k_split_at) return k_image[a];
        const u32 o = a - seg_start;  // (wraps to a huge value below the frame)
        if (o < k_stride) return frame[o];
        return UNDEF;
    }
    __device__ __forceinline__ u32 peek_u(u32 a) const {  // wave-uniform address
        u32 v = rfl(peek(a));
        if (v == VM_PENDING) {  // a digest the host has not computed yet (lm_vm_device.h): not None — the segment gives up
            fail(VM_E_UNSUPPORTED, 4);
            v = UNDEF;
        }
        return v;
    }
    // deferred write list: the lanes of `active` append (addr, value) in lane order
    __device__ __forceinline__ void defer_lanes(bool active, u32 a, u32 v) {
        const u64 m = __ballot(active);
        if (!m) return;
        const u32 cnt = (u32)__popcll(m);
        if (n_def + cnt > k_cap_def) {
            fail(VM_E_LOG_CAPACITY, 5);
            return;
        }
        if (active) {
            const u32 at = n_def + lanes_below(m);
            def[2 * at] = a;
            def[2 * at + 1] = v;
        }
        n_def += cnt;
    }
    // per-lane write (distinct addresses per lane): own frame = write-once cell, everything else is deferred
    __device__ __forceinline__ void set_lanes(bool active, u32 a, u32 v) {
        const u32 o = a - seg_start;
        const bool mine = o < k_stride;
        bool clash = false;
        if (active && mine) {
            const u32 c = frame[o];
            if (c == UNDEF)
                frame[o] = v;
            else
                clash = c != v;
        }
        if (__ballot(clash)) {
            fail(VM_E_MEMORY_ALREADY_SET, a);
            return;
        }
        defer_lanes(active && !mine, a, v);
    }
    // wave-uniform write: every lane holds the same (a, v) — no ballots, every lane stores the same word
    __device__ __forceinline__ void set_u(u32 a, u32 v) {
        const u32 o = a - seg_start;
        if (o < k_stride) {
            const u32 c = rfl(frame[o]);
            if (c == UNDEF)
                frame[o] = v;
            else if (c != v)
                fail(VM_E_MEMORY_ALREADY_SET, a);
        } else if (n_def >= k_cap_def)
            fail(VM_E_LOG_CAPACITY, 5);
        else {
            if (lane == 0) def[2 * n_def] = a, def[2 * n_def + 1] = v;
            n_def++;
        }
    }

    __device__ __forceinline__ u32 need_mem(u32 a) {
        const u32 v = peek_u(a);
        if (v == UNDEF) fail(VM_E_UNDEFINED_MEMORY, a);
        return v;
    }
    // MemOrConstant / MemOrFpOrConstant::read_value: UNDEF when the value is unknown
    __device__ __forceinline__ u32 read(u32 mode, u32 canon, u32 monty) const {
        if (mode == LM_VM_ARG_CONST) return monty;
        if (mode == LM_VM_ARG_MEM) return peek_u(fp + canon);
        return to_monty_d(fp + canon);
    }
    __device__ __forceinline__ u32 need(u32 mode, u32 canon, u32 monty) {
        const u32 v = read(mode, canon, monty);
        if (v == UNDEF) fail(VM_E_UNDEFINED_MEMORY, fp + canon);
        return v;
    }
    static __device__ __forceinline__ u32 usize(u32 monty) { return from_monty(monty); }

    // ---- hints (isa/hint.rs:270-386, CustomHint::execute :137-203) ----------------------------------------------------------------------
    __device__ __forceinline__ u32 hint_arg(const VmHintRec& h, int k) {
        const u32 mode = h.mode[k];
        if (mode == LM_VM_ARG_CONST) return to_monty_d(h.args[k]);
        if (mode == LM_VM_ARG_MEM) return need_mem(fp + h.args[k]);
        return to_monty_d(fp + h.args[k]);
    }
    __device__ __forceinline__ void run_hint(const VmHintRec& h) {
        if ((h.args[0] | h.args[1] | h.args[2] | h.args[3]) >> 30) {  // (32-bit address arithmetic: such an operand goes to the host runner)
            fail(VM_E_UNSUPPORTED, 2);
            return;
        }
        switch (h.kind) {
            case LM_VM_HINT_REQUEST_MEMORY: {
                const u32 size = hint_arg(h, 1);
                if (err) return;
                set_u(fp + h.args[0], to_monty_d((u32)(ap % P)));
                ap += usize(size);
                break;
            }
            case LM_VM_HINT_INVERSE: {
                const u32 v = hint_arg(h, 0);
                if (err) return;
                set_u(fp + h.args[1], v ? inv(v) : 0u);
                break;
            }
            case LM_VM_HINT_DEREF: {
                if (n_pend >= k_cap_pend) {
                    fail(VM_E_LOG_CAPACITY, 4);
                    return;
                }
                if (lane == 0) pend[2 * n_pend] = (u32)(fp + h.args[1]), pend[2 * n_pend + 1] = (u32)(fp + h.args[0]);  // (target, src)
                n_pend++;
                break;
            }
            case LM_VM_HINT_DECOMPOSE_BITS_XMSS: {
                const u32 dp = hint_arg(h, 0), sp = hint_arg(h, 1), nn = hint_arg(h, 2), cs = hint_arg(h, 3);
                if (err) return;
                const u64 chunk = usize(cs);
                if (chunk == 0 || 24 % chunk) {
                    fail(VM_E_HINT, 1);
                    return;
                }
                u64 out = usize(dp);
                const u64 src = usize(sp), num = usize(nn);
                const u32 per = (u32)(24 / chunk);  // <= 24 digits of one value: one per lane
                for (u64 i = 0; i < num && !err; i++) {
                    const u32 v = need_mem(src + i);
                    if (err) return;
                    const u64 x = usize(v);
                    set_lanes(lane < per, out + lane, to_monty_d((u32)((x >> (chunk * (lane < per ? lane : 0))) & ((1ull << chunk) - 1))));
                    out += per;
                }
                break;
            }
            case LM_VM_HINT_DECOMPOSE_BITS_MERKLE_WHIR: {
                const u32 dp = hint_arg(h, 0), vv = hint_arg(h, 1), cs = hint_arg(h, 2);
                if (err) return;
                const u64 chunk = usize(cs), x = usize(vv);
                if (chunk == 0 || 24 % chunk) {
                    fail(VM_E_HINT, 2);
                    return;
                }
                const u32 per = (u32)(24 / chunk);
                set_lanes(lane < per, usize(dp) + lane, to_monty_d((u32)((x >> (chunk * (lane < per ? lane : 0))) & ((1ull << chunk) - 1))));
                break;
            }
            case LM_VM_HINT_DECOMPOSE_BITS: {  // to_big_endian_in_field(to_decompose, num_bits)
                const u32 vv = hint_arg(h, 0), mi = hint_arg(h, 1), nb = hint_arg(h, 2);
                if (err) return;
                const u64 x = usize(vv), at = usize(mi), bits = usize(nb);
                if (bits > 31) {
                    fail(VM_E_HINT, 3);
                    return;
                }
                set_lanes(lane < bits, at + lane, ((x >> (bits - 1 - (lane < bits ? lane : 0))) & 1) ? ONE : 0u);
                break;
            }
            case LM_VM_HINT_LESS_THAN: {
                const u32 a = hint_arg(h, 0), b = hint_arg(h, 1);
                if (err) return;
                if (h.mode[2] != LM_VM_ARG_MEM) {
                    fail(VM_E_NOT_A_POINTER, 0);
                    return;
                }
                set_u(fp + h.args[2], usize(a) < usize(b) ? ONE : 0u);
                break;
            }
            case LM_VM_HINT_LOG2_CEIL: {
                const u32 n = hint_arg(h, 0);
                if (err) return;
                if (h.mode[1] != LM_VM_ARG_MEM) {
                    fail(VM_E_NOT_A_POINTER, 0);
                    return;
                }
                const u64 x = usize(n);
                u32 l = 0;
                while ((1ull << l) < x) l++;
                set_u(fp + h.args[1], to_monty_d(l));
                break;
            }
            case LM_VM_HINT_WITNESS_INLINE:
            case LM_VM_HINT_WITNESS_INDIRECT: {
                const u32 name = h.args[0];
                if (name >= A.n_names) {
                    fail(VM_E_HINT, 4);
                    return;
                }
                const u64 e = A.wit_name_begin[name] + cursors[name];
                if (e >= A.wit_name_begin[name + 1]) {
                    fail(VM_E_HINT, 5);
                    return;
                }
                cursors[name] = cursors[name] + 1;  // (every lane stores the same value)
                u64 dest;
                if (h.kind == LM_VM_HINT_WITNESS_INLINE)
                    dest = fp + h.args[1];
                else {
                    const u32 p = need_mem(fp + h.args[1]);
                    if (err) return;
                    dest = usize(p);
                }
                const u64 k0 = rfl64(A.wit_entry_offset[e]), k1 = rfl64(A.wit_entry_offset[e + 1]);
                if ((k1 - k0) >> 30) {
                    fail(VM_E_UNSUPPORTED, 3);
                    return;
                }
                for (u64 k = k0; k < k1 && !err; k += 64) {
                    const bool on = k + lane < k1;
                    set_lanes(on, (u32)(dest + (k - k0)) + lane, on ? A.wit_data[k + lane] : 0u);
                }
                break;
            }
            case LM_VM_HINT_DEBUG_ASSERT: {
                const u32 l = hint_arg(h, 0), r = hint_arg(h, 1);
                if (err) return;
                const u64 lv = usize(l), rv = usize(r);
                if (h.args[3] && rv >= (1ull << 16)) {  // MIN_LOG_MEMORY_SIZE
                    fail(VM_E_DEBUG_ASSERT, 1);
                    return;
                }
                bool ok;
                switch (h.args[2]) {
                    case 0: ok = lv == rv; break;
                    case 1: ok = lv != rv; break;
                    case 2: ok = lv < rv; break;
                    default: ok = lv <= rv; break;
                }
                if (!ok) fail(VM_E_DEBUG_ASSERT, 0);
                break;
            }
            default:
                break;
        }
    }

    // ---- Poseidon16Precompile::execute (poseidon_16/mod.rs:209-289) ------------------------------------------------------------------------
    __device__ __forceinline__ void poseidon(const VmInstr& in, u32 va, u32 vb, u32 vc) {
        const bool permute = in.x0 & 1, half = in.x0 & 2, hard = in.x0 & 4;
        const u32 arg_a = usize(va), arg_b = usize(vb), res = usize(vc);
        const u32 left_first = hard ? in.x1 : arg_a;
        const u32 left_second = hard ? arg_a : arg_a + 4;
        if (n_pos >= k_cap_pos) {
            fail(VM_E_LOG_CAPACITY, 2);
            return;
        }
        const u32 l = lane & 15;  // the four 16-lane rows of the wave compute the same permutation
        const u32 src = l < 4 ? left_first + l : (l < 8 ? left_second + (l - 4) : arg_b + (l - 8));
        const u32 s = peek(src);
        if (__ballot(s == UNDEF || s == VM_PENDING)) {
            fail(__ballot(s == VM_PENDING) ? VM_E_UNSUPPORTED : VM_E_UNDEFINED_MEMORY, src);
            return;
        }
        CoopRegs R;
        {   // the LDS copy is TRANSPOSED (word i of lane l at 16 + i * 16 + l): the 16 lanes of a row read 16 consecutive banks; in the
            // global layout (l * 128 + i) they would all hit one bank, a 16-way conflict on each of the 119 loads
            const u32 l = lane & 15;
            u32* dst = reinterpret_cast<u32*>(&R.t);
#pragma unroll
            for (u32 i = 0; i < COOP_TAB_STRIDE - 9; i++) dst[i] = coop_tab[16 + i * 16 + l];
#pragma unroll
            for (int k = 0; k < 16; k++) R.mds[k] = coop_tab[k];  // uniform
        }
        {   // all 135 table words in registers BEFORE the dependent chain starts: left to the scheduler the loads sink next to their
            // uses and every multiply of the permutation waits for its own LDS round trip (14 us per call instead of 3)
            u32* rw = reinterpret_cast<u32*>(&R);
#pragma unroll
            for (u32 i = 0; i < sizeof(CoopRegs) / 4; i++) asm volatile("" : "+v"(rw[i]));
        }
#ifdef LM_VM_DEBUG
        const u32 o = (k_dbg & 1) ? s : (permute ? coop_permute(s, R) : coop_compress(s, R));
#else
        const u32 o = permute ? coop_permute(s, R) : coop_compress(s, R);
#endif
        const u32 n_out = permute ? 16u : (half ? 4u : 8u);
        set_lanes(lane < n_out, res + lane, o);
        if (err) return;
        if (lane == 0 && !dbg_bit(2)) {
            u32* rec = pos + (u64)n_pos * LM_VM_POSEIDON_CALL_WORDS;
            rec[0] = arg_a, rec[1] = arg_b, rec[2] = res, rec[3] = half ? 1u : 0u, rec[4] = hard ? 1u : 0u, rec[5] = hard ? in.x1 : 0u;
            rec[6] = left_first, rec[7] = left_second, rec[8] = permute ? 1u : 0u;
        }
        n_pos++;
    }

    // ---- extension_op/exec.rs --------------------------------------------------------------------------------------------------------------
    enum { OP_ADD = 8, OP_MUL = 16, OP_POLY_EQ = 32 };
    static __device__ __forceinline__ EF compute_elem(const EF& a, const EF& b, u32 op) {
        if (op == OP_ADD) return ef_add(a, b);
        const EF ab = ef_mul(a, b);
        if (op == OP_MUL) return ab;
        return ef_add_base(ef_sub(ef_sub(ef_dbl(ab), a), b), ONE);  // 2ab - a - b + 1
    }
    __device__ __forceinline__ EF frobenius(const EF& a) const {
        EF r = ef_zero();
#pragma unroll
        for (int i = 0; i < 5; i++) r = ef_add(r, ef_mul_base(A.frob[i], a.v[i]));
        return r;
    }
    __device__ __forceinline__ EF ef_inv_d(const EF& a) const {  // quintic_extension/extension.rs:585-607 (the value is unique)
        const EF f1 = frobenius(a);
        const EF f12 = ef_mul(f1, frobenius(f1));
        const EF f34 = frobenius(frobenius(f12));
        const EF conj = ef_mul(f12, f34);
        const EF n = ef_mul(a, conj);
        return ef_mul_base(conj, inv(n.v[0]));
    }
    __device__ __forceinline__ bool peek_ef(u64 at, EF& out) const {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            out.v[k] = peek_u(at + k);
            ok = ok && out.v[k] != UNDEF;
        }
        return ok;
    }
    __device__ __forceinline__ bool need_ef(u64 at, EF& out) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            out.v[k] = need_mem(at + k);
            if (err) return false;
        }
        return true;
    }
    __device__ __forceinline__ void set_ef(u64 at, const EF& v) {
        u32 w = v.v[0];
#pragma unroll
        for (int k = 1; k < 5; k++) w = lane == (u32)k ? v.v[k] : w;
        set_lanes(lane < 5, at + lane, w);
    }
    __device__ __forceinline__ bool make_slices_equal_and_defined(u64 p0, u64 p1, u32 len) {  // memory.rs:41-66
        for (u32 i = 0; i < len; i++) {
            const u32 v0 = peek_u(p0 + i), v1 = peek_u(p1 + i);
            if (v0 != UNDEF && v1 != UNDEF) {
                if (v0 != v1) {
                    fail(VM_E_NOT_EQUAL, p0 + i);
                    return false;
                }
            } else if (v0 != UNDEF)
                set_u(p1 + i, v0);
            else if (v1 != UNDEF)
                set_u(p0 + i, v1);
            else {
                set_u(p0 + i, 0);
                if (!err) set_u(p1 + i, 0);
            }
            if (err) return false;
        }
        return true;
    }
    __device__ __forceinline__ bool solve_unknowns(u64 pa, u64 pb, u64 pr, bool is_be, u32 op) {  // exec.rs:29-104
        EF a, b, c;
        bool ka, kb_, kc;
        if (is_be) {
            const u32 v = peek_u(pa);
            ka = v != UNDEF;
            a = ef_from_base(ka ? v : 0);
        } else
            ka = peek_ef(pa, a);
        kb_ = peek_ef(pb, b);
        kc = peek_ef(pr, c);
        if (op == OP_MUL && !is_be) {  // "copy_5"
            if (kb_ && ef_eq(b, ef_one())) return make_slices_equal_and_defined(pa, pr, 5);
            if (ka && ef_eq(a, ef_one())) return make_slices_equal_and_defined(pb, pr, 5);
        }
        if (ka && kb_ && kc) {
            if (!ef_eq(compute_elem(a, b, op), c)) {
                fail(VM_E_EXTENSION_OP, 0);
                return false;
            }
        } else if (ka && kb_ && !kc) {
        } else if (!ka && kb_ && kc) {
            const EF x = op == OP_ADD ? ef_sub(c, b) : ef_mul(c, ef_inv_d(b));
            if (is_be) {
                if (x.v[1] | x.v[2] | x.v[3] | x.v[4]) {
                    fail(VM_E_EXTENSION_OP, 1);
                    return false;
                }
                set_u(pa, x.v[0]);
            } else
                set_ef(pa, x);
            return !err;
        } else if (ka && !kb_ && kc) {
            const EF x = op == OP_ADD ? ef_sub(c, a) : ef_mul(c, ef_inv_d(a));
            set_ef(pb, x);
            return !err;
        } else {
            fail(VM_E_EXTENSION_OP, 2);
            return false;
        }
        return true;
    }
    __device__ __forceinline__ void extension_op(const VmInstr& in, u32 va, u32 vb, u32 vc) {  // exec_multi_row (exec.rs:106-190)
        const u32 op = in.x0 & (OP_ADD | OP_MUL | OP_POLY_EQ);
        const bool is_be = in.x0 & 4;
        const u64 size = in.x1, pa = usize(va), pb = usize(vb), pr = usize(vc);
        if (size == 1 && op != OP_POLY_EQ && !solve_unknowns(pa, pb, pr, is_be, op)) return;
        if ((u64)n_ext + size > k_cap_ext) {
            fail(VM_E_LOG_CAPACITY, 3);
            return;
        }
        const u64 a_stride = is_be ? 1 : 5;
        u32* rows = ext + (u64)n_ext * LM_VM_EXTENSION_ROW_WORDS;
        // comp[i] = elem[i] (+ or *) comp[i + 1]: one backward pass writes every row except comp[0], which follows
        EF comp = ef_zero();
        for (u64 i = size; i-- > 0;) {
            EF a, b;
            if (is_be) {
                const u32 v = need_mem(pa + i);
                if (err) return;
                a = ef_from_base(v);
            } else if (!need_ef(pa + i * a_stride, a))
                return;
            if (!need_ef(pb + i * 5, b)) return;
            const EF e = compute_elem(a, b, op);
            comp = i == size - 1 ? e : (op == OP_POLY_EQ ? ef_mul(e, comp) : ef_add(e, comp));
            if (lane == 0) {
                u32* r = rows + i * LM_VM_EXTENSION_ROW_WORDS;
                r[0] = is_be, r[1] = i == 0, r[2] = op == OP_ADD, r[3] = op == OP_MUL, r[4] = op == OP_POLY_EQ, r[5] = (u32)(size - i);
                r[6] = (u32)(pa + i * a_stride), r[7] = (u32)(pb + i * 5), r[8] = (u32)pr;
#pragma unroll
                for (int k = 0; k < 5; k++) r[9 + k] = b.v[k], r[19 + k] = comp.v[k];
            }
        }
        set_ef(pr, comp);
        if (err) return;
        for (u64 i = lane; i < size * 5; i += 64) rows[(i / 5) * LM_VM_EXTENSION_ROW_WORDS + 14 + (i % 5)] = comp.v[i % 5];
        n_ext += (u32)size;
    }

    // ---- one instruction (isa/instruction.rs:146-246) ------------------------------------------------------------------------------------------
    __device__ __forceinline__ void step(const VmInstr& in) {
        switch (in.kind) {
            case VM_K_ADD:
            case VM_K_MUL: {  // nu_a (arg_a) op nu_c (arg_c) = nu_b (res)
                const bool mu = in.kind == VM_K_MUL;
                // the three operand reads are independent and free of side effects: issued together (one LDS round trip, not three)
                const u32 r = read(in.mb, in.b, in.bm), a = read(in.ma, in.a, in.am), c = read(in.mc, in.c, in.cm);
                if (r == UNDEF) {
                    if (a == UNDEF) {
                        fail(VM_E_UNDEFINED_MEMORY, fp + in.a);
                        return;
                    }
                    if (c == UNDEF) {
                        fail(VM_E_UNDEFINED_MEMORY, fp + in.c);
                        return;
                    }
                    set_u(fp + in.b, mu ? mul(a, c) : add(a, c));
                } else {
                    if (a == UNDEF) {  // a = res inv_op c
                        if (c == UNDEF) {
                            fail(VM_E_UNDEFINED_MEMORY, fp + in.c);
                            return;
                        }
                        if (mu && c == 0) {
                            fail(VM_E_DIV_BY_ZERO, 0);
                            return;
                        }
                        set_u(fp + in.a, mu ? mul(r, inv(c)) : sub(r, c));
                    } else {
                        if (c == UNDEF) {
                            if (in.mc != LM_VM_ARG_MEM) {
                                fail(VM_E_NOT_A_POINTER, 0);
                                return;
                            }
                            if (mu && a == 0) {
                                fail(VM_E_DIV_BY_ZERO, 0);
                                return;
                            }
                            set_u(fp + in.c, mu ? mul(r, inv(a)) : sub(r, a));
                        } else {
                            const u32 v = mu ? mul(a, c) : add(a, c);
                            if (v != r) {
                                fail(VM_E_NOT_EQUAL, 0);
                                return;
                            }
                        }
                    }
                }
                if (err) return;
                if (mu)
                    n_mul++;
                else
                    n_add++;
                pc++;
                break;
            }
            case VM_K_DEREF: {  // res = m[m[fp + shift_0] + shift_1]
                const u32 r = read(in.mc, in.c, in.cm);
                if (r == UNDEF) {
                    if (in.mc != LM_VM_ARG_MEM) {
                        fail(VM_E_NOT_A_POINTER, 0);
                        return;
                    }
                    const u32 p = need_mem(fp + in.a);
                    if (err) return;
                    const u32 v = peek_u(usize(p) + in.b);
                    if (v != UNDEF) set_u(fp + in.c, v);
                    // else: a range check, resolved by resolve_deref_hints
                } else {
                    const u32 p = need_mem(fp + in.a);
                    if (err) return;
                    set_u(usize(p) + in.b, r);
                }
                if (err) return;
                n_deref++;
                pc++;
                break;
            }
            case VM_K_JUMP: {
                const u32 cond = need(in.ma, in.a, in.am);
                if (err) return;
                if (cond == 0)
                    pc++;
                else if (cond == ONE) {
                    const u32 d = need(in.mb, in.b, in.bm);
                    if (err) return;
                    const u32 f = need(in.mc, in.c, in.cm);
                    if (err) return;
                    pc = (u32)usize(d);
                    fp = usize(f);
                } else {
                    fail(VM_E_JUMP_CONDITION, 0);
                    return;
                }
                n_jump++;
                break;
            }
            default: {
                const u32 a = read(in.ma, in.a, in.am), b = read(in.mb, in.b, in.bm), c = read(in.mc, in.c, in.cm);
                if (a == UNDEF || b == UNDEF || c == UNDEF) {
                    fail(VM_E_UNDEFINED_MEMORY, fp + (a == UNDEF ? in.a : (b == UNDEF ? in.b : in.c)));
                    return;
                }
                if (in.kind == VM_K_POSEIDON)
                    poseidon(in, a, b, c);
                else
                    extension_op(in, a, b, c);
                if (err) return;
                pc++;
                break;
            }
        }
    }

    // run_loop (runner.rs:121-204) from batch_pc until the loop comes back to it
    __device__ __forceinline__ void run() {
        for (;;) {
            if (pc == k_ending_pc) {
                fail(VM_E_REACHED_END, 0);
                return;
            }
            if (pc >= k_n_instructions) {
                fail(VM_E_PC_OUT_OF_BOUNDS, 0);
                return;
            }
            if (n_cyc >= k_cap_cyc) {
                fail(VM_E_LOG_CAPACITY, 1);
                return;
            }
            if (lane == 0 && !dbg_bit(4)) pcs[n_cyc] = pc, fps[n_cyc] = (u32)fp;
            n_cyc++;
            if (pc - win_base >= win_n) {  // refill the instruction window at pc
                win_base = pc;
                win_n = min((u32)VM_WIN, k_n_instructions - pc);
                const u32* src = reinterpret_cast<const u32*>(k_code + pc);
                for (u32 k = lane; k < win_n * 9; k += 64) wcode[k] = src[k];
                for (u32 k = lane; k <= win_n; k += 64) whb[k] = k_hint_begin[pc + k];
            }
            const u32 wi = pc - win_base;
            const u32 h0 = rfl(whb[wi]), h1 = rfl(whb[wi + 1]);
            for (u32 h = h0; h < h1; h++) {
                if (h - hwin_base >= hwin_n) {  // refill the hint window at h
                    hwin_base = h;
                    hwin_n = min((u32)VM_HWIN, k_n_hints - h);
                    const u32* src = reinterpret_cast<const u32*>(k_hints + h);
                    for (u32 k = lane; k < hwin_n * 6; k += 64) whint[k] = src[k];
                }
                VmHintRec hr;
                {
                    const u32* src = whint + (h - hwin_base) * 6;
                    u32* dst = reinterpret_cast<u32*>(&hr);
#pragma unroll
                    for (int k = 0; k < 6; k++) dst[k] = rfl(src[k]);
                }
                if (hr.kind == LM_VM_HINT_PARALLEL_BATCH_START) {
                    if (pc != k_batch_pc) {  // an inner batch: left to the host runner
                        fail(VM_E_NESTED_BATCH, 0);
                        return;
                    }
                    continue;
                }
                run_hint(hr);
                if (err) return;
            }
            VmInstr in;
            {
                const u32* src = wcode + wi * 9;
                u32* dst = reinterpret_cast<u32*>(&in);
#pragma unroll
                for (int k = 0; k < 9; k++) dst[k] = rfl(src[k]);
            }
            step(in);
            if (err) return;
            pc = rfl(pc);
            fp = rfl(fp);
            if (pc == k_batch_pc) return;  // StopReason::LoopBack
        }
    }
};

// blockDim.x / 64 segments per workgroup: the waves are independent after the common set-up (table, prefix cache), each on its own
// LDS slice; LDS layout in words: [coop table | prefix cache | per wave: frame, cursors, instruction window, hint window]
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_vm_segments(const VmSegArgs A, u32 n_par) {
    extern __shared__ u32 lds[];
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 seg = blockIdx.x * (blockDim.x >> 6) + wave;
    const u32 stride_pad = (u32)((A.stride + 1) & ~1ull);
    const u32 pc_pad = (A.prefix_cache + 1) & ~1u;
    u32* tab = lds;
    u32* pcache = lds + COOP_TAB_WORDS;
    for (u32 k = threadIdx.x; k < COOP_TAB_WORDS; k += blockDim.x) {  // transposed lane tables behind the 16 uniform words (Machine::poseidon)
        const u32 l = (k - 16) / COOP_TAB_STRIDE, i = (k - 16) % COOP_TAB_STRIDE;
        tab[k < 16 ? k : 16 + i * 16 + l] = A.coop_tab[k];
    }
    for (u32 k = threadIdx.x; k < A.prefix_cache; k += blockDim.x) pcache[k] = A.image[k];
    if (blockIdx.x == 0 && threadIdx.x < VM_SUMMARY_WORDS) A.summary[threadIdx.x] = 0;  // (k_vm_apply_deferred follows on the stream)
    __syncthreads();
    if (seg >= n_par) return;
    u32* mine = lds + COOP_TAB_WORDS + pc_pad + wave * vm_wave_lds_words(stride_pad);
    Machine m(A);
    m.hot_args();
    m.coop_tab = tab, m.pcache = pcache;
    m.frame = mine;
    m.cursors = reinterpret_cast<u64*>(mine + stride_pad);
    m.wcode = mine + stride_pad + 2 * VM_DEV_MAX_NAMES;
    m.whb = m.wcode + VM_WIN * 9;
    m.whint = m.whb + VM_WIN + 2;
    m.seg_start = (u32)(A.split_at + (u64)seg * A.stride);
    m.lane = lane;
    for (u32 k = lane; k < A.stride; k += 64) {
        const u32 a = m.seg_start + k;
        m.frame[k] = a < A.init_len ? A.image[a] : UNDEF;
    }
    for (u32 k = lane; k < A.n_names; k += 64) m.cursors[k] = A.cur_index[k] + (u64)seg * A.per_iter[k];
    m.pcs = A.pcs + (u64)seg * A.cap_cyc, m.fps = A.fps + (u64)seg * A.cap_cyc;
    m.pos = A.pos + (u64)seg * A.cap_pos * LM_VM_POSEIDON_CALL_WORDS, m.ext = A.ext + (u64)seg * A.cap_ext * LM_VM_EXTENSION_ROW_WORDS;
    m.pend = A.pend + (u64)seg * A.cap_pend * 2, m.def = A.def + (u64)seg * A.cap_def * 2;
    // write_call_frame (runner.rs:353-367) for iteration seg + 1: its frame starts at the segment's own slice
    m.pc = A.batch_pc;
    m.fp = m.seg_start;
    m.ap = m.fp + A.frame_size;
    m.set_u(m.fp, A.return_pc_m);
    if (!m.err) m.set_u(m.fp + 1, A.saved_fp_m);
    if (!m.err) m.set_u(m.fp + 2, to_monty_d((u32)((A.start_value + seg + 1) % P)));
    for (u32 j = 1; j < A.n_args && !m.err; j++) m.set_u(m.fp + 2 + j, A.args_m[j]);
    if (!m.err) m.run();
    for (u32 k = lane; k < A.stride; k += 64) A.image[m.seg_start + k] = m.frame[k];
    if (lane == 0) {
        u32* c = A.counts + (u64)seg * VM_SEG_WORDS;
        c[VM_SEG_CYC] = m.n_cyc, c[VM_SEG_POS] = m.n_pos, c[VM_SEG_EXT] = m.n_ext, c[VM_SEG_PEND] = m.n_pend, c[VM_SEG_DEF] = m.n_def;
        c[VM_SEG_ADD] = m.n_add, c[VM_SEG_MUL] = m.n_mul, c[VM_SEG_DEREF] = m.n_deref, c[VM_SEG_JUMP] = m.n_jump;
        c[VM_SEG_ERR] = m.err, c[VM_SEG_ERR_PC] = m.pc, c[VM_SEG_ERR_AUX] = m.err_aux;
    }
}

// Summary block of a batch (u32 words) followed by the dirty list: what comes back to the host after the segments ran
//   [0] conflicting deferred writes  [1] dirty entries  [2] deferred writes the device cannot place (beyond the image)
//   [3] first failed segment + 1 (0 = none)  [4] its error code  [5] pc  [6] aux
//   [8..16) totals as u64: cycles, Poseidon calls, extension rows, pending derefs  [16..24) as u64: ADD, MUL, DEREF, JUMP
//   [VM_SUMMARY_WORDS ..) dirty list: (address, value) of cells OUTSIDE the window that a deferred write defined
// Deferred writes (SegmentMemory::into_deferred_writes applied with Memory::set, runner.rs:466-472): every cell is write-once, so
// the order of application only decides WHICH of two conflicting writes fails; any conflict sends the batch to the host runner.
__global__ __launch_bounds__(256) void k_vm_apply_deferred(u32* __restrict__ image, u64 image_cap, const u32* __restrict__ def, const u32* __restrict__ counts,
                                                           u32 cap_def, u64 lo, u64 hi, u32* __restrict__ summary, u32 dirty_cap) {
    const u32 seg = blockIdx.x;
    const u32 n = counts[(u64)seg * VM_SEG_WORDS + VM_SEG_DEF];
    const u32* d = def + (u64)seg * cap_def * 2;
    for (u32 k = threadIdx.x; k < n; k += 256) {
        const u64 a = d[2 * k];
        const u32 v = d[2 * k + 1];
        if (a >= image_cap) {
            atomicAdd(summary + 2, 1u);
            continue;
        }
        const u32 old = atomicCAS(image + a, UNDEF, v);
        if (old != UNDEF) {
            if (old != v) atomicAdd(summary, 1u);
        } else if (a < lo || a >= hi) {
            const u32 at = atomicAdd(summary + 1, 1u);
            if (at < dirty_cap) summary[VM_SUMMARY_WORDS + 2 * at] = (u32)a, summary[VM_SUMMARY_WORDS + 2 * at + 1] = v;
        }
    }
}
// exclusive prefix sums of the per-segment counts (offsets of Trace::merge), totals, first error
__global__ __launch_bounds__(1024) void k_vm_summary(const u32* __restrict__ counts, u32 n_par, u64* __restrict__ offs, u32* __restrict__ summary) {
    __shared__ u64 part[1024][4];
    __shared__ u64 tot8[8];
    __shared__ u32 first_err;
    const u32 t = threadIdx.x;
    const u32 per = (n_par + 1023) / 1024;
    const u32 b = t * per, e = min(n_par, b + per);
    u64 s[4] = {0, 0, 0, 0}, ops[4] = {0, 0, 0, 0};
    u32 bad = 0xFFFFFFFFu;
    for (u32 i = b; i < e; i++) {
        const u32* c = counts + (u64)i * VM_SEG_WORDS;
        s[0] += c[VM_SEG_CYC], s[1] += c[VM_SEG_POS], s[2] += c[VM_SEG_EXT], s[3] += c[VM_SEG_PEND];
        ops[0] += c[VM_SEG_ADD], ops[1] += c[VM_SEG_MUL], ops[2] += c[VM_SEG_DEREF], ops[3] += c[VM_SEG_JUMP];
        if (c[VM_SEG_ERR] && bad == 0xFFFFFFFFu) bad = i;
    }
    if (t == 0) {
        first_err = 0xFFFFFFFFu;
        for (int k = 0; k < 8; k++) tot8[k] = 0;
    }
    for (int k = 0; k < 4; k++) part[t][k] = s[k];
    __syncthreads();
    atomicMin(&first_err, bad);
    for (int k = 0; k < 4; k++) atomicAdd((unsigned long long*)&tot8[4 + k], (unsigned long long)ops[k]);
    if (t < 4) {  // serial scan of the 1024 partial sums of one kind (n_par is a few thousand)
        u64 run = 0;
        for (u32 j = 0; j < 1024; j++) {
            const u64 x = part[j][t];
            part[j][t] = run;
            run += x;
        }
        tot8[t] = run;
    }
    __syncthreads();
    u64 run[4] = {part[t][0], part[t][1], part[t][2], part[t][3]};
    for (u32 i = b; i < e; i++) {
        const u32* c = counts + (u64)i * VM_SEG_WORDS;
        for (int k = 0; k < 4; k++) offs[(u64)i * 4 + k] = run[k];
        run[0] += c[VM_SEG_CYC], run[1] += c[VM_SEG_POS], run[2] += c[VM_SEG_EXT], run[3] += c[VM_SEG_PEND];
    }
    if (t == 0) {
        if (first_err != 0xFFFFFFFFu) {
            const u32* c = counts + (u64)first_err * VM_SEG_WORDS;
            summary[3] = first_err + 1, summary[4] = c[VM_SEG_ERR], summary[5] = c[VM_SEG_ERR_PC], summary[6] = c[VM_SEG_ERR_AUX];
        } else
            summary[3] = 0;
        u64* o = reinterpret_cast<u64*>(summary + 8);
        for (int k = 0; k < 8; k++) o[k] = tot8[k];
    }
}

// Trace::merge in iteration order: the segment slots into contiguous arrays
__global__ __launch_bounds__(256) void k_vm_splice(const u32* __restrict__ counts, const u64* __restrict__ offsets, const u32* __restrict__ s_pcs,
                                                   const u32* __restrict__ s_fps, const u32* __restrict__ s_pos, const u32* __restrict__ s_ext,
                                                   const u32* __restrict__ s_pend, u32 cap_cyc, u32 cap_pos, u32 cap_ext, u32 cap_pend, u64 b_cyc, u64 b_pos,
                                                   u64 b_ext, u64 b_pend, u32* __restrict__ pcs, u32* __restrict__ fps, u32* __restrict__ pos,
                                                   u32* __restrict__ ext, u32* __restrict__ pend) {
    const u32 seg = blockIdx.x, t = threadIdx.x;
    const u32* c = counts + (u64)seg * VM_SEG_WORDS;
    const u64* o = offsets + (u64)seg * 4;
    const u32 n_cyc = c[VM_SEG_CYC], n_pos = c[VM_SEG_POS] * LM_VM_POSEIDON_CALL_WORDS, n_ext = c[VM_SEG_EXT] * LM_VM_EXTENSION_ROW_WORDS,
              n_pend = c[VM_SEG_PEND] * 2;
    for (u32 k = t; k < n_cyc; k += 256) {
        pcs[b_cyc + o[0] + k] = s_pcs[(u64)seg * cap_cyc + k];
        fps[b_cyc + o[0] + k] = s_fps[(u64)seg * cap_cyc + k];
    }
    for (u32 k = t; k < n_pos; k += 256) pos[(b_pos + o[1]) * LM_VM_POSEIDON_CALL_WORDS + k] = s_pos[(u64)seg * cap_pos * LM_VM_POSEIDON_CALL_WORDS + k];
    for (u32 k = t; k < n_ext; k += 256) ext[(b_ext + o[2]) * LM_VM_EXTENSION_ROW_WORDS + k] = s_ext[(u64)seg * cap_ext * LM_VM_EXTENSION_ROW_WORDS + k];
    for (u32 k = t; k < n_pend; k += 256) pend[(b_pend + o[3]) * 2 + k] = s_pend[(u64)seg * cap_pend * 2 + k];
}

// resolve_deref_hints (runner.rs:206-236): memory[target] = memory[memory[src]], repeated until a round resolves nothing
__global__ __launch_bounds__(256) void k_vm_resolve_round(u32* __restrict__ image, u64 len, const u32* __restrict__ pend, u64 n, uint8_t* __restrict__ status,
                                                          u32* __restrict__ info, u32 round) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || status[i]) return;
    const u64 target = pend[2 * i], src = pend[2 * i + 1];
    const u32 a = src < len ? __hip_atomic_load(image + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : UNDEF;
    if (a == UNDEF || target >= len) {  // memory.0[src_addr].unwrap() panics: the host runner reports it
        atomicAdd(info, 1u);
        status[i] = 1;
        return;
    }
    const u64 addr = from_monty(a);
    const u32 v = addr < len ? __hip_atomic_load(image + addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : UNDEF;
    if (v == UNDEF) return;
    const u32 old = atomicCAS(image + target, UNDEF, v);
    if (old != UNDEF && old != v) atomicAdd(info, 1u);
    status[i] = 1;
    atomicAdd(info + 1 + round, 1u);
}
// after the fixpoint: targets still unresolved get 0.  Acts only when round `last_round` resolved nothing (the host launches a few
// rounds and this kernel back to back and reads the counters once).
__global__ __launch_bounds__(256) void k_vm_resolve_finish(u32* __restrict__ image, u64 len, const u32* __restrict__ pend, u64 n,
                                                           const uint8_t* __restrict__ status, u32* __restrict__ info, u32 last_round) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || status[i]) return;
    if (__hip_atomic_load(info + 1 + last_round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const u64 target = pend[2 * i];
    if (target >= len) {
        atomicAdd(info, 1u);
        return;
    }
    const u32 old = atomicCAS(image + target, UNDEF, 0u);
    if (old != UNDEF && old != 0u) atomicAdd(info, 1u);
}

// (n_total > n: the committed memory column — the image, then `tail` (24 words: [0 x 16 | poseidon16(0)], trace_gen.rs:106-110), then
// zeros up to the padded length — in one launch instead of a kernel, a copy and a fill)
struct VmTail {
    u32 w[24];
};
__global__ __launch_bounds__(256) void k_vm_image_export(u32* __restrict__ dst, const u32* __restrict__ src, u64 n, uint8_t* __restrict__ defined,
                                                         u64 n_total, VmTail tail) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_total; i += (u64)gridDim.x * 256) {
        if (i < n) {
            const u32 v = src[i];
            dst[i] = v == UNDEF ? 0u : v;
            if (defined) defined[i] = v != UNDEF;
        } else {
            dst[i] = i - n < 24 ? tail.w[i - n] : 0u;
        }
    }
}
}  // namespace

namespace lmh {
int vm_dev_segments(lm_ctx* ctx, const VmSegArgs& a, u64 n_par) {
    LM_REQUIRE(ctx && n_par > 0 && n_par < (1ull << 31) && a.stride > 0 && a.stride <= VM_DEV_MAX_STRIDE && a.n_names <= VM_DEV_MAX_NAMES &&
               a.n_args <= VM_DEV_MAX_ARGS && a.prefix_cache <= VM_DEV_PREFIX_CACHE && a.prefix_cache <= a.split_at);
    // segments per workgroup: as many (<= 4) as keep two workgroups on a CU (160 KB of LDS, 8 waves at 2 per SIMD)
    static const bool attr_ok = hipFuncSetAttribute((const void*)k_vm_segments, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    const u32 stride_pad = (u32)((a.stride + 1) & ~1ull);
    const size_t shared_words = (size_t)COOP_TAB_WORDS + ((a.prefix_cache + 1) & ~1u);
    u32 spw = 4;
    while (spw > 1 && (shared_words + (size_t)spw * vm_wave_lds_words(stride_pad)) * 4 > (attr_ok ? 80u : 64u) * 1024) spw >>= 1;
    const size_t lds_bytes = (shared_words + (size_t)spw * vm_wave_lds_words(stride_pad)) * 4;
    LM_REQUIRE(lds_bytes <= (attr_ok ? 160u : 64u) * 1024);
    (void)hipGetLastError();
    LM_LAUNCH(ctx, k_vm_segments, dim3((unsigned)((n_par + spw - 1) / spw)), dim3(64 * spw), lds_bytes, a, (u32)n_par);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
int vm_dev_apply_deferred(lm_ctx* ctx, const VmSegArgs& a, u64 n_par, u64 image_cap, u64 lo, u64 hi, u64* d_offsets, u32* d_summary, u32 dirty_cap) {
    LM_REQUIRE(ctx && d_offsets && d_summary && n_par > 0);
    LM_LAUNCH(ctx, k_vm_apply_deferred, dim3((unsigned)n_par), dim3(256), 0, a.image, image_cap, (const u32*)a.def, (const u32*)a.counts, a.cap_def, lo, hi,
              d_summary, dirty_cap);
    LM_LAUNCH(ctx, k_vm_summary, dim3(1), dim3(1024), 0, (const u32*)a.counts, (u32)n_par, d_offsets, d_summary);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
int vm_dev_splice(lm_ctx* ctx, const VmSegArgs& a, u64 n_par, const u64* d_offsets, const u64 base[4], u32* pcs, u32* fps, u32* pos, u32* ext, u32* pend) {
    LM_REQUIRE(ctx && d_offsets && n_par > 0);
    LM_LAUNCH(ctx, k_vm_splice, dim3((unsigned)n_par), dim3(256), 0, (const u32*)a.counts, d_offsets, (const u32*)a.pcs, (const u32*)a.fps, (const u32*)a.pos,
              (const u32*)a.ext, (const u32*)a.pend, a.cap_cyc, a.cap_pos, a.cap_ext, a.cap_pend, base[0], base[1], base[2], base[3], pcs, fps, pos, ext,
              pend);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
int vm_dev_resolve(lm_ctx* ctx, u32* image, u64 image_len, const u32* pend, u64 n, uint8_t* status, u32* d_info, u32 first_round, u32 n_rounds) {
    if (n == 0) return LM_OK;
    LM_REQUIRE(first_round + n_rounds < VM_RESOLVE_INFO_WORDS - 1 && n_rounds > 0);
    const dim3 grid((unsigned)((n + 255) / 256));
    for (u32 r = first_round; r < first_round + n_rounds; r++) LM_LAUNCH(ctx, k_vm_resolve_round, grid, dim3(256), 0, image, image_len, pend, n, status, d_info, r);
    LM_LAUNCH(ctx, k_vm_resolve_finish, grid, dim3(256), 0, image, image_len, pend, n, (const uint8_t*)status, d_info, first_round + n_rounds - 1);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
// The host's parts of a run's log (sequential head and tail: a few thousand cycles, their precompile calls, the pending list) go up in
// ONE step: the pieces are packed into the pinned staging ring and a kernel reads them from there and stores each at its place between
// the spliced segment logs; the same launch zeroes the resolver's status bytes and counters (ten copy / fill commands before).
struct VmPlaceArgs {
    u32 n;
    u32* dst[VM_PLACE_MAX];
    u64 src_off[VM_PLACE_MAX], n_words[VM_PLACE_MAX];  // (src_off = ~0: zero fill)
};
__global__ __launch_bounds__(256) void k_vm_place(const u32* __restrict__ src, VmPlaceArgs a) {
    const u32 part = blockIdx.y;
    if (part >= a.n) return;
    u32* __restrict__ dst = a.dst[part];
    const u64 n = a.n_words[part], off = a.src_off[part];
    const bool zero = off == ~0ull;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) dst[i] = zero ? 0u : src[off + i];
}
int vm_dev_place(lm_ctx* ctx, const VmPart* parts, u32 n_parts) {
    u64 total = 0;
    for (u32 i = 0; i < n_parts; i++)
        if (parts[i].src) total += (parts[i].n_words + 15) & ~15ull;
    void* st = nullptr;
    if (total) {
        int rc = lm_stage_alloc(ctx, total * 4, &st);
        if (rc) return rc;
    }
    if (total && !st) {  // no room in the ring: plain copies (synchronous semantics of the sources are the caller's: they stay alive)
        for (u32 i = 0; i < n_parts; i++) {
            if (!parts[i].n_words) continue;
            if (parts[i].src)
                LM_HIP(hipMemcpyAsync(parts[i].dst, parts[i].src, parts[i].n_words * 4, hipMemcpyHostToDevice, ctx->stream));
            else
                LM_HIP(hipMemsetAsync(parts[i].dst, 0, parts[i].n_words * 4, ctx->stream));
        }
        return LM_OK;
    }
    u32* img = static_cast<u32*>(st);
    u64 at = 0;
    for (u32 p0 = 0; p0 < n_parts; p0 += VM_PLACE_MAX) {
        VmPlaceArgs a;
        a.n = 0;
        u64 longest = 0;
        for (u32 i = p0; i < n_parts && i < p0 + VM_PLACE_MAX; i++) {
            if (!parts[i].n_words) continue;
            a.dst[a.n] = parts[i].dst, a.n_words[a.n] = parts[i].n_words;
            if (parts[i].src) {
                memcpy(img + at, parts[i].src, parts[i].n_words * 4);
                a.src_off[a.n] = at;
                at += (parts[i].n_words + 15) & ~15ull;
            } else {
                a.src_off[a.n] = ~0ull;
            }
            longest = std::max<u64>(longest, parts[i].n_words);
            a.n++;
        }
        if (!a.n) continue;
        const unsigned bx = (unsigned)std::min<u64>((longest + 255) / 256, 64);
        LM_LAUNCH(ctx, k_vm_place, dim3(bx, a.n), dim3(256), 0, (const u32*)img, a);
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}
int vm_dev_fill(lm_ctx* ctx, u32* d, u32 word, u64 n) {
    if (n == 0) return LM_OK;
    LM_HIP(hipMemsetD32Async((hipDeviceptr_t)d, (int)word, n, ctx->stream));
    return LM_OK;
}
int vm_dev_download(lm_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
    LM_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP(hipStreamSynchronize(ctx->stream));
    return LM_OK;
}
int vm_dev_upload(lm_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    if (bytes == 0) return LM_OK;
    LM_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return LM_OK;
}
int vm_dev_mark(lm_ctx* ctx) {
    LM_HIP(hipEventRecord(ctx->fork_event, ctx->stream));
    return LM_OK;
}
int vm_dev_wait_mark(lm_ctx* ctx) {
    LM_HIP(hipEventSynchronize(ctx->fork_event));
    return LM_OK;
}
const u32* vm_dev_coop_table(lm_ctx* ctx) { return ctx->d_coop; }
int vm_dev_image_export(lm_ctx* ctx, u32* dst, const u32* src, u64 n, uint8_t* defined, u64 n_total, const u32* tail24) {
    if (n_total < n) n_total = n;
    if (n_total == 0) return LM_OK;
    VmTail t;
    memset(&t, 0, sizeof t);
    if (tail24) memcpy(t.w, tail24, sizeof t.w);
    const unsigned blocks = (unsigned)std::min<u64>((n_total + 255) / 256, 8192);
    LM_LAUNCH(ctx, k_vm_image_export, dim3(blocks), dim3(256), 0, dst, src, n, defined, n_total, t);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
}  // namespace lmh
