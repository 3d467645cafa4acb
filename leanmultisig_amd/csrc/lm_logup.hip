// logup numerators / denominators (reference: crates/sub_protocols/src/logup.rs:88-199), natural order, one launch.
#include <algorithm>
#include "lm_common.h"

using namespace kb;

struct LogupSec {
    u64 out_offset, chunk_begin;
    const u32* num_col;
    const u32* data[LM_LOGUP_MAX_DATA];
    u32 stride[LM_LOGUP_MAX_DATA];
    u32 add_m[LM_LOGUP_MAX_DATA];  // Montgomery
    u32 log_len, num_mode, n_data, neg_den;  // num_mode 4: a hole between sections, filled with the neutral pair (0, 1)
    u64 len;
    EF contrib;  // alpha_eq[15] * domsep
};

static constexpr u32 LG_CHUNK = 1024;

__global__ __launch_bounds__(256) void k_logup_fill(const LogupSec* __restrict__ secs, u32 n_secs, const EF* __restrict__ alphas,
                                                    EF c, u64 plane, u32* __restrict__ nums, u32* __restrict__ dens, u32 r2) {
    u32 lo = 0, hi = n_secs - 1;
    const u64 b = blockIdx.x;
    while (lo < hi) {
        u32 mid = (lo + hi + 1) >> 1;
        if (secs[mid].chunk_begin <= b)
            lo = mid;
        else
            hi = mid - 1;
    }
    const LogupSec& s = secs[lo];
    const u64 len = s.len;
    const u64 base = (b - s.chunk_begin) * LG_CHUNK;
    for (u32 u = 0; u < LG_CHUNK / 256; u++) {
        const u64 i = base + u * 256 + threadIdx.x;
        if (i >= len) break;
        if (s.num_mode == 4) {  // neutral pair: everything outside the sections (bytecode padding, holes, the tail)
            nums[s.out_offset + i] = 0;
            dens[s.out_offset + i] = ONE;
#pragma unroll
            for (int k = 1; k < 5; k++) dens[(u64)k * plane + s.out_offset + i] = 0;
            continue;
        }
        u32 n;
        if (s.num_mode == 0)
            n = 0;
        else if (s.num_mode == 1)
            n = ONE;
        else {
            n = s.num_col[i];
            if (s.num_mode == 3) n = neg(n);
        }
        EF fp = s.contrib;
        const u32 im = mul((u32)i, r2);  // Montgomery form of the row index (i < p)
        for (u32 j = 0; j < s.n_data; j++) {
            u32 d = s.data[j] ? s.data[j][i * s.stride[j]] : im;
            d = add(d, s.add_m[j]);
            fp = ef_add(fp, ef_mul_base(alphas[j], d));
        }
        const EF den = s.neg_den ? ef_sub(c, fp) : ef_add(c, fp);
        nums[s.out_offset + i] = n;
#pragma unroll
        for (int k = 0; k < 5; k++) dens[(u64)k * plane + s.out_offset + i] = den.v[k];
    }
}
// ---- access counters (crates/lean_prover/src/prove_execution.rs:90-110) ---------------------------------------------
// acc[index(row) + j] += 1 for every row of an index column and j < n_values: a histogram of ~12 M word accesses over the
// memory image.  The reference loops sequentially ("TODO parallelize").  Round 1 used one device-scope atomic per (pair of)
// counter(s): on this multi-XCD part those execute memory-side at ~2 G/s — 0.8 ms on round 1's synthetic trace (whole waves
// on one address) but 4.0 ms on the trace of real XMSS verifications, where every address is touched once or twice.
// Now: no atomics on HBM in the data path.  The address space is cut into windows of ACC_WIN counters that fit LDS;
//   1. k_acc_hist     per tile of rows: LDS histogram of window ids -> global totals (a handful of atomics per tile)
//   2. k_acc_scan     exclusive scan of the window totals (one workgroup)
//   3. k_acc_scatter  per tile: reserve a range of every window's list (one atomic per touched window), store the accesses
//                     as packed (n_values - 1, address) words
//   4. k_acc_window   one workgroup per window: LDS counters, then plain coalesced stores of the finished window (the
//                     window has exactly one owner); runs that straddle the end of a window leave <= 15 counts in a side array
//   5. k_acc_finish   adds those boundary counts and converts counts to field elements
struct AccessJob {
    const u32* index_col;  // Montgomery words, canonical value = address
    u64 n_rows;
    u32 n_values, tile_begin;  // first tile (workgroup) of this job
};
static constexpr u32 ACC_WIN_LOG = 13, ACC_WIN = 1u << ACC_WIN_LOG;  // 8192 counters = 32 KiB of LDS
static constexpr u32 ACC_TILE = 8192;                                // rows per workgroup in passes 1 and 3
static constexpr u32 ACC_MAX_RUN = 16, ACC_MAX_JOBS = 16;
struct AccessJobs {  // a kernel argument (no upload, no synchronisation)
    AccessJob j[ACC_MAX_JOBS];
    u32 n;
};
__device__ __forceinline__ bool acc_tile(const AccessJobs& jobs, AccessJob& jb, u64& r0) {
    u32 j = 0;
    while (j + 1 < jobs.n && jobs.j[j + 1].tile_begin <= blockIdx.x) j++;
    jb = jobs.j[j];
    r0 = (u64)(blockIdx.x - jb.tile_begin) * ACC_TILE;
    return r0 < jb.n_rows;
}
// MODE 0: count list items per window; MODE 1: scatter them.  `nb` windows.  dynamic LDS: nb words (+ nb in MODE 1).
// Runs of rows that read the SAME address are run-length encoded per wave (padding rows: a quarter of the execution table
// points at one cell — as plain items that cell's window would receive ~10^6 of them and its single owner in pass 4 would
// serialise them on one LDS counter): a wave whose 64 lanes agree extends its current run instead of emitting 64 items.
// item = (multiplicity - 1) << 17 | (n_values - 1) << 13 | address inside the window        (multiplicity <= 64 x 32)
template <int MODE>
__device__ __forceinline__ void acc_emit(u32 a, u32 mult, u32 n_values, u32* hist, const u32* base, u32* __restrict__ items) {
    const u32 w = a >> ACC_WIN_LOG;
    const u32 slot = atomicAdd(&hist[w], 1u);
    if (MODE == 1) items[base[w] + slot] = ((mult - 1) << 17) | ((n_values - 1) << 13) | (a & (ACC_WIN - 1));
}
// Each wave keeps one RUN (address, count) across its iterations: lanes that read the run's address only bump the count; the
// run ends when an iteration has no such lane.  Hot addresses are a feature of real traces, not only of padding: every
// instruction operand that is an immediate looks up memory[0] (trace_gen.rs:46-60), interleaved with real addresses.
template <int MODE>
__device__ __forceinline__ void acc_rows(const AccessJob& jb, u64 r0, u64 r1, u64 len, u32* hist, const u32* base, u32* __restrict__ items,
                                         u32* __restrict__ out_of_range) {
    const u32 lane = threadIdx.x & 63;
    u32 run_addr = 0, run_cnt = 0;  // wave-uniform
    for (u64 rb = r0; rb < r1; rb += 256) {
        const u64 r = rb + threadIdx.x;
        const u32 a = r < r1 ? from_monty(jb.index_col[r]) : 0u;
        const bool ok = r < r1 && (u64)a + jb.n_values <= len;
        if (MODE == 0 && out_of_range) {  // rows that point outside the image are skipped AND reported (the reference would panic)
            const u64 bad = __ballot(r < r1 && !ok);
            if (bad && lane == 0) atomicAdd(out_of_range, (u32)__popcll(bad));
        }
        u64 rest = __ballot(ok);
        if (!rest) continue;
        if (run_cnt) {
            const u64 m = __ballot(ok && a == run_addr);
            if (m) {
                run_cnt += (u32)__popcll(m);
                rest &= ~m;
            } else {
                if (lane == 0) acc_emit<MODE>(run_addr, run_cnt, jb.n_values, hist, base, items);
                run_cnt = 0;
            }
        }
        if (rest) {  // one more group: the lanes that agree with the first remaining one
            const u32 leader = (u32)__builtin_ctzll(rest);
            const u32 fa = (u32)__shfl((int)a, (int)leader, 64);
            const u64 g = __ballot(((rest >> lane) & 1) && a == fa);
            const u32 gn = (u32)__popcll(g);
            if (gn >= 2) {
                if (!run_cnt) {
                    run_addr = fa;
                    run_cnt = gn;
                } else if (lane == leader) {
                    acc_emit<MODE>(fa, gn, jb.n_values, hist, base, items);
                }
                rest &= ~g;
            }
            if ((rest >> lane) & 1) acc_emit<MODE>(a, 1u, jb.n_values, hist, base, items);
        }
    }
    if (run_cnt && lane == 0) acc_emit<MODE>(run_addr, run_cnt, jb.n_values, hist, base, items);
}
template <int MODE>
__global__ __launch_bounds__(256) void k_acc_pass(const AccessJobs jobs, u64 len, u32 nb, u32* __restrict__ totals,
                                                  u32* __restrict__ cursor, u32* __restrict__ items) {
    extern __shared__ u32 sh[];
    u32* hist = sh;            // per-window item count of this tile, then (MODE 1) running cursor inside the reserved range
    u32* base = sh + nb;       // MODE 1: start of this tile's range in each window's list
    AccessJob jb;
    u64 r0;
    const bool live = acc_tile(jobs, jb, r0);
    for (u32 i = threadIdx.x; i < nb; i += 256) hist[i] = 0;
    __syncthreads();
    const u64 r1 = live ? (r0 + ACC_TILE < jb.n_rows ? r0 + ACC_TILE : jb.n_rows) : 0;
    // (each wave walks a fixed quarter of the tile's iterations: the run-length state is per wave)
    acc_rows<0>(jb, r0, r1, len, hist, base, items, MODE == 0 ? totals + nb : nullptr);  // count (totals[nb]: rows outside the image)
    __syncthreads();
    if (MODE == 0) {
        for (u32 i = threadIdx.x; i < nb; i += 256)
            if (hist[i]) atomicAdd(&totals[i], hist[i]);
        return;
    }
    // reserve a range per touched window (the counts are exact: pass 3 repeats the walk of pass 1), then scatter
    for (u32 i = threadIdx.x; i < nb; i += 256) {
        base[i] = hist[i] ? atomicAdd(&cursor[i], hist[i]) : 0u;  // cursor starts at the window's offset (k_acc_scan)
        hist[i] = 0;
    }
    __syncthreads();
    acc_rows<1>(jb, r0, r1, len, hist, base, items, nullptr);
}
// offsets[i] = sum_{j < i} totals[j]; cursor = offsets (one workgroup, nb <= 8192)
__global__ __launch_bounds__(1024) void k_acc_scan(const u32* __restrict__ totals, u32 nb, u32* __restrict__ offsets, u32* __restrict__ cursor) {
    __shared__ u32 part[1024];
    const u32 per = (nb + 1023) / 1024, lo = threadIdx.x * per;
    u32 s = 0;
    for (u32 i = lo; i < lo + per && i < nb; i++) s += totals[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        const u32 v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    u32 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (u32 i = lo; i < lo + per && i < nb; i++) {
        offsets[i] = run;
        cursor[i] = run;
        run += totals[i];
    }
    if (threadIdx.x == 1023) offsets[nb] = part[1023];
}
// one workgroup per window: counters in LDS; cnt[w * ACC_WIN ..] stored once, boundary[w][0..16) = counts past the window's end.
// A window whose list is long (real traces have them: the window of the public input, the constants and the zero vector
// receives ~4 x 10^5 items where the others get ~6 x 10^3) is split between up to ACC_SPLIT workgroups, which ADD their
// non-zero counters to the (pre-zeroed) output instead — the only HBM atomics left, a few 10^4 per proof.
static constexpr u32 ACC_SPLIT = 16, ACC_HEAVY = 16384;
__global__ __launch_bounds__(256) void k_acc_window(const u32* __restrict__ items, const u32* __restrict__ offsets, u64 len, u32 nb,
                                                    u32* __restrict__ cnt, u32* __restrict__ boundary) {
    __shared__ u32 c[ACC_WIN + ACC_MAX_RUN];
    const u32 w = blockIdx.x % nb, part = blockIdx.x / nb;  // part-major: the workgroups that always have work come first
    const u32 k0 = offsets[w], k1 = offsets[w + 1], n = k1 - k0;
    const u32 parts = n <= ACC_HEAVY ? 1u : min(ACC_SPLIT, (n + ACC_HEAVY / 2 - 1) / (ACC_HEAVY / 2));
    if (part >= parts) return;
    const u32 per = (n + parts - 1) / parts, lo = k0 + part * per, hi = min(k1, lo + per);
    for (u32 i = threadIdx.x; i < ACC_WIN + ACC_MAX_RUN; i += 256) c[i] = 0;
    __syncthreads();
    const u32 w0 = w << ACC_WIN_LOG;
    for (u32 k = lo + threadIdx.x; k < hi; k += 256) {
        const u32 it = items[k];
        const u32 a = it & (ACC_WIN - 1), nv = ((it >> 13) & 15) + 1, mult = (it >> 17) + 1;
        for (u32 j = 0; j < nv; j++) atomicAdd(&c[a + j], mult);
    }
    __syncthreads();
    if (parts == 1) {
        for (u32 i = threadIdx.x; i < ACC_WIN; i += 256)
            if ((u64)w0 + i < len) cnt[(u64)w0 + i] = c[i];
        if (threadIdx.x < ACC_MAX_RUN) boundary[w * ACC_MAX_RUN + threadIdx.x] = c[ACC_WIN + threadIdx.x];
    } else {
        for (u32 i = threadIdx.x; i < ACC_WIN; i += 256)
            if (c[i] && (u64)w0 + i < len) atomicAdd(&cnt[(u64)w0 + i], c[i]);
        if (threadIdx.x < ACC_MAX_RUN && c[ACC_WIN + threadIdx.x]) atomicAdd(&boundary[w * ACC_MAX_RUN + threadIdx.x], c[ACC_WIN + threadIdx.x]);
    }
}
__global__ __launch_bounds__(256) void k_acc_finish(u32* __restrict__ v, u64 n, const u32* __restrict__ boundary,
                                                    const u32* __restrict__ out_of_range, u32* __restrict__ h_err) {
    // rows outside the image: added to the context's sticky error word in pinned memory (lm_access_errors)
    if (blockIdx.x == 0 && threadIdx.x == 0 && *out_of_range) lm_store_system(h_err, __hip_atomic_load(h_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + *out_of_range);
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        u32 x = v[i];
        const u32 w = (u32)(i >> ACC_WIN_LOG), o = (u32)i & (ACC_WIN - 1);
        if (boundary && w > 0 && o < ACC_MAX_RUN) x += boundary[(w - 1) * ACC_MAX_RUN + o];
        v[i] = to_monty(x);
    }
}

extern "C" int lm_access_counts(lm_ctx* ctx, uint32_t* d_acc, uint64_t len, uint32_t n_jobs, const uint32_t* const* d_index_cols,
                                const uint64_t* n_rows, const uint32_t* n_values) {
    LM_REQUIRE(ctx && d_acc && len > 0 && len < (1ull << 28));
    LM_REQUIRE(n_jobs == 0 || (d_index_cols && n_rows && n_values));
    const u32 fin_blocks = (unsigned)std::min<u64>((len + 255) / 256, 4096);
    if (n_jobs == 0) {
        LM_HIP(hipMemsetAsync(d_acc, 0, len * 4, ctx->stream));
        return LM_OK;  // zero is zero in Montgomery form
    }
    LM_REQUIRE(n_jobs <= ACC_MAX_JOBS);
    AccessJobs jobs{};
    jobs.n = n_jobs;
    u64 total_rows = 0;
    u32 tiles = 0;
    for (u32 i = 0; i < n_jobs; i++) {
        LM_REQUIRE(d_index_cols[i] && n_values[i] >= 1 && n_values[i] <= ACC_MAX_RUN && n_values[i] <= len && n_rows[i] < (1ull << 40));
        jobs.j[i] = {d_index_cols[i], n_rows[i], n_values[i], tiles};
        tiles += (u32)((n_rows[i] + ACC_TILE - 1) / ACC_TILE);
        total_rows += n_rows[i];
    }
    if (tiles == 0) {
        LM_HIP(hipMemsetAsync(d_acc, 0, len * 4, ctx->stream));
        return LM_OK;
    }
    LM_REQUIRE(total_rows < (1ull << 31));
    const u32 nb = (u32)((len + ACC_WIN - 1) >> ACC_WIN_LOG);
    LM_REQUIRE(nb <= 8192);  // 2 nb words of dynamic LDS (64 KiB), memory images up to 2^26 words (MAX_LOG_MEMORY_SIZE)
    // scratch (words): totals nb + 1 (the last: rows outside the image) | offsets nb+1 | cursor nb | boundary nb*16 | items total_rows
    u32* s;
    int rc = lm_scratch(ctx, 3ull * nb + 8 + (u64)nb * ACC_MAX_RUN + total_rows + 64, &s);
    if (rc) return rc;
    u32* d_totals = s;
    u32* d_offsets = d_totals + nb + 1;
    u32* d_cursor = d_offsets + nb + 1;
    u32* d_boundary = d_cursor + nb;
    u32* d_items = d_boundary + (u64)nb * ACC_MAX_RUN;
    // totals and boundary counts start from zero (split windows add into the boundary counts and into d_acc); the offsets and cursors
    // between them in the scratch block are written by k_acc_scan before anything reads them: one fill for the whole block
    LM_HIP(hipMemsetAsync(d_totals, 0, (size_t)((d_boundary + (u64)nb * ACC_MAX_RUN) - d_totals) * 4, ctx->stream));
    LM_HIP(hipMemsetAsync(d_acc, 0, len * 4, ctx->stream));
    LM_LAUNCH(ctx, (k_acc_pass<0>), dim3(tiles), dim3(256), (size_t)nb * 4, jobs, len, nb, d_totals, d_cursor, d_items);
    LM_LAUNCH(ctx, k_acc_scan, dim3(1), dim3(1024), 0, (const u32*)d_totals, nb, d_offsets, d_cursor);
    LM_LAUNCH(ctx, (k_acc_pass<1>), dim3(tiles), dim3(256), (size_t)nb * 8, jobs, len, nb, d_totals, d_cursor, d_items);
    LM_LAUNCH(ctx, k_acc_window, dim3(nb * ACC_SPLIT), dim3(256), 0, (const u32*)d_items, (const u32*)d_offsets, len, nb, d_acc, d_boundary);
    LM_LAUNCH(ctx, k_acc_finish, dim3(fin_blocks), dim3(256), 0, d_acc, len, (const u32*)d_boundary, (const u32*)(d_totals + nb),
              ctx->h_res + lm_ctx::RES_FLAG + lm_ctx::ERR_WORD);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// fill_trace_extension_op (extension_op/exec.rs:192-203): value_a[k][row] = memory[idx_a[row] + k].  A gather; rows are
// consecutive addresses inside a call (stride 1 or 5), so a wave's reads fall into a few lines.
struct ExtOpCols {
    u32* va[5];
};
__global__ __launch_bounds__(256) void k_extension_op_trace(const u32* __restrict__ memory, u64 mem_len, const u32* __restrict__ idx_a,
                                                            ExtOpCols out, u64 n_rows) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (u64)gridDim.x * 256) {
        const u64 a = from_monty(idx_a[i]);
#pragma unroll
        for (u32 k = 0; k < 5; k++) out.va[k][i] = a + k < mem_len ? memory[a + k] : 0u;
    }
}
extern "C" int lm_extension_op_trace(lm_ctx* ctx, const uint32_t* d_memory, uint64_t memory_len, const uint32_t* d_idx_a,
                                     uint32_t* const* d_va_cols, uint64_t n_rows) {
    LM_REQUIRE(ctx && d_memory && d_idx_a && d_va_cols);
    if (n_rows == 0) return LM_OK;
    ExtOpCols o;
    for (u32 k = 0; k < 5; k++) {
        LM_REQUIRE(d_va_cols[k]);
        o.va[k] = d_va_cols[k];
    }
    LM_LAUNCH(ctx, k_extension_op_trace, dim3((unsigned)std::min<u64>((n_rows + 255) / 256, 4096)), dim3(256), 0, d_memory, memory_len,
              d_idx_a, o, n_rows);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// get_execution_trace, main loop (lean_prover/src/trace_gen.rs:27-100): the 20 committed + 4 temporary columns of the
// execution table from the VM's (pc, fp) log, the instruction table and the memory image.  One cycle per lane: three uint4
// reads of the instruction row, up to three dependent memory gathers (value_a feeds DEREF's address), 24 column stores
// (coalesced: consecutive cycles are consecutive rows).
struct ExecCols {
    u32* c[24];
};
__global__ __launch_bounds__(256) void k_execution_trace(const u32* __restrict__ pcs, const u32* __restrict__ fps, u64 n_cycles,
                                                         const u32* __restrict__ bytecode, u64 bytecode_rows,
                                                         const u32* __restrict__ memory, u64 mem_len, ExecCols out) {
    const u32 TWO = add(ONE, ONE), HALF = to_monty((P + 1) / 2);
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_cycles; i += (u64)gridDim.x * 256) {
        const u32 pc = pcs[i], fp = to_monty(fps[i]);
        uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0, f2 = f0;
        if (pc < bytecode_rows) {
            const uint4* row = reinterpret_cast<const uint4*>(bytecode + (u64)pc * 16);
            f0 = row[0], f1 = row[1], f2 = row[2];
        }
        const u32 op_a = f0.x, op_b = f0.y, op_c = f0.z, flag_a = f0.w;
        const u32 flag_b = f1.x, flag_c = f1.y, flag_c_fp = f1.z, flag_ab_fp = f1.w;
        const u32 mulf = f2.x, jump = f2.y, aux = f2.z, pdata = f2.w;
        auto mem = [&](u32 addr_m) {  // memory.get(addr).flatten().unwrap_or_default()
            const u64 a = from_monty(addr_m);
            return a < mem_len ? memory[a] : 0u;
        };
        const u32 fpa = add(fp, op_a), fpb = add(fp, op_b), fpc = add(fp, op_c);
        const u32 addr_a = (flag_a == 0 && flag_ab_fp == 0) ? fpa : 0u;
        const u32 value_a = mem(addr_a);
        u32 addr_b = 0;
        if (flag_b == 0 && flag_ab_fp == 0)
            addr_b = fpb;
        else if (aux == TWO)  // DEREF: addr_B = value_A + operand_B
            addr_b = add(value_a, op_b);
        const u32 value_b = mem(addr_b);
        const u32 addr_c = (flag_c == 0 && flag_c_fp == 0) ? fpc : 0u;
        const u32 value_c = mem(addr_c);
        const u32 nu_a = add(add(mul(flag_a, op_a), mul(sub(sub(ONE, flag_a), flag_ab_fp), value_a)), mul(flag_ab_fp, fpa));
        const u32 nu_b = add(add(mul(flag_b, op_b), mul(sub(sub(ONE, flag_b), flag_ab_fp), value_b)), mul(flag_ab_fp, fpb));
        const u32 nu_c = add(add(mul(flag_c, op_c), mul(sub(sub(ONE, flag_c), flag_c_fp), value_c)), mul(flag_c_fp, fpc));
        // is_precompile as the AIR defines it (execution/air.rs:102-104): 1 - (add + mul + deref + jump)
        const u32 add_ = sub(add(aux, aux), mul(aux, aux)), deref = mul(mul(aux, sub(aux, ONE)), HALF);
        const u32 is_pre = sub(ONE, add(add(add(add_, mulf), deref), jump));
        const u32 v[24] = {to_monty(pc), fp, addr_a, addr_b, addr_c, value_a, value_b, value_c, op_a, op_b, op_c, flag_a,
                           flag_b, flag_c, flag_c_fp, flag_ab_fp, mulf, jump, aux, pdata, is_pre, nu_a, nu_b, nu_c};
#pragma unroll
        for (int c = 0; c < 24; c++) out.c[c][i] = v[c];
    }
}
extern "C" int lm_execution_table_trace(lm_ctx* ctx, const uint32_t* d_pcs, const uint32_t* d_fps, uint64_t n_cycles,
                                        const uint32_t* d_bytecode, uint64_t bytecode_rows, const uint32_t* d_memory,
                                        uint64_t memory_len, uint32_t* const* d_cols) {
    LM_REQUIRE(ctx && d_pcs && d_fps && d_bytecode && d_memory && d_cols);
    if (n_cycles == 0) return LM_OK;
    ExecCols o;
    for (int c = 0; c < 24; c++) {
        LM_REQUIRE(d_cols[c]);
        o.c[c] = d_cols[c];
    }
    LM_LAUNCH(ctx, k_execution_trace, dim3((unsigned)std::min<u64>((n_cycles + 255) / 256, 8192)), dim3(256), 0, d_pcs, d_fps, n_cycles,
              d_bytecode, bytecode_rows, d_memory, memory_len, o);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// get_execution_trace (lean_prover/src/trace_gen.rs:118-147): on rows with flag_permute = 0 the output columns that the
// AIR leaves unconstrained are overwritten with the memory words their lookup reads — outputs_right (columns 101..108) <-
// memory[res + 8 ..], and with flag_half_output = 1 also columns 97..100 <- memory[res + 4 ..].
struct PosOutCols {
    const u32 *half, *permute, *res;
    u32* out[12];  // columns 97..108
};
__global__ __launch_bounds__(256) void k_poseidon_outputs_from_memory(PosOutCols c, u64 n_rows, const u32* __restrict__ memory, u64 mem_len) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (u64)gridDim.x * 256) {
        if (c.permute[i] != 0) continue;
        const u64 base = from_monty(c.res[i]);
        if (c.half[i] == ONE) {
#pragma unroll
            for (u32 j = 0; j < 4; j++) c.out[j][i] = base + 4 + j < mem_len ? memory[base + 4 + j] : 0u;
        }
#pragma unroll
        for (u32 j = 0; j < 8; j++) c.out[4 + j][i] = base + 8 + j < mem_len ? memory[base + 8 + j] : 0u;
    }
}
extern "C" int lm_poseidon_trace_outputs_from_memory(lm_ctx* ctx, uint32_t* const* d_cols, uint64_t n_rows, const uint32_t* d_memory,
                                                     uint64_t memory_len) {
    LM_REQUIRE(ctx && d_cols && d_memory);
    if (n_rows == 0) return LM_OK;
    PosOutCols c;
    LM_REQUIRE(d_cols[2] && d_cols[3] && d_cols[8]);
    c.res = d_cols[2], c.half = d_cols[3], c.permute = d_cols[8];
    for (int j = 0; j < 12; j++) {
        LM_REQUIRE(d_cols[97 + j]);
        c.out[j] = d_cols[97 + j];
    }
    LM_LAUNCH(ctx, k_poseidon_outputs_from_memory, dim3((unsigned)std::min<u64>((n_rows + 255) / 256, 4096)), dim3(256), 0, c, n_rows,
              d_memory, memory_len);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// ---- precompile tables from the VM runner's call records ------------------------------------------------------
// Poseidon16Precompile::execute pushes the flag / index columns and the 16 input words per call (poseidon_16/mod.rs:262-286); the
// runner keeps 9 canonical words per call and the inputs are read here from the final (write-once) memory image.
struct PosCallCols {
    u32* c[27];  // columns 0..24, then 109 (index_input_left), 110 (precompile_data)
};
__global__ __launch_bounds__(256) void k_poseidon_table_from_calls(const u32* __restrict__ calls, u64 n_calls, const u32* __restrict__ memory,
                                                                   u64 mem_len, PosCallCols out) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_calls; i += (u64)gridDim.x * 256) {
        const u32* r = calls + i * 9;
        const u32 arg_a = r[0], arg_b = r[1], res = r[2], half = r[3], hard = r[4], off = r[5], lf = r[6], ls = r[7], perm = r[8];
        out.c[0][i] = ONE;
        out.c[1][i] = to_monty(arg_b);
        out.c[2][i] = to_monty(res);
        out.c[3][i] = half ? ONE : 0u;
        out.c[4][i] = hard ? ONE : 0u;
        out.c[5][i] = to_monty(off);
        out.c[6][i] = to_monty(lf);
        out.c[7][i] = to_monty(ls);
        out.c[8][i] = perm ? ONE : 0u;
#pragma unroll
        for (u32 j = 0; j < 4; j++) out.c[9 + j][i] = (u64)lf + j < mem_len ? memory[(u64)lf + j] : 0u;
#pragma unroll
        for (u32 j = 0; j < 4; j++) out.c[13 + j][i] = (u64)ls + j < mem_len ? memory[(u64)ls + j] : 0u;
#pragma unroll
        for (u32 j = 0; j < 8; j++) out.c[17 + j][i] = (u64)arg_b + j < mem_len ? memory[(u64)arg_b + j] : 0u;
        out.c[25][i] = to_monty(arg_a);
        out.c[26][i] = to_monty(1u + 2u * perm + 4u * half + 8u * hard + 16u * off);
    }
}
extern "C" int lm_poseidon_table_from_calls(lm_ctx* ctx, const uint32_t* d_calls, uint64_t n_calls, const uint32_t* d_memory, uint64_t memory_len,
                                            uint32_t* const* d_cols) {
    LM_REQUIRE(ctx && d_cols && d_memory);
    if (n_calls == 0) return LM_OK;
    LM_REQUIRE(d_calls);
    PosCallCols o;
    for (int c = 0; c < 25; c++) {
        LM_REQUIRE(d_cols[c]);
        o.c[c] = d_cols[c];
    }
    LM_REQUIRE(d_cols[109] && d_cols[110]);
    o.c[25] = d_cols[109], o.c[26] = d_cols[110];
    LM_LAUNCH(ctx, k_poseidon_table_from_calls, dim3((unsigned)std::min<u64>((n_calls + 255) / 256, 4096)), dim3(256), 0, d_calls, n_calls,
              d_memory, memory_len, o);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
// exec_multi_row's pushes (extension_op/exec.rs:149-186) from the runner's 24-word row records; VALUE_A stays with
// lm_extension_op_trace (fill_trace_extension_op).
struct ExtRowCols {
    u32* c[31];
};
__global__ __launch_bounds__(256) void k_extension_table_from_rows(const u32* __restrict__ rows, u64 n_rows, ExtRowCols out) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (u64)gridDim.x * 256) {
        const u32* r = rows + i * 24;
        const u32 is_be = r[0], start = r[1], fadd = r[2], fmul = r[3], fpe = r[4], len = r[5];
        out.c[0][i] = is_be ? ONE : 0u;
        out.c[1][i] = start ? ONE : 0u;
        out.c[2][i] = to_monty(len);
        out.c[3][i] = fadd ? ONE : 0u;
        out.c[4][i] = fmul ? ONE : 0u;
        out.c[5][i] = fpe ? ONE : 0u;
        out.c[6][i] = to_monty(r[6]);
        out.c[7][i] = to_monty(r[7]);
        out.c[13][i] = to_monty(r[8]);
#pragma unroll
        for (u32 k = 0; k < 5; k++) {
            out.c[19 + k][i] = r[9 + k];   // VB
            out.c[24 + k][i] = r[14 + k];  // VRES
            out.c[8 + k][i] = r[19 + k];   // COMP
        }
        out.c[29][i] = start ? ONE : 0u;  // activation flag
        out.c[30][i] = to_monty(4u * is_be + 8u * fadd + 16u * fmul + 32u * fpe + 64u * len);
    }
}
extern "C" int lm_extension_table_from_rows(lm_ctx* ctx, const uint32_t* d_rows, uint64_t n_rows, uint32_t* const* d_cols) {
    LM_REQUIRE(ctx && d_cols);
    if (n_rows == 0) return LM_OK;
    LM_REQUIRE(d_rows);
    ExtRowCols o;
    for (int c = 0; c < 31; c++) {
        if (c >= 14 && c < 19) {
            o.c[c] = nullptr;
            continue;
        }
        LM_REQUIRE(d_cols[c]);
        o.c[c] = d_cols[c];
    }
    LM_LAUNCH(ctx, k_extension_table_from_rows, dim3((unsigned)std::min<u64>((n_rows + 255) / 256, 4096)), dim3(256), 0, d_rows, n_rows, o);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

static int logup_build(lm_ctx* ctx, const lm_logup_section* sections, uint32_t n_sections, const uint32_t c[5], const uint32_t* alphas_eq16,
                       uint32_t n_vars, uint32_t* d_nums, uint32_t* d_dens, uint64_t* out_active_len);
extern "C" int lm_logup_build(lm_ctx* ctx, const lm_logup_section* sections, uint32_t n_sections, const uint32_t c[5],
                              const uint32_t* alphas_eq16, uint32_t n_vars, uint32_t* d_nums, uint32_t* d_dens) {
    return logup_build(ctx, sections, n_sections, c, alphas_eq16, n_vars, d_nums, d_dens, nullptr);
}
extern "C" int lm_logup_build_active(lm_ctx* ctx, const lm_logup_section* sections, uint32_t n_sections, const uint32_t c[5],
                                     const uint32_t* alphas_eq16, uint32_t n_vars, uint32_t* d_nums, uint32_t* d_dens, uint64_t* out_active_len) {
    LM_REQUIRE(out_active_len);
    return logup_build(ctx, sections, n_sections, c, alphas_eq16, n_vars, d_nums, d_dens, out_active_len);
}
// out_active_len != NULL: only [0, end of the last section rounded up to 8) is written (neutral pairs in the holes); the tail is
// left untouched for lm_gkr_build_active
static int logup_build(lm_ctx* ctx, const lm_logup_section* sections, uint32_t n_sections, const uint32_t c[5], const uint32_t* alphas_eq16,
                       uint32_t n_vars, uint32_t* d_nums, uint32_t* d_dens, uint64_t* out_active_len) {
    LM_REQUIRE(ctx && sections && n_sections && c && alphas_eq16 && d_nums && d_dens && n_vars <= 30);
    const u64 plane = 1ull << n_vars;
    u64 active = 0;
    for (u32 k = 0; k < n_sections; k++) active = std::max<u64>(active, sections[k].out_offset + (1ull << sections[k].log_len));
    const u64 fill_len = out_active_len ? std::min<u64>((active + 7) & ~7ull, plane) : plane;
    if (out_active_len) *out_active_len = active;
    std::vector<LogupSec> hs(n_sections);
    u64 chunks = 0;
    EF a15;
    memcpy(a15.v, alphas_eq16 + 75, 20);
    for (u32 k = 0; k < n_sections; k++) {
        const lm_logup_section& in = sections[k];
        LM_REQUIRE(in.n_data <= LM_LOGUP_MAX_DATA && in.num_mode <= 3 && in.log_len <= n_vars);
        LM_REQUIRE(in.out_offset + (1ull << in.log_len) <= plane);
        LM_REQUIRE(in.num_mode < 2 || in.d_num_col);
        LogupSec& s = hs[k];
        s.out_offset = in.out_offset;
        s.len = 1ull << in.log_len;
        s.chunk_begin = chunks;
        chunks += ((1ull << in.log_len) + LG_CHUNK - 1) / LG_CHUNK;
        s.num_col = in.d_num_col;
        s.log_len = in.log_len;
        s.num_mode = in.num_mode;
        s.n_data = in.n_data;
        s.neg_den = in.den_sign < 0;
        s.contrib = ef_mul_base(a15, to_monty(in.domsep));
        for (u32 j = 0; j < LM_LOGUP_MAX_DATA; j++) {
            s.data[j] = j < in.n_data ? in.d_data[j] : nullptr;
            s.stride[j] = j < in.n_data ? in.stride[j] : 0;
            s.add_m[j] = j < in.n_data ? to_monty(in.add[j]) : 0;
        }
    }
    // the neutral pair (0, 1) everywhere outside the sections, written by the same launch: the holes between the sections (sorted by
    // offset; sections are disjoint) and the tail up to fill_len become sections of their own — every word is written exactly once
    // (round 2 wrote the whole domain with a separate kernel first: 472 MB that the fill largely rewrote)
    {
        std::vector<std::pair<u64, u64>> ranges(n_sections);
        for (u32 k = 0; k < n_sections; k++) ranges[k] = {sections[k].out_offset, sections[k].out_offset + (1ull << sections[k].log_len)};
        std::sort(ranges.begin(), ranges.end());
        u64 at = 0;
        auto hole = [&](u64 from, u64 to) {
            if (to <= from) return;
            LogupSec h;
            memset(&h, 0, sizeof h);
            h.out_offset = from;
            h.len = to - from;
            h.num_mode = 4;
            h.chunk_begin = chunks;
            chunks += (h.len + LG_CHUNK - 1) / LG_CHUNK;
            hs.push_back(h);
        };
        for (const auto& r : ranges) {
            LM_REQUIRE(r.first >= at);  // disjoint
            hole(at, r.first);
            at = r.second;
        }
        hole(at, fill_len);
    }
    n_sections = (uint32_t)hs.size();
    LM_REQUIRE(chunks < (1ull << 31));
    const u64 w_secs = (sizeof(LogupSec) * n_sections + 3) / 4;
    u32* s;
    int rc = lm_scratch(ctx, w_secs + 16 + 80, &s);
    if (rc) return rc;
    u32* d_al = s + ((w_secs + 15) & ~15ull);
    if ((rc = lm_stage_upload(ctx, s, hs.data(), sizeof(LogupSec) * n_sections)) || (rc = lm_stage_upload(ctx, d_al, alphas_eq16, 320))) return rc;
    EF cc;
    memcpy(cc.v, c, 20);
    LM_LAUNCH(ctx, k_logup_fill, dim3((unsigned)chunks), dim3(256), 0, (const LogupSec*)s, n_sections, (const EF*)d_al, cc,
              plane, d_nums, d_dens, to_monty(to_monty(1)));
    LM_HIP(hipGetLastError());
    return LM_OK;
}
