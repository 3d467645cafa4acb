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
    u32 log_len, num_mode, n_data, neg_den;
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
    const u64 len = 1ull << s.log_len;
    const u64 base = (b - s.chunk_begin) * LG_CHUNK;
    for (u32 u = 0; u < LG_CHUNK / 256; u++) {
        const u64 i = base + u * 256 + threadIdx.x;
        if (i >= len) break;
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
// neutral pair (0, 1) everywhere (sections overwrite their ranges afterwards)
__global__ __launch_bounds__(256) void k_logup_neutral(u64 plane, u32* __restrict__ nums, u32* __restrict__ dens) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < plane; i += (u64)gridDim.x * 256) {
        nums[i] = 0;
        dens[i] = ONE;
#pragma unroll
        for (int k = 1; k < 5; k++) dens[(u64)k * plane + i] = 0;
    }
}

// ---- access counters (crates/lean_prover/src/prove_execution.rs:90-110) ---------------------------------------------
// acc[index(row) + j] += 1 for every row of an index column and j < n_values: a histogram.  The reference loops
// sequentially ("TODO parallelize"); here: integer atomics, with the whole-wave-same-address case (padding rows, which all
// point at one address) collapsed into one atomic, then one pass converts counts to field elements in place.
struct AccessJob {
    const u32* index_col;  // Montgomery words, canonical value = address
    u64 n_rows;
    u32 n_values, pad;
};
// (A difference-array variant — +1 at the address, -1 after the last word, prefix sum — was measured SLOWER on MI355X:
// device-scope atomics are executed memory-side, and n_values atomics on one cache line cost less than two on distant lines.)
// n consecutive counters += v each: 64-bit atomics cover two adjacent 32-bit counters at once (a counter never carries
// into its neighbour: counts stay far below 2^32)
__device__ __forceinline__ void add_run(u32* __restrict__ cnt, u32 a, u32 n, u32 v) {
    u32 j = 0;
    if ((a & 1) && n) {
        atomicAdd(&cnt[a], v);
        j = 1;
    }
    const unsigned long long vv = ((unsigned long long)v << 32) | v;
    for (; j + 2 <= n; j += 2) atomicAdd(reinterpret_cast<unsigned long long*>(&cnt[a + j]), vv);
    if (j < n) atomicAdd(&cnt[a + j], v);
}
__global__ __launch_bounds__(256) void k_access_count(const AccessJob* __restrict__ jobs, u32* __restrict__ cnt, u64 len) {
    const AccessJob jb = jobs[blockIdx.y];
    for (u64 r = (u64)blockIdx.x * 256 + threadIdx.x; r < jb.n_rows; r += (u64)gridDim.x * 256) {
        const u32 a = from_monty(jb.index_col[r]);
        const bool ok = (u64)a + jb.n_values <= len;
        const u32 first = __builtin_amdgcn_readfirstlane(a);
        const u64 same = __ballot(a == first);
        if (same == __ballot(1)) {  // every active lane hits the same address: one lane adds for the wave
            if (ok && (threadIdx.x & 63) == (u32)__builtin_ctzll(same)) add_run(cnt, a, jb.n_values, (u32)__popcll(same));
        } else if (ok) {
            add_run(cnt, a, jb.n_values, 1u);
        }
    }
}
__global__ __launch_bounds__(256) void k_counts_to_field(u32* __restrict__ v, u64 n) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) v[i] = to_monty(v[i]);
}

extern "C" int lm_access_counts(lm_ctx* ctx, uint32_t* d_acc, uint64_t len, uint32_t n_jobs, const uint32_t* const* d_index_cols,
                                const uint64_t* n_rows, const uint32_t* n_values) {
    LM_REQUIRE(ctx && d_acc && len > 0 && len < (1ull << 31));
    LM_REQUIRE(n_jobs == 0 || (d_index_cols && n_rows && n_values));
    LM_HIP(hipMemsetAsync(d_acc, 0, len * 4, ctx->stream));
    if (n_jobs) {
        std::vector<AccessJob> jobs(n_jobs);
        u64 max_rows = 1;
        for (u32 i = 0; i < n_jobs; i++) {
            LM_REQUIRE(d_index_cols[i] && n_values[i] >= 1 && n_values[i] <= len);
            jobs[i] = {d_index_cols[i], n_rows[i], n_values[i], 0};
            max_rows = std::max(max_rows, n_rows[i]);
        }
        u32* s;
        int rc = lm_scratch(ctx, (sizeof(AccessJob) * n_jobs + 3) / 4 + 16, &s);
        if (rc) return rc;
        LM_HIP(hipMemcpyAsync(s, jobs.data(), sizeof(AccessJob) * n_jobs, hipMemcpyHostToDevice, ctx->stream));
        LM_HIP(hipStreamSynchronize(ctx->stream));  // `jobs` is a local
        const u32 blocks = (u32)std::min<u64>((max_rows + 255) / 256, 4096);
        LM_LAUNCH(ctx, k_access_count, dim3(blocks, n_jobs), dim3(256), 0, (const AccessJob*)s, d_acc, len);
    }
    LM_LAUNCH(ctx, k_counts_to_field, dim3((unsigned)std::min<u64>((len + 255) / 256, 4096)), dim3(256), 0, d_acc, len);
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

extern "C" int lm_logup_build(lm_ctx* ctx, const lm_logup_section* sections, uint32_t n_sections, const uint32_t c[5],
                              const uint32_t* alphas_eq16, uint32_t n_vars, uint32_t* d_nums, uint32_t* d_dens) {
    LM_REQUIRE(ctx && sections && n_sections && c && alphas_eq16 && d_nums && d_dens && n_vars <= 30);
    const u64 plane = 1ull << n_vars;
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
    LM_REQUIRE(chunks < (1ull << 31));
    const u64 w_secs = (sizeof(LogupSec) * n_sections + 3) / 4;
    u32* s;
    int rc = lm_scratch(ctx, w_secs + 16 + 80, &s);
    if (rc) return rc;
    u32* d_al = s + ((w_secs + 15) & ~15ull);
    LM_HIP(hipMemcpyAsync(s, hs.data(), sizeof(LogupSec) * n_sections, hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(hipMemcpyAsync(d_al, alphas_eq16, 320, hipMemcpyHostToDevice, ctx->stream));
    LM_HIP(hipStreamSynchronize(ctx->stream));
    EF cc;
    memcpy(cc.v, c, 20);
    LM_LAUNCH(ctx, k_logup_neutral, dim3((unsigned)std::min<u64>((plane + 255) / 256, 4096)), dim3(256), 0, plane, d_nums, d_dens);
    LM_LAUNCH(ctx, k_logup_fill, dim3((unsigned)chunks), dim3(256), 0, (const LogupSec*)s, n_sections, (const EF*)d_al, cc,
              plane, d_nums, d_dens, to_monty(to_monty(1)));
    LM_HIP(hipGetLastError());
    return LM_OK;
}
