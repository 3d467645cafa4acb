// AIR sumcheck sessions on gfx950 (reference: crates/sub_protocols/src/air_sumcheck.rs — AirSumcheckSession /
// OuterSumcheckSession :34-292, compute_raw_poly_impl :560-634).
//
// A session holds the columns of one table (column-major, natural row order; base words before the first fold, SoA EF
// afterwards) and serves one sumcheck round at a time, LSB first:
//   lm_air_round : raw[z] = sum_pairs eq_prefix(pair) * sum_k alpha^k C_k(lo + z (hi - lo)),   z in {0, 2, .., degree}
//   lm_air_bind  : fold every column with the challenge (fold_multilinear_at_bit :117-159, bit 0 in natural order)
// Work decomposition: grid.y = evaluation point z, one row pair per lane; the constraint evaluators (air_tables.h) keep
// only the live part of a row in registers.  "Shift" columns (next-row view of the first n_shift columns,
// compute_shifted_columns :683-694) are read in place from the base columns in round 0 and materialised by the first fold.
#include <algorithm>
#include "air_tables.h"
#include "lm_common.h"
#include "lm_eqsplit.h"

using namespace kb;

struct lm_air {
    int table = 0;
    u32 log_rows = 0, round = 0;
    u32 n_cols = 0, n_shift = 0, deg = 0;
    u32 n_virt = 0;                     // virtual columns appended after the committed ones (Poseidon: air::POS_N_VIRT)
    u32* d_virt = nullptr;              // their base-field values, n_virt x 2^log_rows
    const u32** d_base_cols = nullptr;  // device array of n_cols + n_virt device pointers (caller's base columns, then virtual)
    u32* ef[2] = {nullptr, nullptr};    // ping-pong: (n_cols + n_virt + n_shift) columns x 5 planes
    int cur = -1;                       // -1: base columns; -2: base columns seen through the first challenge (FoldCols); 0 / 1: ef[cur]
    bool lazy_ok = false;               // the first fold may stay unmaterialised (lm_air_new)
    EF r1;                              // cur == -2: the first challenge
    air::Extra* d_extra = nullptr;
    air::Extra h_extra;                  // host copies outlive the asynchronous uploads (no synchronisation in lm_air_new)
    std::vector<const u32*> h_cols;
    PrefixEqTables eqt;
    u32* d_partial = nullptr;            // per-block partial sums of a round (own buffer: sessions of one batch run back to back)
    u32 res_off = 0;                     // this session's slice of the pinned result buffer
    u32 pending_seq = 0;                 // sequence number of the launched, not yet collected round (0 = none)
    // The session runs on its own stream (ctx->aux_stream[aux]) from the end of lm_air_new on: the sessions of a batched
    // round are independent chains (round kernel -> host -> fold kernel -> ...), and in the last ~12 rounds each of them is
    // one latency-bound workgroup — side by side they cost the longest of the three instead of the sum.
    u64 n_active = 0;                    // rows >= n_active are identical padding rows (lm_air_set_active_rows; default: all rows active)
    std::vector<EF> h_point;             // the eq point (host copy): tail sums of eq weights for the padded pairs
    std::vector<u32> h_final;            // lm_air_final_evals_begin on a session without a stream of its own: the values, until _end
    int aux = 0;
    hipStream_t stream = nullptr;
    u32* d_sync = nullptr;               // this session's "writers done" counter (two words, zero between kernels)
};
static constexpr u32 AIR_MAX_BLOCKS = 2048;

// column value at the evaluation point z of a row pair: lo + z (hi - lo)
__device__ __forceinline__ u32 lerp(u32 lo, u32 hi, u32 zm) { return add(lo, mul(sub(hi, lo), zm)); }
__device__ __forceinline__ EF lerp(const EF& lo, const EF& hi, u32 zm) { return ef_add(lo, ef_mul_base(ef_sub(hi, lo), zm)); }

struct BaseCols {
    const u32* const* cols;
    u64 n_rows;
    u32 n_flat;
    // column c < n_flat: flat; c >= n_flat: shift view of column c - n_flat
    // RELOAD: the pointer table is re-read next to each use.  Hoisted out of the row loop, the 42 column pointers of the
    // ExtensionOp evaluation are spilled from SGPRs to VGPR lanes (1.5 k v_readlane of 4.9 k instructions, see
    // ExtCols::at); the execution table (22 columns) and the per-segment Poseidon kernels touch few columns each and are
    // better off with the hoisted loads (execution base round 241 -> 208 us without the reload).
    __device__ __forceinline__ uint2 raw(u32 c, u64 j) const { return *reinterpret_cast<const uint2*>(cols[c] + 2 * j); }  // flat column
    template <bool RELOAD>
    __device__ __forceinline__ u32 at(u32 c, u64 j, u32 zm) const {
        const u32* const* cp = cols;
        if constexpr (RELOAD) asm volatile("" : "+s"(cp));
        if (c < n_flat) {
            uint2 v = *reinterpret_cast<const uint2*>(cp[c] + 2 * j);
            return lerp(v.x, v.y, zm);
        }
        const u32* p = cp[c - n_flat];
        const u64 i1 = 2 * j + 1, i2 = (2 * j + 2 < n_rows) ? 2 * j + 2 : n_rows - 1;
        return lerp(p[i1], p[i2], zm);
    }
};
struct ExtCols {
    const u32* buf;  // column c plane k at buf + (c * 5 + k) * n_rows
    u64 n_rows;
    template <bool RELOAD>
    __device__ __forceinline__ EF at(u32 c, u64 j, u32 zm) const {
        EF lo, hi;
        // The plane stride is made opaque at every access: as a loop invariant, the compiler hoists all (columns x 5) plane
        // addresses out of the row loop into SGPR pairs — ~1100 of them for the Poseidon table — and spills them to VGPR
        // lanes (3 k v_readlane + 2.7 k v_writelane per evaluation).  Recomputed next to the loads they are 3 scalar
        // instructions per plane on the otherwise idle scalar unit.
        u64 nr = n_rows;
        asm volatile("" : "+s"(nr));
        const u32* p = buf + (u64)c * 5 * nr + 2 * j;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint2 v = *reinterpret_cast<const uint2*>(p + k * nr);
            lo.v[k] = v.x;
            hi.v[k] = v.y;
        }
        return lerp(lo, hi, zm);
    }
};

// The table after its FIRST fold, not materialised: row i of the folded table is a_{2i} + r (a_{2i+1} - a_{2i}) with base-field a's, so
// a column's value at the evaluation point z of the folded pair (2j, 2j+1) is  [a0 + z (a2 - a0)] + r [(a1 - a0) + z ((a3 - a2) - (a1 -
// a0))]  from the FOUR base words 4j .. 4j+3: two base products by z and one base-by-extension product.  Round 1 reads 16 bytes per
// column and pair instead of 40 from a folded copy, and the copy — the largest one of the session, 5 x the committed columns — is never
// written: the second challenge folds the base columns twice in one pass (k_air_fold2_base).  Same field values as ExtCols over the
// output of k_air_fold_base.
template <bool SHIFT>  // SHIFT = false: the table has no shift columns (the Poseidon table): no second path behind every access
struct FoldCols {
    const u32* const* cols;
    u64 n_rows;  // of the BASE table
    u32 n_flat;
    EF r;
    template <bool RELOAD>
    __device__ __forceinline__ EF at(u32 c, u64 j, u32 zm) const {
        // (the pointer table is re-read next to every use, whatever RELOAD says: hoisted out of the row loop, the 160 column pointers of
        // the Poseidon table are 320 SGPRs, spilled to VGPR lanes — BaseCols::at, ExtCols::at)
        const u32* const* cp = cols;
        asm volatile("" : "+s"(cp));
        u32 a0, a1, a2, a3;
        if (!SHIFT || c < n_flat) {
            const uint4 v = *reinterpret_cast<const uint4*>(cp[c] + 4 * j);
            a0 = v.x, a1 = v.y, a2 = v.z, a3 = v.w;
        } else {  // shift view: row i of the view is row min(i + 1, n_rows - 1) of the column
            const u32* p = cp[c - n_flat];
            const u64 i = 4 * j;
            a0 = p[i + 1], a1 = p[i + 2], a2 = p[i + 3], a3 = p[(i + 4 < n_rows) ? i + 4 : n_rows - 1];
        }
        const u32 d0 = sub(a1, a0), d1 = sub(a3, a2);
        EF o = ef_mul_base(r, lerp(d0, d1, zm));
        o.v[0] = add(o.v[0], lerp(a0, a2, zm));
        return o;
    }
};

// FoldCols of the Poseidon table with the two base coefficients of a value on offer (air_tables.h: eval_poseidon16_segment_affine)
struct FoldColsAff : FoldCols<false> {
    air::TPowers tp;  // r, r^2, r^3
    __device__ __forceinline__ air::Ab ab(u32 c, u64 j, u32 zm) const {
        const u32* const* cp = cols;
        asm volatile("" : "+s"(cp));
        const uint4 v = *reinterpret_cast<const uint4*>(cp[c] + 4 * j);
        air::Ab o;
        o.a = lerp(v.x, v.z, zm);
        o.b = lerp(sub(v.y, v.x), sub(v.w, v.z), zm);
        return o;
    }
};

// The execution table after its first fold, not materialised, as the two base coefficients of every value (air_tables.h:
// eval_execution_affine): the 20 flat columns from one 16-byte load, the two shift views from rows 4j+1 .. 4j+4.
struct FoldColsExec {
    const u32* const* cols;
    u64 n_rows;  // of the BASE table
    u32 n_flat;
    air::ExecAffine ex;
    __device__ __forceinline__ air::Ab ab(u32 c, u64 j, u32 zm) const {
        u32 a0, a1, a2, a3;
        if (c < n_flat) {
            const uint4 v = *reinterpret_cast<const uint4*>(cols[c] + 4 * j);
            a0 = v.x, a1 = v.y, a2 = v.z, a3 = v.w;
        } else {  // shift view: row i of the view is row min(i + 1, n_rows - 1) of the column
            const u32* p = cols[c - n_flat];
            const u64 i = 4 * j;
            a0 = p[i + 1], a1 = p[i + 2], a2 = p[i + 3], a3 = p[(i + 4 < n_rows) ? i + 4 : n_rows - 1];
        }
        air::Ab o;
        o.a = lerp(a0, a2, zm);
        o.b = lerp(sub(a1, a0), sub(a3, a2), zm);
        return o;
    }
};
template <class C>
struct cols_exec_affine {
    static constexpr bool value = false;
};
template <>
struct cols_exec_affine<FoldColsExec> {
    static constexpr bool value = true;
};

template <class C>
struct cols_lazy_fold {
    static constexpr bool value = false;
};
template <bool S>
struct cols_lazy_fold<FoldCols<S>> {
    static constexpr bool value = true;
};
template <>
struct cols_lazy_fold<FoldColsAff> {
    static constexpr bool value = true;
};
template <class C>
struct cols_affine {
    static constexpr bool value = false;
};
template <>
struct cols_affine<FoldColsAff> {
    static constexpr bool value = true;
};

#ifndef AIR_EXT_WAVES
#define AIR_EXT_WAVES 1
#endif
#ifndef AIR_BASE_SEG_WAVES
#define AIR_BASE_SEG_WAVES 4
#endif
static constexpr u64 AIR_SPLIT_LAUNCH_PAIRS = 1ull << 13;
static constexpr u32 AIR_POS_POINTS = 10;                      // evaluation points of the Poseidon table (degree 10)
static constexpr u32 AIR_POS_SLOTS = 4 * AIR_POS_POINTS + 4;   // rows of its partial-sum matrix, see k_air_round

struct AirLagrange {
    u32 c[AIR_POS_POINTS][4];
};
// Small rounds finish inside k_air_round: the workgroup that takes the last ticket adds up the per-workgroup partial sums
// (at most AIR_INLINE_MAX_BLOCKS x slots of them), applies the degree-3 extrapolation of the Poseidon partial-round segment
// and publishes — no k_air_reduce launch.  out == nullptr: the partials are left for k_air_reduce (large rounds, thousands
// of partials per point).
struct AirFinish {
    u32* out;            // this session's slice of the pinned result buffer, or nullptr
    u32* flag_word;      // the sequence flag of this session's stream
    u32* done_counter;
    u32 seq, deg, n_main, n_low;
    u64 low_offset;
    AirLagrange lag;
    // Active prefix (air_sumcheck.rs:194-200,236-240): the round sums only the pairs that contain an active row; every pair
    // behind them consists of identical padding rows and evaluates to the same value, so ONE of them (pad_pair) is evaluated
    // and weighted with the sum of the eq weights of all of them (pad_w, computed on the host).  pad_on = 0: full domain.
    u32 pad_on;
    u64 pad_pair;
    EF pad_w;
};
static constexpr u32 AIR_INLINE_MAX_BLOCKS = 8;
__device__ __forceinline__ void air_finish_inline(const u32* __restrict__ partial, u32 blocks_x, const AirFinish& fin) {
    __shared__ u32 last;
    __syncthreads();
    if (threadIdx.x == 0) last = lm_ticket(fin.done_counter) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x < fin.deg * 5) {
        const u32 zi = threadIdx.x / 5, k = threadIdx.x % 5;
        u32 v = 0;
        for (u32 b = 0; b < fin.n_main; b++) v = add(v, lm_load_agent(partial + ((u64)zi * fin.n_main + b) * 5 + k));
        if (fin.n_low) {
            for (u32 t = 0; t < 4; t++) {
                u32 w = 0;
                for (u32 b = 0; b < fin.n_low; b++) w = add(w, lm_load_agent(partial + (fin.low_offset + (u64)t * fin.n_low + b) * 5 + k));
                v = add(v, mul(w, fin.lag.c[zi][t]));
            }
        }
        lm_store_system(fin.out + threadIdx.x, v);
        lm_wait_stores();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lm_store_agent(fin.done_counter, 0);
        lm_publish_flag_word(fin.flag_word, fin.seq);
    }
}

template <int TABLE, class T, class Cols, int SEG>
__device__ __forceinline__ EF eval_table(const Cols& cols, u64 j, u32 zm, u32 seg, const air::Extra& x) {
    if constexpr (TABLE == air::T_POSEIDON16 && sizeof(T) == sizeof(u32) && SEG >= 0) {
        // Base-field round, one segment per launch: a wave evaluates ONE row pair (3-4 k instructions), so its life is the
        // chain of column loads, not the arithmetic.  All columns of the segment are requested up front (<= 42 loads in
        // flight per lane, 84 VGPRs) instead of lazily next to their use (~10 dependent waits).  (Segments 0, 1, 3 read
        // their output block as the 5 planes of one virtual column, air_tables.h: POS_VIRT_O.)
        u32 val[air::POS_VIRT_O + 15];
        auto fetch = [&](auto FIRST, auto COUNT) {
            constexpr int first = decltype(FIRST)::value, count = decltype(COUNT)::value;
            uint2 raw[count];
            static_for<0, count>([&](auto I) { raw[decltype(I)::value] = cols.raw(first + decltype(I)::value, j); });
            static_for<0, count>([&](auto I) {
                constexpr int i = decltype(I)::value;
                val[first + i] = lerp(raw[i].x, raw[i].y, zm);
            });
        };
        using kb::IntC;
        if constexpr (SEG == 0) {
            fetch(IntC<0>{}, IntC<25>{});  // flags, inputs
            fetch(IntC<air::POS_VIRT_O>{}, IntC<5>{});
        } else if constexpr (SEG == 1) {
            fetch(IntC<25>{}, IntC<16>{});
            fetch(IntC<air::POS_VIRT_O + 5>{}, IntC<5>{});
        } else if constexpr (SEG == 2) {
            fetch(IntC<57>{}, IntC<20>{});
            fetch(IntC<air::POS_VIRT_Y>{}, IntC<20>{});
        } else if constexpr (SEG == 3) {
            fetch(IntC<air::POS_VIRT_E>{}, IntC<16>{});
            fetch(IntC<air::POS_VIRT_O + 10>{}, IntC<5>{});
        } else {
            fetch(IntC<77>{}, IntC<32>{});
            fetch(IntC<3>{}, IntC<1>{});
            fetch(IntC<8>{}, IntC<9>{});  // flag_permute, inputs 0..7
        }
        auto col = [&](int c) { return val[c]; };
        return air::eval_poseidon16_segment<T, SEG>(col, x);
    } else if constexpr (TABLE == air::T_POSEIDON16 && cols_affine<Cols>::value) {
        // round 1 through the first challenge with the base coefficients of every value at hand (segment uniform per workgroup)
        auto col = [&](int c) { return cols.template at<false>((u32)c, j, zm); };
        auto ab = [&](int c) { return cols.ab((u32)c, j, zm); };
        if (seg == 0) return air::eval_poseidon16_segment_affine<0>(col, ab, cols.tp, x);
        if (seg == 1) return air::eval_poseidon16_segment_affine<1>(col, ab, cols.tp, x);
        if (seg == 2) return air::eval_poseidon16_segment_affine<2>(col, ab, cols.tp, x);
        if (seg == 3) return air::eval_poseidon16_segment_affine<3>(col, ab, cols.tp, x);
        return air::eval_poseidon16_segment_affine<4>(col, ab, cols.tp, x);
    } else if constexpr (TABLE == air::T_POSEIDON16) {
        auto col = [&](int c) { return cols.template at<false>((u32)c, j, zm); };
        if constexpr (SEG >= 0) return air::eval_poseidon16_segment<T, SEG>(col, x);
        // segment is uniform per workgroup (blockIdx.y), so this switch does not diverge
        if (seg == 0) return air::eval_poseidon16_segment<T, 0>(col, x);
        if (seg == 1) return air::eval_poseidon16_segment<T, 1>(col, x);
        if (seg == 2) return air::eval_poseidon16_segment<T, 2>(col, x);
        if (seg == 3) return air::eval_poseidon16_segment<T, 3>(col, x);
        return air::eval_poseidon16_segment<T, 4>(col, x);
    } else if constexpr (TABLE == air::T_EXECUTION && cols_exec_affine<Cols>::value) {
        air::Ab flat[20], shift[2];
        static_for<0, 20>([&](auto C) { flat[decltype(C)::value] = cols.ab(decltype(C)::value, j, zm); });
        static_for<0, 2>([&](auto C) { shift[decltype(C)::value] = cols.ab(20 + decltype(C)::value, j, zm); });
        (void)seg;
        return air::eval_execution_affine(flat, shift, x, cols.ex);
    } else {
        constexpr int NF = air::n_columns(TABLE), NS = air::n_shift(TABLE);
        // (one copy of the loads per instantiation of the evaluator: a part of the ExtensionOp constraints leaves the columns it
        // does not touch unread)
        auto run = [&](auto PARTC) {
            T flat[NF], shift[NS];
#pragma unroll
            for (int c = 0; c < NF; c++) flat[c] = cols.template at<(TABLE != air::T_EXECUTION)>(c, j, zm);
#pragma unroll
            for (int c = 0; c < NS; c++) shift[c] = cols.template at<(TABLE != air::T_EXECUTION)>(NF + c, j, zm);
            if constexpr (TABLE == air::T_EXECUTION)
                return air::eval_execution<T>(flat, shift, x);
            else
                return air::eval_extension_op<T, decltype(PARTC)::value>(flat, shift, x);
        };
        if constexpr (TABLE == air::T_EXTENSION_OP && SEG == -2) {  // the part is uniform per workgroup: no divergence
            if (seg == 0) return run(kb::IntC<0>{});
            if (seg == 1) return run(kb::IntC<1>{});
            if (seg == 2) return run(kb::IntC<2>{});
            return run(kb::IntC<3>{});
        } else {
            (void)seg;
            return run(kb::IntC<-1>{});
        }
    }
}

// 1-D grid of blocks_x * ny workgroups; ny = slots of this launch; partial[(slot * blocks_x + tile) * 5 + k].
// Workgroup -> (tile, y) is XCD-aware: the ny workgroups that evaluate the same row pairs (at different points / segments)
// read the same column words, so they get ids that are equal mod 8 (same XCD, same L2 — workgroups are dealt round-robin
// over the 8 XCDs) and adjacent in dispatch order.  With y as the slow grid dimension every point re-read the columns
// from HBM (rocprofv3 FETCH_SIZE: 10x the algorithmic bytes).
// SEG < 0: all segments in one launch (small rounds: one launch, 5x shorter dependent chains);
// SEG >= 0: one launch per segment (large rounds: each segment gets its own register budget).
// (the combined extension-field Poseidon kernel fits 3 waves per SIMD; asking for it keeps the allocator from drifting to 2)
template <int TABLE, class T, class Cols, int SEG>
__global__ __launch_bounds__(256, (TABLE == air::T_POSEIDON16 && sizeof(T) == sizeof(EF) && SEG < 0) ? (cols_lazy_fold<Cols>::value ? 2 : 3) : ((TABLE == air::T_POSEIDON16 && sizeof(T) == sizeof(u32) && SEG >= 0) ? AIR_BASE_SEG_WAVES : (TABLE == air::T_EXECUTION ? 2 : (sizeof(T) == sizeof(u32) ? AIR_EXT_WAVES : 1)))) void k_air_round(Cols cols, u64 n_pairs, const air::Extra* __restrict__ extra, EqSplit eq,
                                                   u32* __restrict__ partial, u32 blocks_x, u32 ny, AirFinish fin) {
    __shared__ u32 lds[20];
    u32 tile, y;
    if ((blocks_x & 7) == 0) {
        const u32 group = blockIdx.x / (8 * ny), rem = blockIdx.x % (8 * ny);
        y = rem >> 3;
        tile = group * 8 + (rem & 7);
    } else {
        y = blockIdx.x / blocks_x;
        tile = blockIdx.x % blocks_x;
    }
    // slot = row of the partial-sum matrix.  Poseidon: slots 0..39 = (point zi, segment in {0,1,3,4}), slots 40..43 = the
    // partial-round segment 2 at z = 0,1,2,3 — its constraints have degree 3 in the row variable, so four points determine
    // it and k_air_reduce extrapolates the SUMS to the other points (4 evaluations instead of 10).
    u32 seg, z, slot;
    if constexpr (TABLE == air::T_POSEIDON16) {
        if constexpr (SEG < 0) {
            slot = y;
            if (y < 4 * AIR_POS_POINTS) {
                const u32 s4 = y & 3, zi = y >> 2;
                seg = s4 < 2 ? s4 : s4 + 1;
                z = zi == 0 ? 0 : zi + 1;
            } else {
                seg = 2;
                z = y - 4 * AIR_POS_POINTS;
            }
        } else if constexpr (SEG == 2) {
            seg = 2;
            z = y;
            slot = 4 * AIR_POS_POINTS + y;
        } else {
            seg = SEG;
            z = y == 0 ? 0 : y + 1;
            slot = y * 4 + (SEG < 2 ? SEG : SEG - 1);
        }
    } else if constexpr (TABLE == air::T_EXTENSION_OP && SEG == -2) {
        // small rounds: (point, part of the constraint list) per workgroup, air::eval_extension_op<T, PART>
        seg = y % air::EXT_PARTS;
        const u32 zi = y / air::EXT_PARTS;
        z = zi == 0 ? 0 : zi + 1;
        slot = y;
    } else {
        seg = 0;
        z = y == 0 ? 0 : y + 1;  // 0, 2, 3, ..., degree
        slot = y;
    }
    const u32 zm = to_monty(z);
    EF acc = ef_zero();
    // (one call site of the constraint evaluation.  The padding pair takes the grid-stride slot right behind the last active
    // pair, j == n_pairs: the thread that owns it has no more iterations than any other, so small rounds — one evaluation
    // per thread, latency bound — do not get twice as long)
    for (u64 j = (u64)tile * 256 + threadIdx.x;; j += (u64)blocks_x * 256) {
        bool is_pad = false;
        u64 je = j;
        if (j >= n_pairs) {
            if (!(fin.pad_on && j == n_pairs)) break;
            is_pad = true;
            je = fin.pad_pair;
        }
        const EF v = eval_table<TABLE, T, Cols, SEG>(cols, je, zm, seg, *extra);
        acc = ef_add(acc, ef_mul(v, is_pad ? fin.pad_w : eq_split_at(eq, je)));
        if (is_pad) break;
    }
    u32 v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = wave_sum_u32(acc.v[k]);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) lds[wave * 5 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        u32 s = 0;
        for (u32 w = 0; w < 4; w++) s = add(s, lds[w * 5 + threadIdx.x]);
        if (fin.out) {
            lm_store_agent(partial + ((u64)slot * blocks_x + tile) * 5 + threadIdx.x, s);
            lm_wait_stores();
        } else {
            partial[((u64)slot * blocks_x + tile) * 5 + threadIdx.x] = s;
        }
    }
    if (fin.out) air_finish_inline(partial, blocks_x, fin);
}
// ---- lane-cooperative evaluation of the Poseidon table in the small extension-field rounds --------------------------------------
// In the last rounds a table has a few hundred row pairs at most and k_air_round's one-lane-per-(pair, point, segment) evaluation
// is a single dependent chain of ~9 k instructions (~25 us however few pairs are left).  Here SIXTEEN lanes evaluate one (pair,
// point, segment): lane l owns state word l — the S-box layers run in parallel, the circulant MDS is 15 lane rotations inside the
// 16-lane row (as poseidon16_coop.h does for hashing), the segment's constraints are spread over the lanes as alpha^k * A * B
// terms with per-lane operands (data selection, no divergent paths), and a row sum collects them.  The chain is 5-9 extension
// multiplications + 1-2 MDS instead of ~80 + 2.  Exactly the sums of k_air_round (field arithmetic, distributivity):
//   segments 0 / 1 / 3: sum_l beta[l] * cube(MDS(cube(in + rc))_l + rc') - V        (out_block; lane K < 5 subtracts X^K * V_K)
//   segment 0 also: alpha^0 * bus + alpha^1..7 * (flag constraints): lanes 1..7 one product each, lanes 8..11 the four products of
//                   the bus fingerprint, lane 0 finishes the bus value
//   segment 4: full rounds 6 and 7, then lane i < 8: alpha^(76+3i) gate_i (s_i + in_i - out_i) + alpha^(77+3i) flag_permute (s_i - out_i),
//              lane 8 + i: alpha^(78+3i) flag_permute (s_{8+i} - out'_i)
//   segment 2: lane r (and 16 + r for r < 4): alpha^(40+r) * (y_r^3 - partial_rounds[r])
static constexpr u32 AIR_COOP_PAIRS = 32;   // row pairs per workgroup of 512 threads
static constexpr u32 AIR_COOP_MAX_PAIRS = AIR_COOP_PAIRS * AIR_INLINE_MAX_BLOCKS;  // (the round finishes inside the kernel)

__device__ __forceinline__ EF coop_row_sum(EF v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < 5; k++) v.v[k] = add(v.v[k], (u32)__shfl_xor(v.v[k], off, 16));
    return v;
}
__device__ __forceinline__ EF coop_row_get(const EF& v, u32 src) {
    EF r;
#pragma unroll
    for (int k = 0; k < 5; k++) r.v[k] = (u32)__shfl(v.v[k], src, 16);
    return r;
}
// y_l = sum_t col[t] * s_{(l - t) mod 16}  (the circulant of poseidon16.h: mds_circ16), plane by plane
__device__ __forceinline__ EF coop_mds(const EF& s, u32 l) {
    EF y;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        u64 acc = 0;
        static_for<0, 16>([&](auto T) {
            constexpr int t = decltype(T)::value;
            constexpr u32 COL[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
            acc += (u64)(u32)__shfl(s.v[k], (l - t) & 15, 16) * opaque_const(COL[t]);
        });
        y.v[k] = reduce40(acc);
    }
    return y;
}
__device__ __forceinline__ EF coop_mul_xk(EF a, u32 kx) {  // a * X^kx, kx < 5 (air::ef_mul_xk with a run-time exponent)
    for (u32 t = 0; t < kx; t++) {
        const EF b = a;
        a.v[0] = b.v[4], a.v[1] = b.v[0], a.v[2] = sub(b.v[1], b.v[4]), a.v[3] = b.v[2], a.v[4] = b.v[3];
    }
    return a;
}
__device__ __forceinline__ u32 coop_rc(u32 round, u32 l) {
    const PoseidonConsts& pc = poseidon_consts();
    return round < 4 ? pc.rc_init[round][l] : pc.rc_term[round - 4][l];
}
__device__ __forceinline__ EF coop_select(bool c, const EF& a, const EF& b) {
    EF r;
#pragma unroll
    for (int k = 0; k < 5; k++) r.v[k] = c ? a.v[k] : b.v[k];
    return r;
}
// lane l's share of segment `seg` (uniform per workgroup) of row pair j at the evaluation point: the 16 shares add up to
// air::eval_poseidon16_segment<EF, seg>
template <class Cols>
__device__ __forceinline__ EF coop_segment(const Cols& cols, u64 j, u32 zm, u32 seg, u32 l, const air::Extra& x) {
    auto col = [&](u32 c) { return cols.template at<false>(c, j, zm); };
    if (seg == 2) {
        EF c = ef_zero();
#pragma unroll
        for (u32 pass = 0; pass < 2; pass++) {
            const u32 r = pass * 16 + l;
            if (r < 20) {  // (lanes 4..15 skip the second pass together: whole groups of 12 lanes)
                const EF d = ef_sub(air::cube(col(air::POS_VIRT_Y + r)), col(57 + r));
                c = ef_add(c, ef_mul(x.alpha_powers[40 + r], d));
            }
        }
        return c;
    }
    const u32 input_col = seg == 0 ? 9u : seg == 1 ? 25u : seg == 3 ? (u32)air::POS_VIRT_E : 77u;
    const u32 r0 = seg == 0 ? 0u : seg == 1 ? 2u : seg == 3 ? 4u : 6u;
    EF s = air::cube(ef_add_base(col(input_col + l), coop_rc(r0, l)));
    s = coop_mds(s, l);
    if (seg != 4) {
        const u32 S = seg == 0 ? 0u : seg == 1 ? 1u : 2u;
        EF c = ef_mul(x.out_beta[S][l], air::cube(ef_add_base(s, coop_rc(r0 + 1, l))));
        if (l < 5) c = ef_sub(c, coop_mul_xk(col(air::POS_VIRT_O + 5 * S + l), l));
        if (seg == 0) {
            const EF flag_active = col(0), index_b = col(1), index_res = col(2), flag_half = col(3), flag_left = col(4);
            const EF offset_left = col(5), eff_first = col(6), eff_second = col(7), flag_permute = col(8);
            const EF one = ef_one();
            const EF omfl = ef_sub(one, flag_left);
            const EF index_a = ef_sub(eff_second, ef_mul_base(omfl, to_monty(4)));
            // precompile data: 1 + 4 half + 8 left + 16 left*offset + 2 permute
            const EF pdr = ef_add(ef_add(ef_add(ef_add(one, ef_mul_base(flag_half, to_monty(4))), ef_mul_base(flag_left, to_monty(8))),
                                         ef_mul_base(ef_mul(flag_left, offset_left), to_monty(16))),
                                  ef_mul_base(flag_permute, to_monty(2)));
            // one product per lane: A * B
            EF A = ef_zero(), B = ef_zero();
            const EF bv = l == 1 ? flag_active : l == 2 ? flag_half : l == 3 ? flag_left : flag_permute;  // bool_check operands
            if (l >= 1 && l <= 4) A = ef_sub(one, bv), B = bv;
            if (l == 5) A = flag_permute, B = ef_add(flag_half, flag_left);
            if (l == 6) A = flag_left, B = ef_sub(offset_left, eff_first);
            if (l == 7) A = omfl, B = ef_sub(index_a, eff_first);
            if (l >= 8 && l < 12) {
                A = x.logup_eq[l - 8];
                B = l == 8 ? pdr : l == 9 ? index_a : l == 10 ? index_b : index_res;
            }
            const EF P = ef_mul(A, B);
            // the bus value on lane 0: (sum of lanes 8..11 + eq[15]) * beta + flag
            const EF fp = ef_add(ef_add(coop_row_get(P, 8), coop_row_get(P, 9)), ef_add(coop_row_get(P, 10), coop_row_get(P, 11)));
            const EF bus = ef_add(ef_mul(ef_add(fp, x.logup_eq[15]), x.bus_beta), flag_active);
            const EF q = l == 0 ? bus : P;
            if (l < 8) c = ef_add(c, ef_mul(x.alpha_powers[l], q));
        }
        return c;
    }
    s = air::cube(ef_add_base(s, coop_rc(7, l)));
    s = coop_mds(s, l);
    const EF flag_half = col(3), flag_permute = col(8);
    const EF not_permute = ef_sub(ef_one(), flag_permute);
    const EF comp_last4 = ef_sub(not_permute, flag_half);
    const u32 i = l & 7;
    const bool lo = l < 8;
    // every lane: alpha^k * D * G with (lanes 0..7) k = 76 + 3 i, D = s + in_i - out_i, G = the output gate; (lanes 8..15) k = 78 + 3 i,
    // D = s - out'_i, G = flag_permute.  Lanes 0..7 add alpha^(77 + 3 i) * (s - out_i) * flag_permute.
    const EF o = col(lo ? 93 + i : 101 + i);
    const EF d = lo ? ef_sub(ef_add(s, col(9 + i)), o) : ef_sub(s, o);
    const EF gate = lo ? coop_select(i < 4, not_permute, comp_last4) : flag_permute;
    EF c = ef_mul(ef_mul(x.alpha_powers[(lo ? 76 : 78) + 3 * i], d), gate);
    if (lo) c = ef_add(c, ef_mul(ef_mul(x.alpha_powers[77 + 3 * i], ef_sub(s, o)), flag_permute));
    return c;
}

// grid = blocks_x * AIR_POS_SLOTS workgroups of 512 threads; group g = threadIdx.x / 16 of tile t evaluates pairs t * 32 + g, ...
template <class Cols>
__global__ __launch_bounds__(512) void k_air_round_pos_coop(Cols cols, u64 n_pairs, const air::Extra* __restrict__ extra, EqSplit eq,
                                                            u32* __restrict__ partial, u32 blocks_x, AirFinish fin) {
    __shared__ u32 lds[8 * 5];
    const u32 y = blockIdx.x / blocks_x, tile = blockIdx.x % blocks_x;
    u32 seg, z;  // slot y -> (segment, point): k_air_round, SEG < 0
    if (y < 4 * AIR_POS_POINTS) {
        const u32 s4 = y & 3, zi = y >> 2;
        seg = s4 < 2 ? s4 : s4 + 1;
        z = zi == 0 ? 0 : zi + 1;
    } else {
        seg = 2;
        z = y - 4 * AIR_POS_POINTS;
    }
    const u32 zm = to_monty(z);
    const u32 l = threadIdx.x & 15, g = threadIdx.x >> 4;
    EF acc = ef_zero();
    for (u64 j = (u64)tile * AIR_COOP_PAIRS + g;; j += (u64)blocks_x * AIR_COOP_PAIRS) {
        bool is_pad = false;
        u64 je = j;
        if (j >= n_pairs) {
            if (!(fin.pad_on && j == n_pairs)) break;
            is_pad = true;
            je = fin.pad_pair;
        }
        const EF v = coop_row_sum(coop_segment(cols, je, zm, seg, l, *extra));
        if (l == 0) acc = ef_add(acc, ef_mul(v, is_pad ? fin.pad_w : eq_split_at(eq, je)));
        if (is_pad) break;
    }
    u32 v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = wave_sum_u32(acc.v[k]);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) lds[wave * 5 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        u32 s = 0;
        for (u32 w = 0; w < 8; w++) s = add(s, lds[w * 5 + threadIdx.x]);
        lm_store_agent(partial + ((u64)y * blocks_x + tile) * 5 + threadIdx.x, s);
        lm_wait_stores();
    }
    air_finish_inline(partial, blocks_x, fin);
}

// one block per z: out[zi * 5 + k] = sum over the n = n_seg * blocks_x consecutive partials of point zi
// out = pinned result buffer; the block that finishes last publishes the sequence number.
// Block zi sums the n_main consecutive partials of point zi and, for the Poseidon table, adds the degree-3 segment:
// sum_t lag[zi][t] * (sum of the n_low partials of its slot t), lag = Lagrange basis of the nodes 0..3 at the point.
__global__ __launch_bounds__(256) void k_air_reduce(const u32* __restrict__ partial, u32 n_main, u32* __restrict__ out,
                                                    u32* __restrict__ done_counter, u32 seq, u32 n_low, u64 low_offset,
                                                    AirLagrange lag, u32* __restrict__ flag_word) {
    __shared__ u32 lds[20];
    const u32 zi = blockIdx.x;
    u32 v[5] = {0, 0, 0, 0, 0};
    for (u32 b = threadIdx.x; b < n_main; b += 256)
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] = add(v[k], partial[((u64)zi * n_main + b) * 5 + k]);
    if (n_low) {
        for (u32 t = 0; t < 4; t++) {
            u32 w[5] = {0, 0, 0, 0, 0};
            for (u32 b = threadIdx.x; b < n_low; b += 256)
#pragma unroll
                for (int k = 0; k < 5; k++) w[k] = add(w[k], partial[(low_offset + (u64)t * n_low + b) * 5 + k]);
            const u32 c = lag.c[zi][t];
#pragma unroll
            for (int k = 0; k < 5; k++) v[k] = add(v[k], mul(w[k], c));
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = wave_sum_u32(v[k]);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) lds[wave * 5 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        u32 s = 0;
        for (u32 w = 0; w < 4; w++) s = add(s, lds[w * 5 + threadIdx.x]);
        lm_store_system(out + zi * 5 + threadIdx.x, s);
    }
    if (threadIdx.x < 64) {  // the five writers are in wave 0
        lm_wait_stores();
        if (threadIdx.x == 0 && lm_ticket(done_counter) == gridDim.x - 1) {
            lm_store_agent(done_counter, 0);
            lm_publish_flag_word(flag_word, seq);
        }
    }
}

// fold: out[c][k][j] = lo + r (hi - lo); grid (blocks_x, n_cols + n_shift)
__global__ __launch_bounds__(256) void k_air_fold_base(BaseCols cols, u64 n_out, EF r, u32* __restrict__ out) {
    const u32 c = blockIdx.y;
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < n_out; j += (u64)gridDim.x * 256) {
        u32 lo, hi;
        if (c < cols.n_flat) {
            uint2 v = *reinterpret_cast<const uint2*>(cols.cols[c] + 2 * j);
            lo = v.x;
            hi = v.y;
        } else {
            const u32* p = cols.cols[c - cols.n_flat];
            lo = p[2 * j + 1];
            hi = p[(2 * j + 2 < cols.n_rows) ? 2 * j + 2 : cols.n_rows - 1];
        }
        EF o = ef_mul_base(r, sub(hi, lo));
        o.v[0] = add(o.v[0], lo);
#pragma unroll
        for (int k = 0; k < 5; k++) out[((u64)c * 5 + k) * n_out + j] = o.v[k];
    }
}
__global__ __launch_bounds__(256) void k_air_fold_ext(ExtCols cols, u64 n_out, EF r, u32* __restrict__ out) {
    const u32 c = blockIdx.y;
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < n_out; j += (u64)gridDim.x * 256) {
        EF lo, hi;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint2 v = *reinterpret_cast<const uint2*>(cols.buf + ((u64)c * 5 + k) * cols.n_rows + 2 * j);
            lo.v[k] = v.x;
            hi.v[k] = v.y;
        }
        EF o = ef_add(lo, ef_mul(r, ef_sub(hi, lo)));
#pragma unroll
        for (int k = 0; k < 5; k++) out[((u64)c * 5 + k) * n_out + j] = o.v[k];
    }
}

// the first TWO folds of the base columns in one pass (see FoldCols): out[c][k][j] from the base rows 4j .. 4j+3
__global__ __launch_bounds__(256) void k_air_fold2_base(BaseCols cols, u64 n_out, EF r1, EF r2, u32* __restrict__ out) {
    const u32 c = blockIdx.y;
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < n_out; j += (u64)gridDim.x * 256) {
        u32 a0, a1, a2, a3;
        if (c < cols.n_flat) {
            const uint4 v = *reinterpret_cast<const uint4*>(cols.cols[c] + 4 * j);
            a0 = v.x, a1 = v.y, a2 = v.z, a3 = v.w;
        } else {
            const u32* p = cols.cols[c - cols.n_flat];
            const u64 i = 4 * j;
            a0 = p[i + 1], a1 = p[i + 2], a2 = p[i + 3], a3 = p[(i + 4 < cols.n_rows) ? i + 4 : cols.n_rows - 1];
        }
        EF y0 = ef_mul_base(r1, sub(a1, a0)), y1 = ef_mul_base(r1, sub(a3, a2));
        y0.v[0] = add(y0.v[0], a0), y1.v[0] = add(y1.v[0], a2);
        const EF o = ef_add(y0, ef_mul(r2, ef_sub(y1, y0)));
#pragma unroll
        for (int k = 0; k < 5; k++) out[((u64)c * 5 + k) * n_out + j] = o.v[k];
    }
}

// Virtual columns of the Poseidon table (air_tables.h: POS_VIRT_Y / POS_VIRT_E / POS_VIRT_O): per row the 20 + 16 affine
// forms of the partial block over u = (beginning_full_rounds[1] (16), partial_rounds (20)), and the 3 x 5 coefficient planes
// of the challenge-weighted output blocks V_s = sum_i alpha^(k_s + i) * out_s[i].  Base-field in, base-field out.
__global__ __launch_bounds__(256) void k_air_virtual_columns(const u32* const* __restrict__ cols, u64 n_rows, u32* __restrict__ virt,
                                                             const air::Extra* __restrict__ extra) {
    const u64 r0 = (u64)blockIdx.x * 256 + threadIdx.x;
    if (r0 >= n_rows) return;
    u32 u[36];
#pragma unroll
    for (int j = 0; j < 36; j++) u[j] = cols[41 + j][r0];
    static_for<0, 20>([&](auto RR) {
        constexpr int r = decltype(RR)::value;
        virt[(u64)r * n_rows + r0] = add(dot_n<16 + r>(u, air::kPoseidonLinear.y[r]), air::kPoseidonLinear.y[r][36]);
    });
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        virt[(u64)(20 + i) * n_rows + r0] = add(dot_n<36>(u, air::kPoseidonLinear.fin[i]), air::kPoseidonLinear.fin[i][36]);
    });
    static_for<0, 3>([&](auto SS) {
        constexpr int s = decltype(SS)::value;
        u32 o[16];
#pragma unroll
        for (int i = 0; i < 16; i++) o[i] = s == 1 ? u[i] : cols[air::POS_OUT_COL[s] + i][r0];  // (segment 1's outputs are u[0..16))
        static_for<0, 5>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            u32 al[16];
#pragma unroll
            for (int i = 0; i < 16; i++) al[i] = extra->alpha_powers[air::POS_OUT_K0[s] + i].v[k];
            virt[(u64)(36 + 5 * s + k) * n_rows + r0] = dot_n<16>(o, al);
        });
    });
}

// does this round of the Poseidon table (extension-field columns) run on k_air_round_pos_coop?  LM_AIR_NO_COOP=1: never
static bool air_coop_round(u64 n_pairs, bool padding_pair) {
    static const bool on = getenv("LM_AIR_NO_COOP") == nullptr;
    return on && n_pairs + (padding_pair ? 1 : 0) <= AIR_COOP_MAX_PAIRS;
}
// does this round of the ExtensionOp table (extension-field columns) split its constraint list over air::EXT_PARTS workgroups per
// point?  Small rounds only: there the evaluation is one dependent chain per lane and four shorter chains side by side win; in the
// large rounds the parts would repeat the shared products.  LM_AIR_NO_COOP=1: never
static bool air_ext_parts_round(u64 n_pairs, bool padding_pair) {
    static const bool on = getenv("LM_AIR_NO_COOP") == nullptr;
    return on && n_pairs + (padding_pair ? 1 : 0) <= 256;
}
template <int TABLE, class T, class Cols, int SEG>
static int launch_segment(lm_ctx* ctx, hipStream_t stream, const Cols& c, const dim3& grid, u64 n_pairs, const air::Extra* extra, const EqSplit& eq,
                          u32* partial, const AirFinish& fin) {
    LM_LAUNCH_ON(ctx, stream, (k_air_round<TABLE, T, Cols, SEG>), dim3(grid.x * grid.y), dim3(256), 0, c, n_pairs, extra, eq, partial, grid.x, grid.y, fin);
    return LM_OK;
}
template <int TABLE, class T, class Cols>
static int launch_cols(lm_ctx* ctx, lm_air* a, const Cols& c, u64 n_pairs, u32 blocks, const EqSplit& eq, u32* partial, const AirFinish& fin) {
    const air::Extra* extra = a->d_extra;
    if constexpr (TABLE == air::T_POSEIDON16) {
        // Base-field rounds, large: one launch per segment (the combined kernel needs 256 VGPRs there, the per-segment ones
        // 165-240).  Extension-field rounds and small rounds: ONE launch with the segment in the workgroup id — the
        // combined extension-field kernel compiles to 148 VGPRs (3 waves per SIMD) where the specialised ones take
        // 256 + AGPRs (1 wave), and small rounds are latency bound anyway (the five chains run side by side).
        if constexpr (sizeof(T) == sizeof(EF)) {
            if (air_coop_round(n_pairs, fin.pad_on != 0)) {  // small extension-field round: 16 lanes per (pair, point, segment)
                LM_LAUNCH_ON(ctx, a->stream, (k_air_round_pos_coop<Cols>), dim3(blocks * AIR_POS_SLOTS), dim3(512), 0, c, n_pairs, extra, eq,
                             partial, blocks, fin);
                LM_HIP(hipGetLastError());
                return LM_OK;
            }
        }
        bool split = false;
        if constexpr (sizeof(T) == sizeof(u32)) split = n_pairs >= AIR_SPLIT_LAUNCH_PAIRS;
        if constexpr (sizeof(T) != sizeof(u32)) {
            launch_segment<TABLE, T, Cols, -1>(ctx, a->stream, c, dim3(blocks, AIR_POS_SLOTS), n_pairs, extra, eq, partial, fin);
        } else if (split) {
            const dim3 grid(blocks, AIR_POS_POINTS);
            AirFinish none = fin;   // five launches: the partials are summed by k_air_reduce (blocks > AIR_INLINE_MAX_BLOCKS here)
            none.out = nullptr;
            launch_segment<TABLE, T, Cols, 0>(ctx, a->stream, c, grid, n_pairs, extra, eq, partial, none);
            launch_segment<TABLE, T, Cols, 1>(ctx, a->stream, c, grid, n_pairs, extra, eq, partial, none);
            launch_segment<TABLE, T, Cols, 2>(ctx, a->stream, c, dim3(blocks, 4), n_pairs, extra, eq, partial, none);
            launch_segment<TABLE, T, Cols, 3>(ctx, a->stream, c, grid, n_pairs, extra, eq, partial, none);
            launch_segment<TABLE, T, Cols, 4>(ctx, a->stream, c, grid, n_pairs, extra, eq, partial, none);
        } else {
            launch_segment<TABLE, T, Cols, -1>(ctx, a->stream, c, dim3(blocks, AIR_POS_SLOTS), n_pairs, extra, eq, partial, fin);
        }
    } else if constexpr (TABLE == air::T_EXTENSION_OP && sizeof(T) == sizeof(EF)) {
        if (air_ext_parts_round(n_pairs, fin.pad_on != 0))
            launch_segment<TABLE, T, Cols, -2>(ctx, a->stream, c, dim3(blocks, a->deg * air::EXT_PARTS), n_pairs, extra, eq, partial, fin);
        else
            launch_segment<TABLE, T, Cols, -1>(ctx, a->stream, c, dim3(blocks, a->deg), n_pairs, extra, eq, partial, fin);
    } else {
        launch_segment<TABLE, T, Cols, -1>(ctx, a->stream, c, dim3(blocks, a->deg), n_pairs, extra, eq, partial, fin);
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}
template <int TABLE>
static int launch_round(lm_ctx* ctx, lm_air* a, u64 n_pairs, u32 blocks, const EqSplit& eq, u32* partial, const AirFinish& fin) {
    if (a->cur == -1) {
        BaseCols c{a->d_base_cols, 1ull << a->log_rows, a->n_cols + a->n_virt};
        return launch_cols<TABLE, u32, BaseCols>(ctx, a, c, n_pairs, blocks, eq, partial, fin);
    }
    if constexpr (TABLE == air::T_POSEIDON16) {  // (lm_air_new: only that table's first fold stays unmaterialised)
        if (a->cur == -2) {
            // LM_AIR_NO_AFFINE=1: round 1 on extension-field values throughout (A/B measurements)
            static const bool affine = getenv("LM_AIR_NO_AFFINE") == nullptr;
            if (affine && !air_coop_round(n_pairs, fin.pad_on != 0)) {
                FoldColsAff c;
                c.cols = a->d_base_cols, c.n_rows = 1ull << a->log_rows, c.n_flat = a->n_cols + a->n_virt, c.r = a->r1;
                c.tp.t1 = a->r1, c.tp.t2 = ef_mul(a->r1, a->r1), c.tp.t3 = ef_mul(c.tp.t2, a->r1);
                return launch_cols<TABLE, EF, FoldColsAff>(ctx, a, c, n_pairs, blocks, eq, partial, fin);
            }
            FoldCols<false> c{a->d_base_cols, 1ull << a->log_rows, a->n_cols + a->n_virt, a->r1};
            return launch_cols<TABLE, EF, FoldCols<false>>(ctx, a, c, n_pairs, blocks, eq, partial, fin);
        }
    }
    if constexpr (TABLE == air::T_EXECUTION) {  // (lm_air_new: its first fold stays unmaterialised too when round 1 runs on eval_execution_affine)
        if (a->cur == -2) {
            FoldColsExec c;
            c.cols = a->d_base_cols, c.n_rows = 1ull << a->log_rows, c.n_flat = a->n_cols;
            EF t = a->r1;
            for (int m = 0; m < 5; m++) {
                c.ex.tp[m] = t;
                t = ef_mul(t, a->r1);
            }
            for (int i = 0; i < 4; i++) c.ex.eq_beta[i] = ef_mul(a->h_extra.logup_eq[i], a->h_extra.bus_beta);
            c.ex.eq15_beta = ef_mul(a->h_extra.logup_eq[15], a->h_extra.bus_beta);
            return launch_cols<TABLE, EF, FoldColsExec>(ctx, a, c, n_pairs, blocks, eq, partial, fin);
        }
    }
    ExtCols c{a->ef[a->cur], 1ull << (a->log_rows - a->round)};  // (rows of the folded table: n_pairs may be the active prefix only)
    return launch_cols<TABLE, EF, ExtCols>(ctx, a, c, n_pairs, blocks, eq, partial, fin);
}

extern "C" {

void lm_air_free(lm_ctx* ctx, lm_air* a) {
    if (!a) return;
    // Join WITHOUT a host synchronisation (two hipStreamSynchronize per session, three sessions: ~0.1 ms of idle GPU between the AIR
    // sumcheck and the opening): the context's stream waits, on the device, for everything the session enqueued on its own stream.  The
    // pool is ordered on ctx->stream, so the blocks returned below are handed out again only behind that point; the uploads of
    // lm_air_new went through the staging ring (lm_stage_upload copies the host image before it returns), so *a may go at once.
    if (a->stream && a->stream != ctx->stream) {
        if (hipEventRecord(ctx->fork_event, a->stream) != hipSuccess || hipStreamWaitEvent(ctx->stream, ctx->fork_event, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(a->stream);
        }
    }
    lm_pool_free(ctx, a->d_sync);
    lm_pool_free(ctx, (void*)a->d_base_cols);
    lm_pool_free(ctx, a->d_virt);
    for (int i = 0; i < 2; i++) lm_pool_free(ctx, a->ef[i]);
    lm_pool_free(ctx, a->d_extra);
    lm_pool_free(ctx, a->eqt.d_buf);
    lm_pool_free(ctx, a->d_partial);
    delete a;
}

int lm_air_new(lm_ctx* ctx, uint32_t table, const uint32_t* const* d_cols, uint32_t log_rows, const uint32_t* eq_point,
               const uint32_t alpha[5], const uint32_t* logup_eq16, const uint32_t bus_beta[5], lm_air** out) {
    LM_REQUIRE(ctx && d_cols && eq_point && alpha && logup_eq16 && bus_beta && out);
    LM_REQUIRE(table <= 2 && log_rows >= 1 && log_rows <= 30);
    lm_air* a = new lm_air();
    a->table = (int)table;
    a->log_rows = log_rows;
    a->n_cols = air::n_columns((int)table);
    a->n_shift = air::n_shift((int)table);
    a->n_virt = table == air::T_POSEIDON16 ? air::POS_N_VIRT : 0;  // (that table has no shift columns)
    a->n_active = 1ull << log_rows;
    a->h_point.resize(log_rows);
    for (u32 j = 0; j < log_rows; j++) memcpy(a->h_point[j].v, eq_point + 5 * j, 20);
    a->deg = air::degree((int)table);
    const u64 half = 1ull << (log_rows - 1);
    const u64 ef_words0 = (u64)(a->n_cols + a->n_virt + a->n_shift) * 5 * half;
    air::Extra& hx = a->h_extra;
    EF al, p = ef_one();
    memcpy(al.v, alpha, 20);
    for (int i = 0; i < air::MAX_ALPHA; i++) {  // air_alpha.powers() (prove_execution.rs:154-155)
        hx.alpha_powers[i] = p;
        p = ef_mul(p, al);
    }
    // beta[s][j] = sum_i alpha^(k_s + i) * MDS[i][j], MDS[i][j] = col[(i - j) mod 16] (air_tables.h: POS_VIRT_O)
    static constexpr u32 MDS_COL[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
    for (int sgm = 0; sgm < 3; sgm++)
        for (int j = 0; j < 16; j++) {
            EF b = ef_zero();
            for (int i = 0; i < 16; i++)
                b = ef_add(b, ef_mul_base(hx.alpha_powers[air::POS_OUT_K0[sgm] + i], to_monty(MDS_COL[(16 + i - j) & 15])));
            hx.out_beta[sgm][j] = b;
        }
    memcpy(hx.logup_eq, logup_eq16, 16 * 20);
    memcpy(hx.bus_beta.v, bus_beta, 20);
    bool ok = lm_pool_alloc(ctx, (void**)&a->d_base_cols, (a->n_cols + a->n_virt) * sizeof(u32*)) == hipSuccess &&
              (a->n_virt == 0 || lm_pool_alloc_t(ctx, &a->d_virt, ((u64)a->n_virt << log_rows) * 4) == hipSuccess) &&
              lm_pool_alloc_t(ctx, &a->ef[0], std::max<u64>(ef_words0, 64) * 4) == hipSuccess &&
              lm_pool_alloc_t(ctx, &a->ef[1], std::max<u64>(ef_words0 / 2, 64) * 4) == hipSuccess &&
              lm_pool_alloc_t(ctx, &a->d_extra, sizeof(air::Extra)) == hipSuccess &&
              lm_pool_alloc_t(ctx, &a->d_sync, 256) == hipSuccess &&
              lm_pool_alloc_t(ctx, &a->eqt.d_buf, PrefixEqTables::words_needed(log_rows) * 4) == hipSuccess &&
              lm_pool_alloc_t(ctx, &a->d_partial, ((u64)AIR_MAX_BLOCKS * (table == air::T_POSEIDON16 ? AIR_POS_SLOTS : a->deg) * 5 + 64) * 4) == hipSuccess;
    a->res_off = 256 + 64 * table;
    if (!ok) {
        lm_set_error("lm_air_new: device allocation failed");
        lm_air_free(ctx, a);
        return LM_E_NOMEM;
    }
    a->eqt.buf_words = PrefixEqTables::words_needed(log_rows);
    a->h_cols.assign(d_cols, d_cols + a->n_cols);
    for (u32 v = 0; v < a->n_virt; v++) a->h_cols.push_back(a->d_virt + ((u64)v << log_rows));
    // FoldCols reads four consecutive rows of a column with one 16-byte load (LM_AIR_NO_LAZY_FOLD=1: every fold is materialised)
    static const bool lazy = getenv("LM_AIR_NO_LAZY_FOLD") == nullptr;
    // The Poseidon table only: the ExtensionOp evaluation with its shift views compiles to 445 VGPRs over FoldCols (one wave per SIMD:
    // a 2^18-row table took 7 ms longer), and the execution table's first fold is small.
    // The execution table joins when its round 1 runs on base coefficients (eval_execution_affine: LM_AIR_NO_AFFINE=1 switches that off).
    static const bool affine = getenv("LM_AIR_NO_AFFINE") == nullptr;
    a->lazy_ok = lazy && log_rows >= 2 && (table == air::T_POSEIDON16 || (table == air::T_EXECUTION && affine));
    for (const u32* cp : a->h_cols) a->lazy_ok = a->lazy_ok && (reinterpret_cast<uintptr_t>(cp) & 15) == 0;
    int rc;
    if ((rc = lm_stage_upload(ctx, (void*)a->d_base_cols, a->h_cols.data(), a->h_cols.size() * sizeof(u32*))) ||
        (rc = lm_stage_upload(ctx, a->d_extra, &a->h_extra, sizeof(air::Extra)))) {
        lm_air_free(ctx, a);
        return rc;
    }
    if (a->n_virt)
        LM_LAUNCH(ctx, k_air_virtual_columns, dim3((unsigned)(((1ull << log_rows) + 255) / 256)), dim3(256), 0,
                  (const u32* const*)a->d_base_cols, 1ull << log_rows, a->d_virt, (const air::Extra*)a->d_extra);
    rc = a->eqt.build(ctx, eq_point, log_rows);
    if (rc) {
        lm_air_free(ctx, a);
        return rc;
    }
    // fork: everything above (and the caller's columns) was produced on ctx->stream; the session continues on its own stream
    // (LM_AIR_SINGLE_STREAM=1 keeps the session on ctx->stream: for A/B measurements)
    static const bool single = getenv("LM_AIR_SINGLE_STREAM") != nullptr;
    a->aux = single ? -1 : (int)(table % lm_ctx::N_AUX);
    hipError_t e = hipMemsetAsync(a->d_sync, 0, 256, ctx->stream);
    hipStream_t side = ctx->stream;
    if (!single) {
        if ((rc = lm_aux_stream(ctx, a->aux, &side))) {
            lm_air_free(ctx, a);
            return rc;
        }
        if (e == hipSuccess) e = hipEventRecord(ctx->fork_event, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(side, ctx->fork_event, 0);
    }
    if (e != hipSuccess) {
        lm_set_error("lm_air_new: stream fork failed: %s", hipGetErrorString(e));
        lm_air_free(ctx, a);
        return LM_E_DEVICE;
    }
    a->stream = side;
    *out = a;
    return LM_OK;
}

int lm_air_set_active_rows(lm_air* a, uint64_t n_active_rows) {
    if (!a || a->round != 0 || n_active_rows == 0 || n_active_rows > (1ull << a->log_rows)) {
        lm_set_error("lm_air_set_active_rows: before the first round, 1 <= n_active_rows <= 2^log_rows");
        return LM_E_INVALID;
    }
    a->n_active = n_active_rows;
    return LM_OK;
}
uint32_t lm_air_degree(const lm_air* a) { return a ? a->deg : 0; }
uint32_t lm_air_n_evals(const lm_air* a) { return a ? a->n_cols + a->n_shift : 0; }

// raw[zi] for z = 0, 2, 3, .., degree  (degree EF values = 5 * degree words).  lm_air_round_launch only enqueues the round's
// kernels (result -> the session's slice of the pinned buffer), lm_air_round_wait collects it: the sessions of one batched
// round (prove_batched_air_sumcheck) are launched back to back and cost one host round trip instead of one each.
int lm_air_round_launch(lm_ctx* ctx, lm_air* a) {
    LM_REQUIRE(ctx && a && a->round < a->log_rows && a->pending_seq == 0);
    const u32 p = a->log_rows - a->round - 1;
    const u64 n_pairs_full = 1ull << p;
    // active prefix: after `round` folds the first ceil(n_active / 2^round) rows can differ from the padding row; the pairs
    // behind them are identical (folds of identical rows are identical), the last pair of the table among them
    const u64 rows_active = (a->n_active + (1ull << a->round) - 1) >> a->round;
    const u64 pairs_active = (rows_active + 1) / 2;
    const bool prefix = pairs_active + 1 <= n_pairs_full;
    const u64 n_pairs = prefix ? pairs_active : n_pairs_full;
    const bool pos = a->table == air::T_POSEIDON16;
    u32 blocks = n_pairs <= 256 ? 1 : (u32)std::min<u64>((n_pairs + 255) / 256, AIR_MAX_BLOCKS);
    // an active prefix is not a power of two: keep the tile count a multiple of 8 (the XCD-aware workgroup mapping of
    // k_air_round needs it — without it the workgroups that share a row tile spread over all XCDs and the round's HBM
    // traffic grows 6x; surplus tiles find no pairs and write zero partials)
    if (blocks > 8) blocks = std::min<u32>((blocks + 7) & ~7u, AIR_MAX_BLOCKS);
    // the padding pair is one more slot of the grid-stride walk (k_air_round): when the active pairs fill every thread's
    // iterations exactly and those are few, add workgroups so that it lands on an idle thread
    if (prefix && n_pairs % ((u64)blocks * 256) == 0 && n_pairs / ((u64)blocks * 256) < 4 && blocks < AIR_MAX_BLOCKS)
        blocks += (blocks & 7) == 0 ? 8 : 1;
    if (pos && a->cur != -1 && air_coop_round(n_pairs, prefix))  // tiles of AIR_COOP_PAIRS pairs, the padding pair included
        blocks = (u32)((n_pairs + (prefix ? 1 : 0) + AIR_COOP_PAIRS - 1) / AIR_COOP_PAIRS);
    u32* s = a->d_partial;
    int rc;
    const EqSplit eq = a->eqt.at(p);
    static const AirLagrange pos_lag = [] {  // Lagrange basis of the nodes 0,1,2,3 at z = 0,2,3,..,10 (exact field constants)
        AirLagrange l;
        for (u32 zi = 0; zi < AIR_POS_POINTS; zi++) {
            const u32 z = zi == 0 ? 0 : zi + 1;
            for (u32 t = 0; t < 4; t++) {
                u32 num = ONE, den = ONE;
                for (u32 m = 0; m < 4; m++) {
                    if (m == t) continue;
                    num = mul(num, sub(to_monty(z), to_monty(m)));
                    den = mul(den, sub(to_monty(t), to_monty(m)));
                }
                l.c[zi][t] = mul(num, inv(den));
            }
        }
        return l;
    }();
    const u32 seq = ++ctx->res_seq;
    AirFinish fin;
    memset(&fin, 0, sizeof fin);
    if (pos) fin.lag = pos_lag;
    const bool inline_finish = blocks <= AIR_INLINE_MAX_BLOCKS && !(pos && a->cur == -1 && n_pairs >= AIR_SPLIT_LAUNCH_PAIRS);
    fin.out = inline_finish ? ctx->h_res + a->res_off : nullptr;
    fin.flag_word = ctx->h_res + lm_ctx::RES_FLAG + 1 + a->aux;
    fin.done_counter = a->d_sync;
    fin.seq = seq;
    fin.deg = a->deg;
    const bool ext_parts = a->table == air::T_EXTENSION_OP && a->cur != -1 && air_ext_parts_round(n_pairs, prefix);
    fin.n_main = pos ? 4 * blocks : ext_parts ? air::EXT_PARTS * blocks : blocks;
    fin.n_low = pos ? blocks : 0u;
    fin.low_offset = (u64)4 * AIR_POS_POINTS * blocks;
    if (prefix) {
        // sum_{j >= pairs_active} eq(point[0..p), j) = mle_of_zeros_then_ones (poly/src/mle/mle_custom.rs:4-19; coordinate 0 is
        // the most significant bit of j, lm_eqsplit.h): walking the bits of the threshold from the top, every index that agrees
        // so far and has a 1 where the threshold has a 0 is larger
        EF acc = ef_zero(), prefix_w = ef_one();
        for (u32 b = 0; b < p; b++) {
            const EF x = a->h_point[b];
            if ((pairs_active >> (p - 1 - b)) & 1) {
                prefix_w = ef_mul(prefix_w, x);
            } else {
                acc = ef_add(acc, ef_mul(prefix_w, x));
                prefix_w = ef_mul(prefix_w, ef_sub(ef_one(), x));
            }
        }
        fin.pad_on = 1;
        fin.pad_pair = n_pairs_full - 1;
        fin.pad_w = ef_add(acc, prefix_w);  // (+ the threshold index itself)
    }
    if (a->table == air::T_EXECUTION)
        rc = launch_round<air::T_EXECUTION>(ctx, a, n_pairs, blocks, eq, s, fin);
    else if (a->table == air::T_EXTENSION_OP)
        rc = launch_round<air::T_EXTENSION_OP>(ctx, a, n_pairs, blocks, eq, s, fin);
    else
        rc = launch_round<air::T_POSEIDON16>(ctx, a, n_pairs, blocks, eq, s, fin);
    if (rc) return rc;
    if (!inline_finish)
        LM_LAUNCH_ON(ctx, a->stream, k_air_reduce, dim3(a->deg), dim3(256), 0, (const u32*)s, fin.n_main, ctx->h_res + a->res_off, a->d_sync, seq,
                     fin.n_low, fin.low_offset, fin.lag, fin.flag_word);
    LM_HIP(hipGetLastError());
    a->pending_seq = seq;
    return LM_OK;
}
int lm_air_round_wait(lm_ctx* ctx, lm_air* a, uint32_t* out_raw) {
    LM_REQUIRE(ctx && a && out_raw && a->pending_seq != 0);
    int rc = lm_wait_result_aux(ctx, a->aux, a->pending_seq);
    a->pending_seq = 0;
    if (rc) return rc;
    memcpy(out_raw, ctx->h_res + a->res_off, (u64)a->deg * 20);
    return LM_OK;
}
int lm_air_round(lm_ctx* ctx, lm_air* a, uint32_t* out_raw) {
    int rc = lm_air_round_launch(ctx, a);
    return rc ? rc : lm_air_round_wait(ctx, a, out_raw);
}

int lm_air_bind(lm_ctx* ctx, lm_air* a, const uint32_t challenge[5]) {
    LM_REQUIRE(ctx && a && challenge && a->round < a->log_rows);
    EF r;
    memcpy(r.v, challenge, 20);
    const u64 n_out = 1ull << (a->log_rows - a->round - 1);
    const u32 blocks = (u32)std::min<u64>((n_out + 255) / 256, 1024);
    const dim3 grid(blocks, a->n_cols + a->n_virt + a->n_shift);
    if (a->cur == -1 && a->lazy_ok) {  // the first fold stays unmaterialised: round 1 reads the base columns through FoldCols
        a->r1 = r;
        a->cur = -2;
    } else if (a->cur == -2) {
        BaseCols c{a->d_base_cols, 1ull << a->log_rows, a->n_cols + a->n_virt};
        LM_LAUNCH_ON(ctx, a->stream, k_air_fold2_base, grid, dim3(256), 0, c, n_out, a->r1, r, a->ef[0]);
        a->cur = 0;
    } else if (a->cur < 0) {
        BaseCols c{a->d_base_cols, 1ull << a->log_rows, a->n_cols + a->n_virt};
        LM_LAUNCH_ON(ctx, a->stream, k_air_fold_base, grid, dim3(256), 0, c, n_out, r, a->ef[0]);
        a->cur = 0;
    } else {
        ExtCols c{a->ef[a->cur], 2 * n_out};
        LM_LAUNCH_ON(ctx, a->stream, k_air_fold_ext, grid, dim3(256), 0, c, n_out, r, a->ef[1 - a->cur]);
        a->cur = 1 - a->cur;
    }
    LM_HIP(hipGetLastError());
    a->round++;
    return LM_OK;
}

// final_column_evals (air_sumcheck.rs:294-296): (n_cols + n_shift) EF values after log_rows bindings
int lm_air_final_evals(lm_ctx* ctx, lm_air* a, uint32_t* out) {
    int rc = lm_air_final_evals_begin(ctx, a);
    return rc ? rc : lm_air_final_evals_end(ctx, a, out);
}
// the same in two halves (the sessions of a batch publish side by side, one wait each instead of three round trips)
int lm_air_final_evals_begin(lm_ctx* ctx, lm_air* a) {
    LM_REQUIRE(ctx && a && a->round == a->log_rows && a->cur >= 0 && a->pending_seq == 0 && a->h_final.empty());
    const u32 n = (a->n_cols + a->n_shift) * 5;
    if (a->aux < 0) {  // LM_AIR_SINGLE_STREAM: the sessions share one flag word and one slice of the result buffer — fetched at once
        a->h_final.resize(n);
        return lm_fetch_words(ctx, -1, a->ef[a->cur], n, nullptr, 0, 1024u, a->h_final.data());
    }
    // n <= 545 words: the aux slices of the result buffer start at 1024, 1024 words each
    return lm_fetch_words_begin(ctx, a->aux, a->ef[a->cur], n, nullptr, 0, 1024u * (1 + a->aux), &a->pending_seq);
}
int lm_air_final_evals_end(lm_ctx* ctx, lm_air* a, uint32_t* out) {
    LM_REQUIRE(ctx && a && out && (a->pending_seq != 0 || !a->h_final.empty()));
    const u32 n = (a->n_cols + a->n_shift) * 5;
    if (!a->h_final.empty()) {
        memcpy(out, a->h_final.data(), (size_t)n * 4);
        a->h_final.clear();
        return LM_OK;
    }
    const u32 seq = a->pending_seq;
    a->pending_seq = 0;
    return lm_fetch_words_end(ctx, a->aux, seq, 1024u * (1 + a->aux), n, out);
}

}  // extern "C"
