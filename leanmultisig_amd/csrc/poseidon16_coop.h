// Wave-cooperative Poseidon1-16: ONE permutation on 16 adjacent lanes (a DPP "row"), one state word per lane; a wave64
// runs 4 independent permutations.  For the latency-bound places of the prover — the top of every Merkle tree, the
// leaf sponges of the small WHIR-round trees — where one-permutation-per-lane leaves the chip empty and every
// permutation is a ~7 k instruction dependent chain (~17 us).  Here the chain is ~1.1 k instructions (~3 us):
//   full round     S-box lane-local; circulant MDS = 15 DPP row rotations + 16 multiply-adds by uniform constants
//   partial block  the affine forms of gen_poseidon_consts.py::linearise (hashing variant).  Lane l owns y_l (and y_{16+l}
//                  for l < 4) and the exit-state word F_l.  Their parts that depend on the S-box outputs c of the 4th
//                  full round need c rotated to every lane once (15 rotations shared by the three sums); then the 20
//                  S-boxes run in sequence, each followed by one 16-lane broadcast of q_r and one multiply-add per sum.
// Same permutation as poseidon16_permute (poseidon1_koalabear_16.rs:873-912); parity: tests/test_commit_gpu.py.
//
// The per-lane coefficient table is built on the host (lm_coop_table_build) for the rotation direction that a probe kernel
// observes, so nothing here depends on how the ISA manual words "rotate right".
#pragma once
#include "kb.h"
#include "poseidon16.h"

namespace kb {

// per lane l = lane & 15 (COOP_TAB_STRIDE words each), after a 16-word uniform header holding the MDS coefficient of
// every rotation amount
struct CoopLaneTab {
    u32 rc[8];                   // round constants of the 8 full rounds for this lane
    u32 ya[16], yb[16], ff[16];  // coefficient of rot_k(c) in y_l, y_{16+l}, F_l
    u32 qa[20], qb[20], fq[20];  // coefficient of q_r in y_l, y_{16+l}, F_l
    u32 ca, cb, cf;              // constants
    u32 pad[9];
};
static constexpr u32 COOP_TAB_STRIDE = sizeof(CoopLaneTab) / 4;  // 128
static constexpr u32 COOP_TAB_WORDS = 16 + 16 * COOP_TAB_STRIDE;
static_assert(COOP_TAB_STRIDE == 128, "CoopLaneTab layout");

#if defined(__HIPCC__)
// rot_k within the 16-lane row (direction probed by the host, see lm_coop_table_build)
template <int K>
__device__ __forceinline__ u32 coop_rot(u32 x) {
    return (u32)__builtin_amdgcn_mov_dpp((int)x, 0x120 | K, 0xf, 0xf, false);  // row_ror:K
}

struct CoopRegs {
    CoopLaneTab t;
    u32 mds[16];
};
__device__ __forceinline__ void coop_load(CoopRegs& R, const u32* __restrict__ tab) {
    const u32 l = threadIdx.x & 15;
    const u32* src = tab + 16 + l * COOP_TAB_STRIDE;
    u32* dst = reinterpret_cast<u32*>(&R.t);
#pragma unroll
    for (u32 i = 0; i < COOP_TAB_STRIDE - 9; i++) dst[i] = src[i];
#pragma unroll
    for (int k = 0; k < 16; k++) R.mds[k] = tab[k];  // uniform
}

// lane K of every 16-lane row to the whole row: one DPP move (row_newbcast, gfx90a+) instead of a trip through the LDS crossbar
// (ds_bpermute, ~60 cycles of latency on each of the 20 strictly sequential partial rounds)
template <int K>
__device__ __forceinline__ u32 coop_bcast(u32 x) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 | K, 0xf, 0xf, false);  // row_newbcast:K
}
__device__ __forceinline__ u32 coop_mds(u32 s, const CoopRegs& R) {
    // four partial sums: the 16 multiply-adds are 4 deep instead of a chain of 16 (the permutation is one dependent chain per row)
    u64 acc[4] = {(u64)s * R.mds[0], 0, 0, 0};
    static_for<1, 16>([&](auto K) {
        constexpr int k = decltype(K)::value;
        acc[k & 3] += (u64)coop_rot<k>(s) * R.mds[k];
    });
    return reduce40((acc[0] + acc[1]) + (acc[2] + acc[3]));
}

// s: this lane's state word; returns the permuted word.  All 16 lanes of the row must be active.
__device__ __forceinline__ u32 coop_permute(u32 s, const CoopRegs& R) {
    const CoopLaneTab& T = R.t;
#pragma unroll
    for (int r = 0; r < 3; r++) s = coop_mds(cube(add(s, T.rc[r])), R);
    const u32 c = cube(add(s, T.rc[3]));
    // parts of y_l, y_{16+l}, F_l that depend on c: 16 products each, folded every 3 (after the first 4)
    u64 A = (u64)c * T.ya[0], B = (u64)c * T.yb[0], F = (u64)c * T.ff[0];
    static_for<1, 16>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const u32 x = coop_rot<k>(c);
        if (k >= 4 && (k - 4) % 3 == 0) {
            A = fold32(A);
            B = fold32(B);
            F = fold32(F);
        }
        A += (u64)x * T.ya[k];
        B += (u64)x * T.yb[k];
        F += (u64)x * T.ff[k];
    });
    // after k = 15: 16 = 4 + 3*4 products, the last fold came before k = 13 -> 3 products since: no room left
    A = fold32(A);
    B = fold32(B);
    F = fold32(F);
    // 20 partial rounds: owner lane of round r is r & 15 (sum A for r < 16, B for r >= 16)
    static_for<0, 20>([&](auto RR) {
        constexpr int r = decltype(RR)::value;
        const u32 y = add(reduce(fold32(r < 16 ? A : B)), r < 16 ? T.ca : T.cb);  // meaningful in the owner lane only
        const u32 q = coop_bcast<(r & 15)>(cube(y));
        if (r % 3 == 0) {  // room: one product per round since the last fold
            A = fold32(A);
            B = fold32(B);
            F = fold32(F);
        }
        A += (u64)q * T.qa[r];
        B += (u64)q * T.qb[r];
        F += (u64)q * T.fq[r];
    });
    s = add(reduce(fold32(F)), T.cf);
#pragma unroll
    for (int r = 4; r < 8; r++) s = coop_mds(cube(add(s, T.rc[r])), R);
    return s;
}
// compression mode: perm(x) + x
__device__ __forceinline__ u32 coop_compress(u32 s, const CoopRegs& R) { return add(coop_permute(s, R), s); }
#endif

// Host: build the table.  src_of_rot1 = the lane whose value lane 0 receives from coop_rot<1> (15: rot_k delivers
// x[(l - k) & 15]; 1: x[(l + k) & 15]), measured by k_coop_probe.
inline bool lm_coop_table_build(u32 src_of_rot1, u32* out /* COOP_TAB_WORDS */) {
    if (src_of_rot1 != 15 && src_of_rot1 != 1) return false;
    static constexpr u32 COL[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
    auto src = [&](u32 l, u32 k) { return src_of_rot1 == 15 ? (l + 16 - k) & 15 : (l + k) & 15; };
    // MDS: out_l = sum_j col[(l - j) & 15] s_j; rot_k delivers j = src(l, k), so its coefficient is col[(l - src(l,k)) & 15],
    // the same for every lane
    for (u32 k = 0; k < 16; k++) out[k] = COL[(0 + 16 - src(0, k)) & 15];
    const PoseidonLinearHash& L = kPoseidonLinearHash;
    for (u32 l = 0; l < 16; l++) {
        CoopLaneTab t;
        for (int r = 0; r < 4; r++) {
            t.rc[r] = kPoseidonHost.rc_init[r][l];
            t.rc[4 + r] = kPoseidonHost.rc_term[r][l];
        }
        for (u32 k = 0; k < 16; k++) {
            const u32 j = src(l, k);
            t.ya[k] = L.y[l][j];
            t.yb[k] = l < 4 ? L.y[16 + l][j] : 0;
            t.ff[k] = L.fin[l][j];
        }
        for (u32 r = 0; r < 20; r++) {
            t.qa[r] = L.y[l][16 + r];
            t.qb[r] = l < 4 ? L.y[16 + l][16 + r] : 0;
            t.fq[r] = L.fin[l][16 + r];
        }
        t.ca = L.y[l][36];
        t.cb = l < 4 ? L.y[16 + l][36] : 0;
        t.cf = L.fin[l][36];
        for (int i = 0; i < 9; i++) t.pad[i] = 0;
        const u32* w = reinterpret_cast<const u32*>(&t);
        for (u32 i = 0; i < COOP_TAB_STRIDE; i++) out[16 + l * COOP_TAB_STRIDE + i] = w[i];
    }
    return true;
}

}  // namespace kb
