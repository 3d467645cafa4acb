// Quad-cooperative Poseidon1-16: ONE permutation on the 4 lanes of a DPP quad, four state words per lane (lane class
// q = lane & 3 holds words 4q .. 4q+3).  For launches that are one permutation deep and too small to fill the chip with one
// permutation per lane — the proof-of-work searches: 2^16..2^17 candidates are 1-2 waves per SIMD of a ~6 k instruction
// dependent chain (~35 us, 30 searches per proof, all on the critical path).  Here the chain is ~2.2 k instructions and the
// same candidates occupy four times the lanes, so the search is throughput bound again.  (The 16-lane variant of
// poseidon16_coop.h shortens the chain further but needs 3x the instructions per permutation and ~140 VGPRs of tables.)
//   full round     S-boxes lane-local (4 per lane); circulant MDS: the other 12 words arrive by three quad rotations of the
//                  four registers, and because the matrix is circulant the coefficient of "word k of the lane d places to the
//                  right" for "my a-th output" is col[(a - k - 4d) mod 16] — the same literal in every lane;
//   partial block  affine forms of gen_poseidon_consts.py::linearise: class q owns y_q, y_{q+4}, .., y_{q+16} and the exit
//                  words F_{4q..4q+3}.  These coefficients differ per class: they come from a 4 x 368-word table in LDS
//                  (compile-time offsets, base address = class).  The 20 S-boxes run in sequence: the owner's cube is
//                  broadcast to the quad and every lane adds its multiples (zero coefficients where a y is already used).
// Same permutation as poseidon16_permute (poseidon1_koalabear_16.rs:873-912); parity: tests/test_commit_gpu.py
// (lm_poseidon16_permute_quad against the oracle) and the PoW tests.
#pragma once
#include "poseidon16.h"

namespace kb {

// per class (words): rc0[4] | bias[7][4] | ya[5][16] | yc[5] | fa[4][16] | fc[4] | yq[5][20] | fq[4][20]
static constexpr u32 QUAD_RC0 = 0, QUAD_BIAS = 4, QUAD_YA = 32, QUAD_YC = 112, QUAD_FA = 117, QUAD_FC = 181, QUAD_YQ = 185,
                     QUAD_FQ = 285, QUAD_STRIDE = 368, QUAD_TAB_WORDS = 4 * QUAD_STRIDE;

#if defined(__HIPCC__)
// lane q receives x of lane (q + K) & 3
template <int K>
__device__ __forceinline__ u32 quad_rot(u32 x) {
    constexpr int ctrl = ((K + 0) & 3) | (((K + 1) & 3) << 2) | (((K + 2) & 3) << 4) | (((K + 3) & 3) << 6);  // quad_perm
    return (u32)__builtin_amdgcn_mov_dpp((int)x, ctrl, 0xf, 0xf, false);
}
template <int C>
__device__ __forceinline__ u32 quad_from(u32 x) {
    return (u32)__builtin_amdgcn_mov_dpp((int)x, C * 0x55, 0xf, 0xf, false);  // quad_perm:[C,C,C,C]
}

// copies the table (global, QUAD_TAB_WORDS) to LDS; the caller synchronises
__device__ __forceinline__ void quad_load_table(u32* __restrict__ lds, const u32* __restrict__ tab) {
    for (u32 i = threadIdx.x; i < QUAD_TAB_WORDS; i += blockDim.x) lds[i] = tab[i];
}

// all 16 words as seen from this lane: w[d][k] = word 4((q + d) & 3) + k
__device__ __forceinline__ void quad_gather(const u32 s[4], u32 w[4][4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        w[0][k] = s[k];
        w[1][k] = quad_rot<1>(s[k]);
        w[2][k] = quad_rot<2>(s[k]);
        w[3][k] = quad_rot<3>(s[k]);
    }
}
// s <- circ(col) s + bias  (bias: 4 words of this class, may be nullptr)
__device__ __forceinline__ void quad_mds(u32 s[4], const u32* __restrict__ bias) {
    const u32 c1 = opaque_const(1), c2 = opaque_const(2), c3 = opaque_const(3), c13 = opaque_const(13);
    const u32 c22 = opaque_const(22), c67 = opaque_const(67), c15 = opaque_const(15), c63 = opaque_const(63);
    const u32 c101 = opaque_const(101), c17 = opaque_const(17), c11 = opaque_const(11), c51 = opaque_const(51);
    const u32 C[16] = {c1, c3, c13, c22, c67, c2, c15, c63, c101, c1, c2, c17, c11, c1, c51, c1};
    u32 w[4][4];
    quad_gather(s, w);
    static_for<0, 4>([&](auto A) {
        constexpr int a = decltype(A)::value;
        u64 acc = bias ? (u64)bias[a] : 0;
        static_for<0, 4>([&](auto D) {
            constexpr int d = decltype(D)::value;
            static_for<0, 4>([&](auto K) {
                constexpr int k = decltype(K)::value;
                acc += (u64)w[d][k] * C[(64 + a - k - 4 * d) & 15];
            });
        });
        s[a] = reduce40(acc);
    });
}

// s: words 4q..4q+3 of the state; T: this class's table in LDS (lds + q * QUAD_STRIDE).  All four lanes of the quad active.
__device__ __forceinline__ void quad_permute(u32 s[4], const u32* __restrict__ T) {
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = add(s[k], T[QUAD_RC0 + k]);
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) s[k] = cube(s[k]);
        quad_mds(s, T + QUAD_BIAS + 4 * r);
    }
    // S-boxes of the 4th full round; everything below is affine in (c_0..c_15, q_0..q_19)
    u32 c[4], w[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = cube(s[k]);
    quad_gather(c, w);
    u64 Y[5], F[4];
    static_for<0, 5>([&](auto M) {
        constexpr int m = decltype(M)::value;
        u64 acc = 0;
        static_for<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (i >= 4 && (i - 4) % 3 == 0) acc = fold32(acc);
            acc += (u64)w[i >> 2][i & 3] * T[QUAD_YA + 16 * m + i];
        });
        Y[m] = fold32(acc);  // 16 = 4 + 3 * 4 products: no room left
    });
    static_for<0, 4>([&](auto A) {
        constexpr int a = decltype(A)::value;
        u64 acc = 0;
        static_for<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (i >= 4 && (i - 4) % 3 == 0) acc = fold32(acc);
            acc += (u64)w[i >> 2][i & 3] * T[QUAD_FA + 16 * a + i];
        });
        F[a] = fold32(acc);
    });
    // 20 partial rounds: y_r is owned by class r & 3, slot r >> 2
    static_for<0, 20>([&](auto RR) {
        constexpr int r = decltype(RR)::value;
        constexpr int m = r >> 2;
        const u32 y = add(reduce(fold32(Y[m])), T[QUAD_YC + m]);  // meaningful in the owner class only
        const u32 qv = quad_from<(r & 3)>(cube(y));
        if (r % 3 == 0) {  // one product per round since the last fold: room for three
            static_for<m, 5>([&](auto MM) { Y[decltype(MM)::value] = fold32(Y[decltype(MM)::value]); });
            static_for<0, 4>([&](auto A) { F[decltype(A)::value] = fold32(F[decltype(A)::value]); });
        }
        // y_{q + 4 m'} with q + 4 m' > r: slots m' > m in every class, slot m in the classes behind the owner (the table holds
        // zeros where the y has already been consumed)
        static_for<m, 5>([&](auto MM) {
            constexpr int mm = decltype(MM)::value;
            Y[mm] += (u64)qv * T[QUAD_YQ + 20 * mm + r];
        });
        static_for<0, 4>([&](auto A) {
            constexpr int a = decltype(A)::value;
            F[a] += (u64)qv * T[QUAD_FQ + 20 * a + r];
        });
    });
#pragma unroll
    for (int a = 0; a < 4; a++) s[a] = add(reduce(fold32(F[a])), T[QUAD_FC + a]);
    // 4 terminal full rounds
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) s[k] = cube(s[k]);
        quad_mds(s, r < 3 ? T + QUAD_BIAS + 4 * (3 + r) : nullptr);
    }
}
#endif

// Host: the 4 x QUAD_STRIDE table
inline void lm_quad_table_build(u32* out /* QUAD_TAB_WORDS */) {
    const PoseidonLinearHash& L = kPoseidonLinearHash;
    for (u32 q = 0; q < 4; q++) {
        u32* T = out + q * QUAD_STRIDE;
        for (u32 i = 0; i < QUAD_STRIDE; i++) T[i] = 0;
        for (u32 k = 0; k < 4; k++) T[QUAD_RC0 + k] = kPoseidonHost.rc_init[0][4 * q + k];
        // MDS applications: initial rounds 0,1,2 add rc_init[1..3]; terminal rounds 0,1,2 add rc_term[1..3]; the last adds nothing
        for (u32 m = 0; m < 6; m++)
            for (u32 k = 0; k < 4; k++)
                T[QUAD_BIAS + 4 * m + k] = m < 3 ? kPoseidonHost.rc_init[m + 1][4 * q + k] : kPoseidonHost.rc_term[m - 3 + 1][4 * q + k];
        auto word = [&](u32 i) { return 4 * ((q + (i >> 2)) & 3) + (i & 3); };  // the state word behind w[i >> 2][i & 3]
        for (u32 m = 0; m < 5; m++) {
            const u32 r = q + 4 * m;  // this class's m-th y
            for (u32 i = 0; i < 16; i++) T[QUAD_YA + 16 * m + i] = L.y[r][word(i)];
            T[QUAD_YC + m] = L.y[r][36];
            for (u32 k = 0; k < 20; k++) T[QUAD_YQ + 20 * m + k] = k < r ? L.y[r][16 + k] : 0;
        }
        for (u32 a = 0; a < 4; a++) {
            const u32 o = 4 * q + a;
            for (u32 i = 0; i < 16; i++) T[QUAD_FA + 16 * a + i] = L.fin[o][word(i)];
            T[QUAD_FC + a] = (u32)(((u64)L.fin[o][36] + kPoseidonHost.rc_term[0][o]) % P);
            for (u32 k = 0; k < 20; k++) T[QUAD_FQ + 20 * a + k] = L.fin[o][16 + k];
        }
    }
}

}  // namespace kb
