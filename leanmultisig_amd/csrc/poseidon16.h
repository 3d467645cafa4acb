// Poseidon1-16 over KoalaBear (x^3, 4 + 20 + 4 rounds, circulant MDS) for gfx950 lanes and for the host transcript.
// Same permutation as the reference (crates/backend/koala-bear/src/poseidon1_koalabear_16.rs:873-912).  The partial block
// is evaluated through affine forms derived by gen_poseidon_consts.py (the sparse-matrix form of the reference, :399-480,
// is kept for the trace generator, which must emit the per-round lane-0 values); the full rounds use the small-integer
// circulant directly (entries <= 101, row sum 371), accumulated in 64 bits and folded with 2^31 = 2^24 - 1 (mod p).
// One permutation = one lane; the 16-word state lives in VGPRs, every constant is an instruction literal.
#pragma once
#include "kb.h"

namespace kb {

struct PoseidonConsts {
    u32 rc_init[4][16];
    u32 rc_term[4][16];
    u32 dm[16][16];   // D * MDS (fused linear layer of the 4th full round and the partial-block entry)
    u32 dbias[16];    // D * first_rc
    u32 dmat[16][16]; // D alone (Poseidon AIR: the state is re-based on committed columns before the partial block)
    u32 prow[20][16]; // sparse first row per partial round
    u32 pcol[20][16]; // sparse first column (rows 1..15) per partial round
    u32 pscalar[20];  // lane-0 constant added after the S-box of partial round r (r < 19)
};

// constexpr copy: with compile-time indices every entry becomes an instruction literal on the device (no scalar loads to
// hoist and spill); the __constant__ copy below serves dynamically indexed uses (rolled loops).
static constexpr PoseidonConsts kPoseidonHost =
#include "poseidon16_consts.inc"
    ;

// Partial block as affine forms (gen_poseidon_consts.py::linearise, hashing variant).  Lanes 1..15 see no S-box inside the
// block, so with c = S-box outputs of the 4th full round and q_r = lane-0 S-box output of partial round r, the value cubed in
// round r and the state leaving the block are affine in u = (c_0..c_15, q_0..q_19, 1); the MDS of the 4th full round is
// folded in.  20 + 16 dot products with delayed reduction replace the MDS, the dense entry map and 20 sparse rounds of a
// mul+reduce per term: 4.9 vs 3.7 G perm/s on MI355X (tools/ubench/perm_variants.hip).
struct PoseidonLinearHash {
    u32 y[20][37];    // y_r = y[r][0..16+r) . u + y[r][36],  q_r = y_r^3
    u32 fin[16][37];  // state entering the terminal full rounds
};
static constexpr PoseidonLinearHash kPoseidonLinearHash =
#include "poseidon16_linear_hash.inc"
    ;

#if defined(__HIPCC__)
static __constant__ PoseidonConsts kPoseidonDev =
#include "poseidon16_consts.inc"
    ;
#endif

KB_HD const PoseidonConsts& poseidon_consts() {
#if defined(__HIP_DEVICE_COMPILE__)
    return kPoseidonDev;
#else
    return kPoseidonHost;
#endif
}

// On the device small compile-time multipliers are hidden from the optimiser (an empty asm on an SGPR): otherwise x*1,
// x*2 and x*(2^24-1) in 64 bits are strength-reduced to v_lshl_add_u64 / shift-subtract chains, and v_lshl_add_u64 issues
// at about a third of the v_mad_u64_u32 rate on gfx950 (tools/ubench/int_rates.hip, mds_variants.hip: +12% on the MDS).
KB_HD u32 opaque_const(u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(c));
#endif
    return c;
}

// x < 2^43  ->  [0, p).   2^31 = 2^24 - 1 (mod p), applied twice.
KB_HD u32 reduce40(u64 s) {
    const u32 m24 = opaque_const(0x00ffffffu);
    u32 a = (u32)(s >> 31);             // < 2^12
    u32 b = (u32)s & 0x7fffffffu;
    u64 r1 = (u64)a * m24 + b;          // < 2^36 + 2^31
    u32 a2 = (u32)(r1 >> 31);           // < 2^6
    u32 b2 = (u32)r1 & 0x7fffffffu;
    u32 r2 = b2 + a2 * 0x00ffffffu;     // < 2^31 + 2^30 < 2p
    return umin(r2, r2 - P);
}

// The same for a value that is squared next: + rc (a round constant < p), NOT canonical: the result is below 2^31 + 34 * 2^24 < 1.28 p,
// and a Montgomery product a * b only needs a * b < 2^32 p = 2.0158 p^2 (cube: a^2 < 1.63 p^2, then [0, p) * a < 1.28 p^2).  The constant
// costs one 32-bit addition here (against a 64-bit addition in the accumulator or a modular addition behind the reduction) and the
// conditional subtraction at the end is gone: 7 instead of 12.7 issue units per MDS output.
KB_HD u32 reduce40_weak(u64 s, u32 rc) {
    const u32 m24 = opaque_const(0x00ffffffu);
    u32 a = (u32)(s >> 31);                 // < 2^12
    u32 b = ((u32)s & 0x7fffffffu) + rc;    // < 2^31 + p < 2^32
    u64 r1 = (u64)a * m24 + b;              // < 2^36 + 2^32
    u32 a2 = (u32)(r1 >> 31);               // <= 33
    u32 b2 = (u32)r1 & 0x7fffffffu;
    return b2 + a2 * 0x00ffffffu;           // < 2^31 + 34 * 2^24
}

// 64-bit addition as a carry pair.  For `a + b` on 64-bit values clang emits v_lshl_add_u64 on gfx950, which issues in 7.4 cycles per
// wave64 (tools/ubench/int_rates.hip) against 2 + 2 for v_add_co_u32 / v_addc_co_u32 (a 64-bit subtraction already becomes the pair).
KB_HD u64 add64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo, hi;
    asm("v_add_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, %4, %5, vcc"
        : "=&v"(lo), "=v"(hi)
        : "v"((u32)a), "v"((u32)b), "v"((u32)(a >> 32)), "v"((u32)(b >> 32))
        : "vcc");
    return ((u64)hi << 32) | lo;
#else
    return a + b;
#endif
}

// s <- C * s with C[i][j] = col[(i - j) mod 16], col = {1,3,13,22,67,2,15,63,101,1,2,17,11,1,51,1}
// (poseidon1_koalabear_16.rs:22,580-581).  Plain small integers act directly on Montgomery-form values.
// bias: optional per-lane constants added before the reduction (the round constants of the NEXT round ride along for
// free in the 64-bit accumulator instead of costing a modular addition each)
//
// One CRT split of the cyclic convolution, over the INTEGERS (x^16 - 1 = (x^8 - 1)(x^8 + 1)): with s = (lo, hi), c = (c_lo, c_hi)
//     P = (c_lo + c_hi) * (lo + hi)  mod x^8 - 1   (8 x 8 cyclic,     64 unsigned multiply-adds)
//     M = (c_lo - c_hi) * (lo - hi)  mod x^8 + 1   (8 x 8 negacyclic, 64 signed multiply-adds)
//     out_lo = (P + M) / 2,  out_hi = (P - M) / 2  — exact: P = out_lo + out_hi and M = out_lo - out_hi as integers, so the sums
//     are even.  128 v_mad_{u,i}64_{u,i}32 + 16 x (64-bit add, shift) instead of 256 multiply-adds: the 4-cycle multiply-adds are
//     63 % of the permutation's issue cycles.  lo + hi < 2^32 fits a u32, |lo - hi| < 2^31 an i32; every
//     accumulator stays below 2^43.  The optional bias rides along: (b_lo + b_hi) enters P, (b_lo - b_hi) enters M.
// MODE 0: canonical outputs.  1: + bias, canonical outputs (the constants start the accumulators).  2: + bias, outputs for an S-box
// layer (reduce40_weak: below 1.28 p, the constant added on the way).
template <int MODE>
KB_HD void mds_circ16_impl(u32 s[16], const u32* bias) {
    constexpr bool WITH_BIAS = MODE == 1;
    // c_lo + c_hi and c_lo - c_hi of col = {1,3,13,22,67,2,15,63 | 101,1,2,17,11,1,51,1}
    const u32 CP[8] = {opaque_const(102), opaque_const(4), opaque_const(15), opaque_const(39), opaque_const(78), opaque_const(3), opaque_const(66), opaque_const(64)};
    // (both signs as their own scalar constants: the wrap-around terms of the negacyclic product are multiply-adds too)
    const int32_t CM[8] = {(int32_t)opaque_const((u32)-100), (int32_t)opaque_const(2), (int32_t)opaque_const(11), (int32_t)opaque_const(5),
                           (int32_t)opaque_const(56), (int32_t)opaque_const(1), (int32_t)opaque_const((u32)-36), (int32_t)opaque_const(62)};
    const int32_t CN[8] = {(int32_t)opaque_const(100), (int32_t)opaque_const((u32)-2), (int32_t)opaque_const((u32)-11), (int32_t)opaque_const((u32)-5),
                           (int32_t)opaque_const((u32)-56), (int32_t)opaque_const((u32)-1), (int32_t)opaque_const(36), (int32_t)opaque_const((u32)-62)};
    u32 sp[8];
    int32_t sm[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        sp[j] = s[j] + s[j + 8];
        sm[j] = (int32_t)(s[j] - s[j + 8]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 accp = WITH_BIAS ? (u64)bias[i] + bias[i + 8] : 0;
        int64_t accm = WITH_BIAS ? (int64_t)bias[i] - (int64_t)bias[i + 8] : 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            accp += (u64)sp[j] * CP[(8 + i - j) & 7];
            accm += (int64_t)sm[j] * (j <= i ? CM[i - j] : CN[8 + i - j]);
        }
        const u64 sum = add64(accp, (u64)accm) >> 1, dif = (u64)((int64_t)accp - accm) >> 1;
        if constexpr (MODE == 2) {
            s[i] = reduce40_weak(sum, bias[i]);
            s[i + 8] = reduce40_weak(dif, bias[i + 8]);
        } else {
            s[i] = reduce40(sum);
            s[i + 8] = reduce40(dif);
        }
    }
}
KB_HD void mds_circ16(u32 s[16]) { mds_circ16_impl<0>(s, nullptr); }
KB_HD void mds_circ16_bias(u32 s[16], const u32 bias[16]) { mds_circ16_impl<1>(s, bias); }
// the linear layer in front of an S-box layer: C * s + the next round's constants, every output below 1.28 p (fit to be cubed)
KB_HD void mds_circ16_to_sbox(u32 s[16], const u32 bias[16]) { mds_circ16_impl<2>(s, bias); }

// ---- the partial block on CENTRED values -------------------------------------------------------------------------------------------
// The affine forms of the partial block are 36 dot products of up to 36 terms, constant x value.  With both factors in [0, p) four
// products fill a u64 and every fourth costs a fold (fold32: a multiply-add and, for the zero-extended addend, a move).  With both
// factors CENTRED — constants as representatives in (-p/2, p/2), values as v - (p-1)/2 — a product is below 2^60 and EIGHT fit an i64
// between folds (8 * 0.985 * 2^60 + 2^56 < 2^63): 153 folds per permutation instead of 360.  The shift of the values is a constant per
// dot product, (p-1)/2 * sum(c), folded into the affine constant at compile time; the last fold flips the sign bit of the high word
// (hi + 2^31 >= 0: an unsigned multiply-add, result < 2^57) and its 2^31 * 2^32 joins the same constant.
static constexpr u32 HALF_P = (P - 1) / 2;
constexpr int32_t centred(u32 c) { return c > HALF_P ? (int32_t)((int64_t)c - (int64_t)P) : (int32_t)c; }
constexpr u32 mulmod_c(u64 a, u64 b) { return (u32)((a % P) * (b % P) % P); }
constexpr u32 powmod_c(u32 a, u32 e) {
    u32 r = 1;
    while (e) {
        if (e & 1) r = mulmod_c(r, a);
        a = mulmod_c(a, a);
        e >>= 1;
    }
    return r;
}
static constexpr u32 RINV_C = powmod_c(ONE, P - 2);  // 2^-32 mod p (plain integers)
// what reduce(x) of partial_dot lacks of the Montgomery value of sum c_j u_j: ((p-1)/2 * sum c_j - 2^31 * 2^32) * 2^-32
template <int N>
constexpr u32 centred_shift(const u32 (&row)[37]) {
    u64 sum = 0;
    for (int j = 0; j < N; j++) sum = (sum + row[j]) % P;
    const u32 k = (u32)((mulmod_c(HALF_P, sum) + (u64)P - mulmod_c(1u << 31, ONE)) % P);
    return mulmod_c(k, RINV_C);
}
KB_HD int64_t fold32s(int64_t x) { return (int64_t)(int32_t)((u64)x >> 32) * (int64_t)ONE + (int64_t)(u32)x; }

// 16-term dot product with delayed reduction (4 products per fold).
KB_HD u32 dot16(const u32 s[16], const u32 c[16]) { return dot_n<16>(s, c); }

// sum_{j < N} row[j] * (uc[j] + (p-1)/2) as a Montgomery product, up to the compile-time constant centred_shift<N>(row)
template <int N, int ROW, bool FIN>
KB_HD u32 partial_dot(const int32_t (&uc)[36]) {
    int64_t acc = 0;
    static_for<0, N>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int32_t c = centred(FIN ? kPoseidonLinearHash.fin[ROW][j] : kPoseidonLinearHash.y[ROW][j]);
        if constexpr (j > 0 && j % 8 == 0) acc = fold32s(acc);
        acc += (int64_t)uc[j] * c;
    });
    const u32 hb = (u32)((u64)acc >> 32) ^ 0x80000000u;  // hi + 2^31: the accumulator as a non-negative value
    return reduce((u64)hb * ONE + (u32)acc);             // < 2^57 + 2^32
}

KB_HD void poseidon16_permute(u32 s[16]) {
    // 3 plain initial full rounds; the round constants of round r + 1 are added inside the MDS of round r
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        s[i] = add(s[i], kPoseidonHost.rc_init[0][i]);
    });
    static_for<0, 3>([&](auto R) {
        constexpr int r = decltype(R)::value;
        static_for<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            s[i] = cube(s[i]);
        });
        mds_circ16_to_sbox(s, kPoseidonHost.rc_init[r + 1]);
    });
    // S-boxes of the 4th full round, then the partial block through its affine forms, on centred values
    int32_t uc[36];
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        uc[i] = (int32_t)(cube(s[i]) - HALF_P);
    });
    static_for<0, 20>([&](auto R) {
        constexpr int r = decltype(R)::value;
        constexpr u32 c = (u32)(((u64)kPoseidonLinearHash.y[r][36] + centred_shift<16 + r>(kPoseidonLinearHash.y[r])) % P);
        uc[16 + r] = (int32_t)(cube(add(partial_dot<16 + r, r, false>(uc), c)) - HALF_P);
    });
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr u32 c = (u32)(((u64)kPoseidonLinearHash.fin[i][36] + kPoseidonHost.rc_term[0][i] + centred_shift<36>(kPoseidonLinearHash.fin[i])) % P);
        s[i] = add(partial_dot<36, i, true>(uc), c);  // (+ the first terminal constant)
    });
    // 4 terminal full rounds
    static_for<0, 4>([&](auto R) {
        constexpr int r = decltype(R)::value;
        static_for<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            s[i] = cube(s[i]);
        });
        if constexpr (r < 3)
            mds_circ16_to_sbox(s, kPoseidonHost.rc_term[r + 1]);
        else
            mds_circ16(s);
    });
}

// Proof-of-work variant (fiat-shamir/src/challenger.rs: the grinding permutation): the state is (capacity[8], w, 0 x 7) with
// only w varying between candidates, and only word 8 of the result is looked at.
//   * first full round: the S-box outputs of the 15 fixed words and their share of the MDS (+ the next round's constants) are
//     the same for every candidate: `base` (poseidon16_pow_base, computed once on the host); a candidate adds its own column;
//   * last full round: one row of the MDS instead of sixteen.
// ~12 % fewer instructions than poseidon16_permute; identical value of word 8.
KB_HD void poseidon16_pow_base(const u32 cap[8], u32 base[16]) {
    const u32 C[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
    u32 u[16];
    for (int j = 0; j < 16; j++) u[j] = j == 8 ? 0u : cube(add(j < 8 ? cap[j] : 0u, kPoseidonHost.rc_init[0][j]));
    for (int i = 0; i < 16; i++) {
        u64 acc = kPoseidonHost.rc_init[1][i];
        for (int j = 0; j < 16; j++) acc += (u64)u[j] * C[(16 + i - j) & 15];
        base[i] = (u32)(acc % P);
    }
}
KB_HD u32 poseidon16_pow_word8(const u32 base[16], u32 w) {
    const u32 c1 = opaque_const(1), c2 = opaque_const(2), c3 = opaque_const(3), c13 = opaque_const(13);
    const u32 c22 = opaque_const(22), c67 = opaque_const(67), c15 = opaque_const(15), c63 = opaque_const(63);
    const u32 c101 = opaque_const(101), c17 = opaque_const(17), c11 = opaque_const(11), c51 = opaque_const(51);
    const u32 C[16] = {c1, c3, c13, c22, c67, c2, c15, c63, c101, c1, c2, c17, c11, c1, c51, c1};
    u32 s[16];
    const u32 t = cube(add(w, kPoseidonHost.rc_init[0][8]));
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        s[i] = reduce40((u64)t * C[(16 + i - 8) & 15] + base[i]);
    });
    static_for<1, 3>([&](auto R) {
        constexpr int r = decltype(R)::value;
        static_for<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            s[i] = cube(s[i]);
        });
        mds_circ16_bias(s, kPoseidonHost.rc_init[r + 1]);
    });
    u32 u[36];
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        u[i] = cube(s[i]);
    });
    static_for<0, 20>([&](auto R) {
        constexpr int r = decltype(R)::value;
        u[16 + r] = cube(add(dot_n<16 + r>(u, kPoseidonLinearHash.y[r]), kPoseidonLinearHash.y[r][36]));
    });
    static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr u32 c = (u32)(((u64)kPoseidonLinearHash.fin[i][36] + kPoseidonHost.rc_term[0][i]) % P);
        s[i] = add(dot_n<36>(u, kPoseidonLinearHash.fin[i]), c);
    });
    static_for<0, 3>([&](auto R) {
        constexpr int r = decltype(R)::value;
        static_for<0, 16>([&](auto I) {
            constexpr int i = decltype(I)::value;
            s[i] = cube(s[i]);
        });
        mds_circ16_bias(s, kPoseidonHost.rc_term[r + 1]);
    });
    u64 acc = 0;
    static_for<0, 16>([&](auto J) {
        constexpr int j = decltype(J)::value;
        acc += (u64)cube(s[j]) * C[(16 + 8 - j) & 15];
    });
    return reduce40(acc);
}

// compression mode: perm(x) + x (poseidon1_koalabear_16.rs:1018-1030)
KB_HD void poseidon16_compress(u32 s[16]) {
    u32 in[16];
#pragma unroll
    for (int i = 0; i < 16; i++) in[i] = s[i];
    poseidon16_permute(s);
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = add(s[i], in[i]);
}

}  // namespace kb
