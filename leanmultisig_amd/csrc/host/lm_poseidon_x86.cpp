// Host-side Poseidon1-16 for the transcript (Fiat-Shamir challenger, fiat-shamir/src/challenger.rs) and the verifier.
// A proof performs ~1300 strictly sequential host permutations (every sumcheck round: observe the round polynomial, sample
// the challenge) between device launches, so their LATENCY is on the proof's critical path: ~1.1 us each with the scalar
// code of poseidon16.h (throughput bound, ~4000 64-bit multiplies).  This file holds an AVX-512 (F + IFMA) version:
//   * full rounds: the 16 state words are one zmm; the S-box is two 16-lane Montgomery multiplications; the circulant MDS
//     is 16 lane rotations x small constants accumulated in 64-bit lanes (even / odd outputs), reduced with 2^31 = 2^24 - 1;
//   * partial block through the affine forms of poseidon16.h (gen_poseidon_consts.py::linearise): the 16-input parts of
//     the 20 cubed values are a 20 x 16 matrix-vector product with 52-bit multiply-accumulates (vpmadd52: 16 products of
//     31-bit values accumulate in a 64-bit lane without overflow), the chain q_0 -> q_19 (one multiplication and one cube per
//     round on the critical path) is scalar, the state leaving the block is a 16 x 36 matrix-vector product again.
// Same function as kb::poseidon16_permute, bit for bit (tests/test_host_poseidon.py); used when the CPU has the
// instructions (lmh_poseidon_backend() says which one runs), otherwise the scalar code.
// (the build compiles every source as HIP: this file is host-only, the device pass sees nothing)
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>

#include <cstdlib>
#include <cstring>

#include "../poseidon16.h"
#include "lm_host_internal.h"

namespace lmh {
namespace {
using kb::P;

#define LM_AVX512 __attribute__((target("avx512f,avx512dq,avx512bw,avx512vl,avx512ifma")))

struct alignas(64) X86Tables {
    u32 rc0[16];              // constants added before the first S-box layer
    u64 bias_e[7][8];         // per MDS application: round constants of the following round (even / odd outputs); 0 for the last
    u64 bias_o[7][8];
    u64 ya[16][24];           // ya[i][r] = y[r][i]            (r < 20)
    u64 fin[36][16];          // fin[j][i] = fin_table[i][j]
    u32 fin_c[16];            // fin_table[i][36] + first terminal round constant
    // (everything above is loaded with aligned 64-byte loads: keep the sizes multiples of 64)
    u32 yq[20][20];           // yq[r][k] = y[r][16 + k]       (k < r)
    u32 yc[20];               // y[r][36]
};
X86Tables g_tab;

void build_tables() {
    const auto& C = kb::kPoseidonHost;
    const auto& L = kb::kPoseidonLinearHash;
    memset(&g_tab, 0, sizeof g_tab);
    for (int i = 0; i < 16; i++) g_tab.rc0[i] = C.rc_init[0][i];
    // MDS applications in order: initial rounds 0, 1, 2 (bias = rc_init[r + 1]; round 2's feeds the 4th round's S-boxes),
    // terminal rounds 0, 1, 2 (bias = rc_term[r + 1]), terminal round 3 (no bias)
    for (int m = 0; m < 7; m++)
        for (int i = 0; i < 16; i++) {
            const u32 b = m < 3 ? C.rc_init[m + 1][i] : (m < 6 ? C.rc_term[m - 3 + 1][i] : 0u);
            (i & 1 ? g_tab.bias_o : g_tab.bias_e)[m][i >> 1] = b;
        }
    for (int r = 0; r < 20; r++) {
        for (int i = 0; i < 16; i++) g_tab.ya[i][r] = L.y[r][i];
        for (int k = 0; k < r; k++) g_tab.yq[r][k] = L.y[r][16 + k];
        g_tab.yc[r] = L.y[r][36];
    }
    for (int i = 0; i < 16; i++) {
        for (int j = 0; j < 36; j++) g_tab.fin[j][i] = L.fin[i][j];
        g_tab.fin_c[i] = (u32)(((u64)L.fin[i][36] + C.rc_term[0][i]) % P);
    }
}

LM_AVX512 inline __m512i add_mod(__m512i a, __m512i b) {
    const __m512i s = _mm512_add_epi32(a, b);
    return _mm512_min_epu32(s, _mm512_sub_epi32(s, _mm512_set1_epi32((int)P)));
}
// 16-lane Montgomery multiplication, inputs and output in [0, p)
LM_AVX512 inline __m512i mul_mod(__m512i a, __m512i b) {
    const __m512i vp = _mm512_set1_epi32((int)P), vmu = _mm512_set1_epi32((int)kb::MU);
    const __m512i a_o = _mm512_srli_epi64(a, 32), b_o = _mm512_srli_epi64(b, 32);
    const __m512i pe = _mm512_mul_epu32(a, b), po = _mm512_mul_epu32(a_o, b_o);
    __m512i te = _mm512_mul_epu32(pe, vmu), to = _mm512_mul_epu32(po, vmu);   // low words = x_lo * MU
    // (opaque: otherwise clang sees that only the low words are used and turns these into 64-bit vpmullq — 3 uops, 15 cycles)
    asm("" : "+v"(te), "+v"(to));
    const __m512i ue = _mm512_mul_epu32(te, vp), uo = _mm512_mul_epu32(to, vp);     // (t p): low word equals x_lo
    // d = x_hi - (t p)_hi in the high words
    const __m512i de = _mm512_sub_epi64(pe, ue), dox = _mm512_sub_epi64(po, uo);
    const __m512i d = _mm512_mask_blend_epi32(0xAAAA, _mm512_srli_epi64(de, 32), dox);  // even lanes <- hi(de), odd lanes <- hi(do)
    return _mm512_min_epu32(d, _mm512_add_epi32(d, vp));
}
LM_AVX512 inline __m512i cube_mod(__m512i a) { return mul_mod(mul_mod(a, a), a); }

// x < 2^43 per 64-bit lane -> value < 2^31 + 2^29 congruent mod p (2^31 = 2^24 - 1)
LM_AVX512 inline __m512i shrink43(__m512i x) {
    const __m512i m31 = _mm512_set1_epi64(0x7fffffff);
    __m512i a = _mm512_srli_epi64(x, 31), b = _mm512_and_si512(x, m31);
    __m512i r = _mm512_sub_epi64(_mm512_add_epi64(b, _mm512_slli_epi64(a, 24)), a);
    a = _mm512_srli_epi64(r, 31);
    b = _mm512_and_si512(r, m31);
    return _mm512_sub_epi64(_mm512_add_epi64(b, _mm512_slli_epi64(a, 24)), a);
}

// s <- circ(col) s + bias, col = {1,3,13,22,67,2,15,63,101,1,2,17,11,1,51,1} (poseidon16.h mds_circ16)
template <int K>
LM_AVX512 inline void mds_term(__m512i s, __m512i& e, __m512i& o) {
    constexpr u32 C[16] = {1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1};
    const __m512i r = K == 0 ? s : _mm512_alignr_epi32(s, s, (16 - K) & 15);  // r[i] = s[i - K]
    e = _mm512_add_epi64(e, _mm512_mul_epu32(r, _mm512_set1_epi64(C[K])));
    o = _mm512_add_epi64(o, _mm512_mul_epu32(r, _mm512_set1_epi64(C[(K + 1) & 15])));
}
LM_AVX512 inline __m512i mds(__m512i s, int which) {
    __m512i e0 = _mm512_load_si512(g_tab.bias_e[which]), o0 = _mm512_load_si512(g_tab.bias_o[which]);
    __m512i e1 = _mm512_setzero_si512(), o1 = _mm512_setzero_si512();
    mds_term<0>(s, e0, o0);
    mds_term<1>(s, e1, o1);
    mds_term<2>(s, e0, o0);
    mds_term<3>(s, e1, o1);
    mds_term<4>(s, e0, o0);
    mds_term<5>(s, e1, o1);
    mds_term<6>(s, e0, o0);
    mds_term<7>(s, e1, o1);
    mds_term<8>(s, e0, o0);
    mds_term<9>(s, e1, o1);
    mds_term<10>(s, e0, o0);
    mds_term<11>(s, e1, o1);
    mds_term<12>(s, e0, o0);
    mds_term<13>(s, e1, o1);
    mds_term<14>(s, e0, o0);
    mds_term<15>(s, e1, o1);
    const __m512i e = shrink43(_mm512_add_epi64(e0, e1)), o = shrink43(_mm512_add_epi64(o0, o1));
    const __m512i v = _mm512_or_si512(e, _mm512_slli_epi64(o, 32));
    return _mm512_min_epu32(v, _mm512_sub_epi32(v, _mm512_set1_epi32((int)P)));
}

// (hi, lo) accumulators of vpmadd52 over products of Montgomery-form values: V = hi 2^52 + lo, lo < 2^57, hi < 2^16.
// Returns V / 2^32 mod p in [0, p) per 64-bit lane (the Montgomery form of the sum of products).
LM_AVX512 inline __m512i reduce52(__m512i hi, __m512i lo) {
    const __m512i vp = _mm512_set1_epi64(P), vmu = _mm512_set1_epi64(kb::MU);
    const __m512i l_hi = _mm512_srli_epi64(lo, 32);
    __m512i m = _mm512_mul_epu32(lo, vmu);
    asm("" : "+v"(m));  // (see mul_mod)
    const __m512i t = _mm512_srli_epi64(_mm512_mul_epu32(m, vp), 32);  // (lo_lo - m p) / 2^32 = -t, t < p
    // W = hi 2^20 + l_hi + p - t  < 2^36 + 2^25 + 2^31
    const __m512i w = _mm512_sub_epi64(_mm512_add_epi64(_mm512_add_epi64(_mm512_slli_epi64(hi, 20), l_hi), vp), t);
    const __m512i r = shrink43(w);
    return _mm512_min_epu64(r, _mm512_sub_epi64(r, vp));
}

LM_AVX512 void permute_avx512(u32 st[16]) {
    __m512i s = _mm512_loadu_si512(st);
    s = add_mod(s, _mm512_load_si512(g_tab.rc0));
    for (int r = 0; r < 3; r++) s = mds(cube_mod(s), r);
    // partial block: u = (c_0..c_15, q_0..q_19)
    alignas(64) u64 u[36];
    {
        const __m512i c = cube_mod(s);
        _mm512_store_si512(u, _mm512_cvtepu32_epi64(_mm512_castsi512_si256(c)));
        _mm512_store_si512(u + 8, _mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(c, 1)));
    }
    alignas(64) u64 ycv[24];
    {
        __m512i h0 = _mm512_setzero_si512(), h1 = h0, h2 = h0, l0 = h0, l1 = h0, l2 = h0;
        for (int i = 0; i < 16; i++) {
            const __m512i b = _mm512_set1_epi64((long long)u[i]);
            const __m512i c0 = _mm512_load_si512(g_tab.ya[i]), c1 = _mm512_load_si512(g_tab.ya[i] + 8), c2 = _mm512_load_si512(g_tab.ya[i] + 16);
            l0 = _mm512_madd52lo_epu64(l0, b, c0);
            h0 = _mm512_madd52hi_epu64(h0, b, c0);
            l1 = _mm512_madd52lo_epu64(l1, b, c1);
            h1 = _mm512_madd52hi_epu64(h1, b, c1);
            l2 = _mm512_madd52lo_epu64(l2, b, c2);
            h2 = _mm512_madd52hi_epu64(h2, b, c2);
        }
        _mm512_store_si512(ycv, reduce52(h0, l0));
        _mm512_store_si512(ycv + 8, reduce52(h1, l1));
        _mm512_store_si512(ycv + 16, reduce52(h2, l2));
    }
    {
        u32 q[20];
        kb::static_for<0, 20>([&](auto R) {
            constexpr int r = decltype(R)::value;
            // B = constant + 16-input part + the products with q_0 .. q_{r-2} (off the critical path), as an unreduced
            // Montgomery numerator < 2^58; the product with q_{r-1} and ONE reduction follow
            u64 x = ((u64)kb::add((u32)ycv[r], g_tab.yc[r])) << 32;   // value * 2^32: reduces to the value itself
            x = kb::fold32(x);
            int room = 3;
            kb::static_for<0, (r >= 1 ? r - 1 : 0)>([&](auto K) {
                constexpr int k = decltype(K)::value;
                if (room == 0) {
                    x = kb::fold32(x);
                    room = 3;
                }
                x += (u64)q[k] * g_tab.yq[r][k];
                room--;
            });
            if (room == 0) x = kb::fold32(x);
            if constexpr (r >= 1) x += (u64)q[r - 1] * g_tab.yq[r][r - 1];
            const u32 y = kb::reduce(kb::fold32(x));
            q[r] = kb::cube(y);
            u[16 + r] = q[r];
        });
    }
    {
        __m512i h0 = _mm512_setzero_si512(), h1 = h0, l0 = h0, l1 = h0;
        __m512i h2 = h0, h3 = h0, l2 = h0, l3 = h0;  // two chains per output half
        for (int j = 0; j < 36; j += 2) {
            const __m512i b0 = _mm512_set1_epi64((long long)u[j]), b1 = _mm512_set1_epi64((long long)u[j + 1]);
            const __m512i c0 = _mm512_load_si512(g_tab.fin[j]), c1 = _mm512_load_si512(g_tab.fin[j] + 8);
            const __m512i d0 = _mm512_load_si512(g_tab.fin[j + 1]), d1 = _mm512_load_si512(g_tab.fin[j + 1] + 8);
            l0 = _mm512_madd52lo_epu64(l0, b0, c0);
            h0 = _mm512_madd52hi_epu64(h0, b0, c0);
            l1 = _mm512_madd52lo_epu64(l1, b0, c1);
            h1 = _mm512_madd52hi_epu64(h1, b0, c1);
            l2 = _mm512_madd52lo_epu64(l2, b1, d0);
            h2 = _mm512_madd52hi_epu64(h2, b1, d0);
            l3 = _mm512_madd52lo_epu64(l3, b1, d1);
            h3 = _mm512_madd52hi_epu64(h3, b1, d1);
        }
        const __m512i r0 = reduce52(_mm512_add_epi64(h0, h2), _mm512_add_epi64(l0, l2));
        const __m512i r1 = reduce52(_mm512_add_epi64(h1, h3), _mm512_add_epi64(l1, l3));
        s = _mm512_inserti64x4(_mm512_castsi256_si512(_mm512_cvtepi64_epi32(r0)), _mm512_cvtepi64_epi32(r1), 1);
        s = add_mod(s, _mm512_load_si512(g_tab.fin_c));
    }
    for (int r = 0; r < 4; r++) s = mds(cube_mod(s), 3 + r);
    _mm512_storeu_si512(st, s);
}

void permute_scalar(u32 st[16]) { kb::poseidon16_permute(st); }

using PermFn = void (*)(u32*);
const char* g_backend = "scalar";
PermFn pick() {
    __builtin_cpu_init();  // this runs from a static constructor of a shared library
    if (getenv("LM_HOST_POSEIDON_SCALAR") == nullptr && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
        __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512ifma")) {
        build_tables();
        g_backend = "avx512-ifma";
        return permute_avx512;
    }
    return permute_scalar;
}
PermFn g_perm = pick();
}  // namespace

void host_permute(u32 state[16]) { g_perm(state); }
}  // namespace lmh

extern "C" const char* lmh_poseidon_backend(void) { return lmh::g_backend; }
extern "C" void lmh_poseidon16_permute(uint32_t state[16]) { lmh::g_perm(state); }
extern "C" void lmh_poseidon16_permute_scalar(uint32_t state[16]) { kb::poseidon16_permute(state); }
#endif  // !__HIP_DEVICE_COMPILE__
