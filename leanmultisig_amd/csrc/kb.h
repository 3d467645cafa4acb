// KoalaBear field (p = 2^31 - 2^24 + 1) and its quintic extension F_p[X]/(X^5 + X^2 - 1) for gfx950 device code
// and for the host-side transcript.  Values are Montgomery form, R = 2^32, exactly the in-memory representation of
// the reference (crates/backend/koala-bear/src/monty_31/monty_31.rs:33-41, koala_bear.rs:22-26), so buffers can be
// shared with a Rust caller without conversion.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KB_HD __host__ __device__ __forceinline__
#else
#define KB_HD inline
#endif

namespace kb {

typedef uint32_t u32;
typedef uint64_t u64;

static constexpr u32 P = 0x7f000001u;
static constexpr u32 MU = 0x81000001u;   // +p^{-1} mod 2^32  (koala_bear.rs:25)
static constexpr u32 ONE = 0x01fffffeu;  // 2^32 mod p
static constexpr u32 TWO = 0x03fffffcu;
static constexpr u64 P_SHL32 = (u64)P << 32;

KB_HD u32 umin(u32 a, u32 b) { return a < b ? a : b; }

// a + b mod p, inputs in [0,p)
KB_HD u32 add(u32 a, u32 b) {
    u32 s = a + b;
    return umin(s, s - P);
}
KB_HD u32 sub(u32 a, u32 b) {
    u32 d = a - b;
    return umin(d, d + P);
}
KB_HD u32 neg(u32 a) { return a ? P - a : 0u; }
KB_HD u32 dbl(u32 a) { return add(a, a); }

KB_HD u32 mulhi(u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (u32)(((u64)a * b) >> 32);
#endif
}

// Montgomery reduction of x < 2^32 * p  (monty_31/utils.rs:107-127).  x_lo == (t*p)_lo, so the quotient is just
// the difference of the high words.
KB_HD u32 reduce(u64 x) {
    u32 t = (u32)x * MU;
    u32 u = mulhi(t, P);
    u32 d = (u32)(x >> 32) - u;
    return umin(d, d + P);
}
KB_HD u32 mul(u32 a, u32 b) { return reduce((u64)a * b); }
KB_HD u32 sqr(u32 a) { return mul(a, a); }
KB_HD u32 cube(u32 a) { return mul(mul(a, a), a); }

// x < 4 p^2 < 2^64: bring below 2^32 * p with one conditional subtraction, then reduce.
KB_HD u32 reduce4(u64 x) {
    u64 y = x - P_SHL32;
    return reduce(x >= P_SHL32 ? y : x);
}
// sum of up to 4 products, delayed reduction
KB_HD u32 dot4(u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3) {
    u64 x = (u64)a0 * b0 + (u64)a1 * b1 + (u64)a2 * b2 + (u64)a3 * b3;
    return reduce4(x);
}

KB_HD u32 to_monty(u32 x) { return (u32)((((u64)x) << 32) % P); }
KB_HD u32 from_monty(u32 x) { return reduce((u64)x); }

KB_HD u32 pow(u32 a, u64 e) {
    u32 r = ONE;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = mul(a, a);
        e >>= 1;
    }
    return r;
}
KB_HD u32 inv(u32 a) { return pow(a, (u64)P - 2); }

// -------------------------------------------------------------------------------------------------
// Quintic extension, basis 1, X, .., X^4 with X^5 = 1 - X^2
// (reference: quintic_extension/extension.rs:531-548)
// -------------------------------------------------------------------------------------------------
struct EF {
    u32 v[5];
};

KB_HD EF ef_zero() {
    EF r;
    r.v[0] = r.v[1] = r.v[2] = r.v[3] = r.v[4] = 0;
    return r;
}
KB_HD EF ef_one() {
    EF r = ef_zero();
    r.v[0] = ONE;
    return r;
}
KB_HD EF ef_from_base(u32 a) {
    EF r = ef_zero();
    r.v[0] = a;
    return r;
}
KB_HD bool ef_is_zero(const EF& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4]) == 0; }
KB_HD bool ef_eq(const EF& a, const EF& b) {
    return a.v[0] == b.v[0] && a.v[1] == b.v[1] && a.v[2] == b.v[2] && a.v[3] == b.v[3] && a.v[4] == b.v[4];
}
KB_HD EF ef_add(const EF& a, const EF& b) {
    EF r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.v[i] = add(a.v[i], b.v[i]);
    return r;
}
KB_HD EF ef_sub(const EF& a, const EF& b) {
    EF r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.v[i] = sub(a.v[i], b.v[i]);
    return r;
}
KB_HD EF ef_neg(const EF& a) {
    EF r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.v[i] = neg(a.v[i]);
    return r;
}
KB_HD EF ef_dbl(const EF& a) { return ef_add(a, a); }
KB_HD EF ef_mul_base(const EF& a, u32 b) {
    EF r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.v[i] = mul(a.v[i], b);
    return r;
}
KB_HD EF ef_add_base(const EF& a, u32 b) {
    EF r = a;
    r.v[0] = add(a.v[0], b);
    return r;
}

// compile-time loop: f(IntC<I>{}) for I in [I0, N).  Unlike "#pragma unroll" it cannot be refused for body size, so arrays
// indexed by the loop variable always stay in registers and table entries indexed by it become instruction literals.
template <int I>
struct IntC {
    static constexpr int value = I;
};
template <int I, int N, class F>
KB_HD void static_for(F&& f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<I + 1, N>(f);
    }
}

// Delayed reduction.  Products of reduced values are < p^2 < 2^62, so four of them fit a u64.  fold32 brings any u64 back
// below 2^57 + 2^32 without changing it mod p (2^32 = ONE mod p, a 25-bit constant): one v_mad_u64_u32 instead of the
// compare/subtract/select of a conditional subtraction.  After a fold there is room for three more products (< 2^64), or
// for one more product before `reduce` (which needs < 2^32 p).
KB_HD u64 fold32(u64 x) { return (u64)(u32)(x >> 32) * ONE + (u32)x; }

// sum_i a[i*SA] * b[i*SB] mod p (Montgomery product form), N >= 1, inputs in [0,p)
template <int N, int SA = 1, int SB = 1>
KB_HD u32 dot_n(const u32* a, const u32* b) {
    u64 x = 0;
    int room = 4;  // products that may still be added to x without overflow
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (room == 0) {
            x = fold32(x);
            room = 3;
        }
        x += (u64)a[i * SA] * b[i * SB];
        room--;
    }
    // x < 2^32 p holds after at most 1 product on a folded value or 2 products from zero
    const bool small = N <= 2 || (N > 4 && (N - 4) % 3 == 1);
    if (!small) x = fold32(x);
    return reduce(x);
}

// 5-term dot product: 4 products (< 4p^2 < 2^64), fold, the 5th, reduce.
KB_HD u32 dot5(const u32* a, u32 b0, u32 b1, u32 b2, u32 b3, u32 b4) {
    u64 x = (u64)a[0] * b0 + (u64)a[1] * b1 + (u64)a[2] * b2 + (u64)a[3] * b3;
    x = fold32(x) + (u64)a[4] * b4;  // < 2^57 + 2^32 + p^2 < 2^32 p
    return reduce(x);
}
// Product as 5 dot products of length 5: rows of the multiplication-by-b matrix in the basis above.
KB_HD EF ef_mul(const EF& a, const EF& b) {
    u32 b0m3 = sub(b.v[0], b.v[3]);
    u32 b1m4 = sub(b.v[1], b.v[4]);
    u32 b4m2 = sub(b.v[4], b.v[2]);
    u32 b3m14 = sub(b.v[3], b1m4);
    EF r;
    r.v[0] = dot5(a.v, b.v[0], b.v[4], b.v[3], b.v[2], b1m4);
    r.v[1] = dot5(a.v, b.v[1], b.v[0], b.v[4], b.v[3], b.v[2]);
    r.v[2] = dot5(a.v, b.v[2], b1m4, b0m3, b4m2, b3m14);
    r.v[3] = dot5(a.v, b.v[3], b.v[2], b1m4, b0m3, b4m2);
    r.v[4] = dot5(a.v, b.v[4], b.v[3], b.v[2], b1m4, b0m3);
    return r;
}
KB_HD EF ef_sqr(const EF& a) { return ef_mul(a, a); }
KB_HD EF ef_pow(EF a, u64 e) {
    EF r = ef_one();
    while (e) {
        if (e & 1) r = ef_mul(r, a);
        a = ef_sqr(a);
        e >>= 1;
    }
    return r;
}
// Frobenius x -> x^p is F_p-linear: x^p = sum_i x_i (X^i)^p.  The images of the basis are derived once from the field
// definition (the reference tabulates the same matrix, quintic_extension/mod.rs:19-48).
struct FrobeniusTable {
    EF img[5];
};
inline const FrobeniusTable& frobenius_table() {
    static const FrobeniusTable t = [] {
        FrobeniusTable r;
        for (int i = 0; i < 5; i++) {
            EF b = ef_zero();
            b.v[i] = ONE;
            r.img[i] = ef_pow(b, (u64)P);
        }
        return r;
    }();
    return t;
}
inline EF ef_frobenius(const EF& a) {
    const FrobeniusTable& t = frobenius_table();
    EF r = ef_zero();
    for (int i = 0; i < 5; i++) r = ef_add(r, ef_mul_base(t.img[i], a.v[i]));
    return r;
}
// inverse through the norm (host only): conj = a^(p + p^2 + p^3 + p^4), a^-1 = conj / (a * conj), a*conj in F_p.
// The value is unique, so any method matches the reference (quintic_extension/extension.rs:585-607).
inline EF ef_inv(const EF& a) {
    const EF f1 = ef_frobenius(a);                      // a^p
    const EF f12 = ef_mul(f1, ef_frobenius(f1));        // a^(p + p^2)
    const EF f34 = ef_frobenius(ef_frobenius(f12));     // a^(p^3 + p^4)
    const EF conj = ef_mul(f12, f34);
    const EF n = ef_mul(a, conj);
    return ef_mul_base(conj, inv(n.v[0]));
}

}  // namespace kb
