// CPU check of the product's field header (leanmultisig_amd/csrc/kb.h, poseidon16.h) against 128-bit integer arithmetic:
// the delayed-reduction dot products at their overflow boundary (every operand p - 1), the reference's NEON regression
// operands (crates/backend/koala-bear/src/aarch64_neon/packing.rs:44-50: expected value = the scalar dot product), the
// quintic product against schoolbook multiplication mod X^5 + X^2 - 1, and the Poseidon1-16 KAT through the host path.
#include <cstdio>
#include <initializer_list>
#include <cstdlib>
#include <cstring>
#include "../../leanmultisig_amd/csrc/poseidon16.h"
using namespace kb;
typedef unsigned __int128 u128;
static u32 to_m(u64 x) { return (u32)(((u128)(x % P) << 32) % P); }
static u32 from_m(u32 x) { return from_monty(x); }
static int fails = 0;
#define CHECK(c, ...)                 \
    do {                              \
        if (!(c)) {                   \
            fails++;                  \
            printf("FAIL: " __VA_ARGS__); \
            printf("\n");             \
        }                             \
    } while (0)

template <int N>
static void dot_case(const u32* a, const u32* b, const char* what) {
    u128 ref = 0;
    for (int i = 0; i < N; i++) ref += (u128)a[i] * b[i];
    u32 am[N], bm[N];
    for (int i = 0; i < N; i++) am[i] = to_m(a[i]), bm[i] = to_m(b[i]);
    const u32 got = from_m(dot_n<N>(am, bm));
    CHECK(got == (u32)(ref % P), "dot_n<%d> %s: got %u want %u", N, what, got, (u32)(ref % P));
}
template <int N>
static void dot_extremes() {
    u32 a[N], b[N];
    for (int i = 0; i < N; i++) a[i] = b[i] = P - 1;
    dot_case<N>(a, b, "all p-1");
    for (int i = 0; i < N; i++) a[i] = (u32)((0x9E3779B97F4A7C15ull * (i + 1)) % P), b[i] = (u32)((0xC2B2AE3D27D4EB4Full * (i + 7)) % P);
    dot_case<N>(a, b, "pseudo-random");
}
template <int N>
struct DotSweep {
    static void run() {
        dot_extremes<N>();
        DotSweep<N - 1>::run();
    }
};
template <>
struct DotSweep<0> {
    static void run() {}
};

int main() {
    // 1. reference regression operands (dot_product_5_carry_cascade_regression)
    {
        const u32 lhs[5] = {P - 1, 1, 8, P - 3, P - 2}, rhs[5] = {P - 4, 9, P - 2, P - 5, 6};
        dot_case<5>(lhs, rhs, "NEON regression operands");
        u32 am[5], bm[5];
        for (int i = 0; i < 5; i++) am[i] = to_m(lhs[i]), bm[i] = to_m(rhs[i]);
        u128 ref = 0;
        for (int i = 0; i < 5; i++) ref += (u128)lhs[i] * rhs[i];
        CHECK(from_m(dot5(am, bm[0], bm[1], bm[2], bm[3], bm[4])) == (u32)(ref % P), "dot5 regression operands");
    }
    // 2. every dot length used by the kernels (1..40) at the overflow boundary
    DotSweep<40>::run();
    // 3. fold32 preserves the residue and lands below 2^57 + 2^32
    for (u64 x : {0ull, 1ull, 0xffffffffull, 0x100000000ull, 0xffffffffffffffffull, 0x8000000000000000ull, 0x7f000001ull << 32}) {
        const u64 y = fold32(x);
        CHECK(y % P == x % P && y < (1ull << 57) + (1ull << 32), "fold32(%llx)", (unsigned long long)x);
    }
    // 4. quintic product vs schoolbook mod X^5 + X^2 - 1  (X^5 = 1 - X^2)
    for (int t = 0; t < 200; t++) {
        u32 a[5], b[5];
        for (int i = 0; i < 5; i++) {
            a[i] = t == 0 ? P - 1 : (u32)((0x9E3779B97F4A7C15ull * (t * 5 + i + 1)) % P);
            b[i] = t == 0 ? P - 1 : (u32)((0xD6E8FEB86659FD93ull * (t * 5 + i + 3)) % P);
        }
        u128 c[9] = {0};
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 5; j++) c[i + j] += (u128)a[i] * b[j];
        long long r[9];
        for (int m = 0; m < 9; m++) r[m] = (long long)(c[m] % P);
        for (int m = 8; m >= 5; m--) {  // X^m = X^(m-5) - X^(m-3)
            r[m - 5] = (r[m - 5] + r[m]) % P;
            r[m - 3] = ((r[m - 3] - r[m]) % (long long)P + P) % P;
            r[m] = 0;
        }
        EF x, y;
        for (int i = 0; i < 5; i++) x.v[i] = to_m(a[i]), y.v[i] = to_m(b[i]);
        const EF z = ef_mul(x, y);
        for (int i = 0; i < 5; i++) CHECK(from_m(z.v[i]) == (u32)r[i], "ef_mul case %d coefficient %d", t, i);
        const EF inv = ef_inv(x);
        const EF one = ef_mul(x, inv);
        CHECK(ef_eq(one, ef_one()), "ef_inv case %d", t);
    }
    // 5. Poseidon1-16 KAT (poseidon1_koalabear_16.rs:1083-1091) through the host permutation (the transcript's)
    {
        u32 s[16];
        for (int i = 0; i < 16; i++) s[i] = to_m(i);
        poseidon16_permute(s);
        const u32 want[16] = {610090613, 935319874, 1893335292, 796792199, 356405232, 552237741, 55134556, 1215104204,
                              1823723405, 1133298033, 1780633798, 1453946561, 710069176, 1128629550, 1917333254, 1175481618};
        for (int i = 0; i < 16; i++) CHECK(from_m(s[i]) == want[i], "poseidon KAT word %d", i);
        // MDS with and without bias agree with the definition
        u32 v[16], w[16], bias[16];
        for (int i = 0; i < 16; i++) v[i] = w[i] = P - 1 - i, bias[i] = P - 1;
        mds_circ16_bias(v, bias);
        mds_circ16(w);
        for (int i = 0; i < 16; i++) CHECK(v[i] == add(w[i], bias[i]), "mds bias lane %d", i);
    }
    printf("%s\n", fails ? "FAILED" : "kb header ok");
    return fails != 0;
}
