// GKR for a sum of fractions sum n_i / d_i  (reference: crates/sub_protocols/src/quotient_gkr/{mod,layers,sumcheck_utils}.rs).
//
// Device layout: natural index order (the reference's chunk-bit-reversed SIMD packing is a CPU artefact; all
// transcript values are layout independent).  A layer with 2^v entries is {nums, dens}: nums is one base plane for the
// input layer and SoA EF above it, dens is SoA EF.  Children of parent j are entries 2j (left) and 2j+1 (right), so one
// sumcheck pair (parents 2j', 2j'+1) is 4 consecutive entries: one 16-byte load per plane.
// During a layer's sumcheck the four multilinears (n_l, n_r, d_l, d_r) live as 4 SoA EF arrays that halve every round
// (LSB-first folding, sumcheck_utils.rs:278-357).
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include "lm_common.h"
#include "lm_eqsplit.h"

using namespace kb;

struct lm_gkr {
    u32 n_vars = 0;
    const u32* d_nums0 = nullptr;  // caller's input layer (base)
    const u32* d_dens0 = nullptr;  // caller's input layer (SoA EF)
    std::vector<u32*> nums, dens;  // layers n_vars-1 .. 5 (index 0 = 2^(n_vars-1) entries), SoA EF, owned
    // Active prefix (sub_protocols/src/quotient_gkr/sumcheck_utils.rs:136,225,331 do the same symbolically): only the first
    // valid0 entries of the input layer / valid[i] of layer i exist in memory — the rest is the neutral pair (0, 1), which stays
    // (0, 1) under layer construction and (after the alpha shift) (0, alpha, 1, 1) under every fold, so kernels synthesise it
    // and the sums over the all-padding part are closed forms (alpha * sum of eq weights) added on the host.  Multiples of 8.
    u64 valid0 = 0;
    std::vector<u64> valid;
    u64 arr_valid = 0;  // valid length of the current work arrays
    u32* work[2] = {nullptr, nullptr};  // ping-pong: 4 arrays (nl, nr + alpha dr, dl, dr) x 5 planes
    u64 work_words = 0;
    PrefixEqTables eqt;  // prefix eq tables of the current layer
    // state of the layer being proven
    u32 K = 0;        // number of rounds = number of coordinates of the claim point
    u32 round = 0;
    int cur = -1;     // which work buffer holds the current arrays (-1: still in layer storage)
    u64 m = 0;        // current length of each of the 4 arrays (before the pending folds)
    EF alpha;
    std::vector<EF> point;    // the layer's claim point (K coordinates)
    std::vector<EF> pending;  // challenges received but not yet folded into the arrays (at most 2)
    bool la_valid = false;    // the next round's (c0, c2) as quadratics in the pending challenge: c0 = la[0..3), c2 = la[3..6)
    EF la[6];
    u32 fin_m = 0;            // tail of the layer published by the last launch: fin[array][i], i < fin_m <= 4
    EF fin[4][4];
    bool tail_live = false;   // a k_gkr_tail workgroup is resident and waiting for the next pair of challenges
    u32 tail_seq = 0;         // sequence number of its last publication (the next one must be tail_seq + 1)
    u32 tail_W = 1, tail_S = 0;  // its workgroups and the length of a workgroup's slice (the host mirrors the kernel's schedule)
    bool tail_solo = true;       // only workgroup 0 is left (always true for tail_W == 1)
    u32* d_merge = nullptr;      // hand-over scratch of a multi-workgroup tail
    // the next launch of the layer, enqueued ahead of its two challenges (gkr_shoot with a mail number)
    bool ahead = false;
    bool ahead_ok = false;  // decided per layer (lm_gkr_layer_begin) and re-checked per launch: this prover has the device to itself
    // Fail soft (lm_gkr_round): the challenges of the layer in progress, so that the layer can be re-run from its storage with one launch
    // per exchange when a resident kernel (a tail, a launch enqueued ahead) never got its wave slots; no_resident: this object has
    // fallen back once and uses neither for the rest of its life.
    std::vector<EF> history;
    bool no_resident = false;
    u32 exchanges_tail = 0, exchanges_ahead = 0;  // counted for the fault injection of the tests (LM_GKR_FAULT)
    struct GkrShotT {
        u32 t, F, p, seq, mail_no;
        u64 m_out, n_threads;
        bool la, tail;
        int dst;
        u32 tail_W, tail_S;
        bool tail_solo;
    } ahead_shot;
};
using GkrShot = lm_gkr::GkrShotT;

// ---- layer construction (layers.rs:124-189): (n0 d1 + n1 d0, d0 d1) ------------------------------------------------
template <bool BASE>
__global__ __launch_bounds__(256) void k_gkr_layer_up(const u32* __restrict__ n_in, const u32* __restrict__ d_in, u64 m_out,
                                                      u32* __restrict__ n_out, u32* __restrict__ d_out, u64 valid_in, u64 valid_out) {
    const u64 plane_in = 2 * m_out;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < valid_out; i += (u64)gridDim.x * 256) {
        if (2 * i >= valid_in) {  // both children are padding (0, 1): so is the parent
#pragma unroll
            for (int k = 0; k < 5; k++) {
                n_out[(u64)k * m_out + i] = 0;
                d_out[(u64)k * m_out + i] = k == 0 ? ONE : 0;
            }
            continue;
        }
        EF d0, d1;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint2 v = *reinterpret_cast<const uint2*>(d_in + (u64)k * plane_in + 2 * i);
            d0.v[k] = v.x;
            d1.v[k] = v.y;
        }
        EF no;
        if (BASE) {
            uint2 v = *reinterpret_cast<const uint2*>(n_in + 2 * i);
            no = ef_add(ef_mul_base(d1, v.x), ef_mul_base(d0, v.y));
        } else {
            EF n0, n1;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                uint2 v = *reinterpret_cast<const uint2*>(n_in + (u64)k * plane_in + 2 * i);
                n0.v[k] = v.x;
                n1.v[k] = v.y;
            }
            no = ef_add(ef_mul(d1, n0), ef_mul(d0, n1));
        }
        EF dd = ef_mul(d0, d1);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            n_out[(u64)k * m_out + i] = no.v[k];
            d_out[(u64)k * m_out + i] = dd.v[k];
        }
    }
}

// Two levels per pass: thread i reads the four children of grandparent i (one 16-byte load per plane), stores parents 2i, 2i + 1
// and the grandparent — the middle level is written but not read back (layer construction traffic 2.0 -> 1.6 GB at 2^25, half
// the launches).  Same values as two k_gkr_layer_up passes; padding as there: a node whose children are all padding is (0, 1).
template <bool BASE>
__global__ __launch_bounds__(256) void k_gkr_layer_up2(const u32* __restrict__ n_in, const u32* __restrict__ d_in, u64 m_out,
                                                       u32* __restrict__ n_out, u32* __restrict__ d_out, u32* __restrict__ n_out2,
                                                       u32* __restrict__ d_out2, u64 valid_in, u64 valid_out, u64 valid_out2) {
    const u64 plane_in = 2 * m_out, m_out2 = m_out / 2;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < valid_out2; i += (u64)gridDim.x * 256) {
        EF pn[2], pd[2];
        if (4 * i < valid_in) {  // (valid_in is a multiple of 8: the four children exist together)
            EF d[4];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint4 v = *reinterpret_cast<const uint4*>(d_in + (u64)k * plane_in + 4 * i);
                d[0].v[k] = v.x, d[1].v[k] = v.y, d[2].v[k] = v.z, d[3].v[k] = v.w;
            }
            if (BASE) {
                const uint4 v = *reinterpret_cast<const uint4*>(n_in + 4 * i);
                pn[0] = ef_add(ef_mul_base(d[1], v.x), ef_mul_base(d[0], v.y));
                pn[1] = ef_add(ef_mul_base(d[3], v.z), ef_mul_base(d[2], v.w));
            } else {
                EF n[4];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint4 v = *reinterpret_cast<const uint4*>(n_in + (u64)k * plane_in + 4 * i);
                    n[0].v[k] = v.x, n[1].v[k] = v.y, n[2].v[k] = v.z, n[3].v[k] = v.w;
                }
                pn[0] = ef_add(ef_mul(d[1], n[0]), ef_mul(d[0], n[1]));
                pn[1] = ef_add(ef_mul(d[3], n[2]), ef_mul(d[2], n[3]));
            }
            pd[0] = ef_mul(d[0], d[1]);
            pd[1] = ef_mul(d[2], d[3]);
        } else {
            pn[0] = pn[1] = ef_zero();
            pd[0] = pd[1] = ef_one();
        }
        if (2 * i < valid_out) {  // (valid_out is even: both parents are inside or outside together)
#pragma unroll
            for (int k = 0; k < 5; k++) {
                *reinterpret_cast<uint2*>(n_out + (u64)k * m_out + 2 * i) = make_uint2(pn[0].v[k], pn[1].v[k]);
                *reinterpret_cast<uint2*>(d_out + (u64)k * m_out + 2 * i) = make_uint2(pd[0].v[k], pd[1].v[k]);
            }
        }
        EF gn = ef_zero(), gd = ef_one();
        if (2 * i < valid_out) {
            gn = ef_add(ef_mul(pd[1], pn[0]), ef_mul(pd[0], pn[1]));
            gd = ef_mul(pd[0], pd[1]);
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            n_out2[(u64)k * m_out2 + i] = gn.v[k];
            d_out2[(u64)k * m_out2 + i] = gd.v[k];
        }
    }
}

// ---- the layer sumcheck, two rounds per launch -----------------------------------------------------------------------
// Round t of a layer (sumcheck_utils.rs:65-109, 278-357) sums, over pairs (a, b) = entries (2j, 2j+1) of the four arrays,
//     w_t(j) * [ c0 = nl_a dr_a + nr_a dl_a + alpha dl_a dr_a ,  c2 = Dnl Ddr + Dnr Ddl + alpha Ddl Ddr ],  D = b - a.
// Two exact identities restructure it (field arithmetic: every transcript value is unchanged):
//  * alpha is folded into one array: with nr~ = nr + alpha dr (linear, so it commutes with every fold) the bracket is
//    e(a) = nl_a dr_a + nr~_a dl_a and c2 = Dnl Ddr + Dnr~ Ddl — 2 products instead of 3, no separate denominator sums;
//  * look-ahead: after folding by the challenge r of round t the next round pairs y0 = x0 + r (x1 - x0) with
//    y1 = x2 + r (x3 - x2).  Its (c0', c2') are QUADRATIC polynomials in r whose coefficients are sums over quads
//    (x0..x3) that do not depend on r:
//        c0'(r) = E0 + r (E1 - E0 - C01) + r^2 C01            E_i = sum w' e(x_i),  C01 = sum w' c2(x0, x1)
//        c2'(r) = T0 + r (T3 - T0 - T2) + r^2 T2              T0 / T3 / T2 = the c2 form on x2 - x0 / x3 - x1 / (x3 - x2) - (x1 - x0)
//    with the eq weight w'(j') of the quad, and w_t(2j' + b) = w'(j') * (b ? pt : 1 - pt), pt = the last coordinate of round
//    t's eq prefix.  So ONE pass over the data yields round t AND round t+1: the host evaluates the quadratics at r
//    (lm_gkr_round below), and the next launch folds by both challenges at once.  A layer of K rounds costs ceil(K/2)
//    launches / host round trips and reads the big arrays half as often.
// Thread i owns output entry i (after F folds of 2^F consecutive inputs: one 8/16-byte load per plane); the four lanes of a
// quad exchange their entries by DPP quad_perm and lane class c = i & 3 computes  c=0: T2 and T3,  c=1: C01,  c=2: T0,
// c=3: C23 (round t's second pair).  Per-class sums leave the block as 4 classes x 3 slots x 5 words.
template <int C>
__device__ __forceinline__ u32 quad_bcast(u32 x) {
    return (u32)__builtin_amdgcn_mov_dpp((int)x, C * 0x55, 0xf, 0xf, true);  // quad_perm:[C,C,C,C]
}
__device__ __forceinline__ EF fold1(const EF& a, const EF& b, const EF& r) { return ef_add(a, ef_mul(r, ef_sub(b, a))); }
__device__ __forceinline__ EF fold1_base(u32 a, u32 b, const EF& r) {
    EF t = ef_mul_base(r, sub(b, a));
    t.v[0] = add(t.v[0], a);
    return t;
}

static constexpr u32 GKR_SUM_WORDS = 60;  // [class 4][slot 3][5]
static constexpr u32 GKR_FIN_AT = 64;     // h_res offset of the last <= 4 entries of each array: [(array * 5 + k) * 4 + i]

// one plane of layer storage: (left, right) children interleaved; N consecutive pairs starting at src
template <int N>
__device__ __forceinline__ void load_lr(const u32* __restrict__ src, u32 (&l)[N], u32 (&r)[N]) {
    if constexpr (N == 1) {
        const uint2 v = *reinterpret_cast<const uint2*>(src);
        l[0] = v.x, r[0] = v.y;
    } else {
#pragma unroll
        for (int h = 0; h < N / 2; h++) {
            const uint4 v = reinterpret_cast<const uint4*>(src)[h];
            l[2 * h] = v.x, r[2 * h] = v.y, l[2 * h + 1] = v.z, r[2 * h + 1] = v.w;
        }
    }
}
// N consecutive words of one plane of a work array
template <int N>
__device__ __forceinline__ void load_n(const u32* __restrict__ src, u32 (&a)[N]) {
    if constexpr (N == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(src);
        a[0] = v.x, a[1] = v.y, a[2] = v.z, a[3] = v.w;
    } else if constexpr (N == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(src);
        a[0] = v.x, a[1] = v.y;
    } else {
        a[0] = *src;
    }
}
// LSB-first folds of N = 2^F consecutive entries: r0 joins (0,1), (2,3); r1 joins the results
template <int N>
__device__ __forceinline__ EF fold_n(const EF (&in)[N], const EF& r0, const EF& r1) {
    if constexpr (N == 1)
        return in[0];
    else if constexpr (N == 2)
        return fold1(in[0], in[1], r0);
    else
        return fold1(fold1(in[0], in[1], r0), fold1(in[2], in[3], r0), r1);
}
template <int N>
__device__ __forceinline__ EF fold_n_base(const u32 (&in)[N], const EF& r0, const EF& r1) {
    if constexpr (N == 1)
        return ef_from_base(in[0]);
    else if constexpr (N == 2)
        return fold1_base(in[0], in[1], r0);
    else
        return fold1(fold1_base(in[0], in[1], r0), fold1_base(in[2], in[3], r0), r1);
}

// MODE 0: layer storage with base numerators (the caller's input layer), 1: layer storage with EF numerators, 2: the four
// SoA work arrays (nl, nr~, dl, dr).  F: number of pending challenges folded in first.  LA: compute the look-ahead sums.

// the four arrays at output index i: inputs (i << F) .. + 2^F folded by (r0, r1)
template <int MODE, int F>
__device__ __forceinline__ void gkr_entry(const u32* __restrict__ n_in, const u32* __restrict__ d_in, const u32* __restrict__ arr_in, u64 m_in,
                                          u64 i, const EF& r0, const EF& r1, const EF& alpha, EF (&x)[4]) {
    constexpr int NIN = 1 << F;  // inputs per output entry
    if constexpr (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            EF in[NIN];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                u32 a[NIN];
                load_n<NIN>(arr_in + ((u64)q * 5 + k) * m_in + (i << F), a);
#pragma unroll
                for (int e = 0; e < NIN; e++) in[e].v[k] = a[e];
            }
            x[q] = fold_n<NIN>(in, r0, r1);
        }
    } else {
        // storage: array index y <-> entries 2y (left), 2y + 1 (right); this thread's inputs y = (i << F) .. + NIN
        const u64 plane = 2 * m_in;
        const u64 at = i << (F + 1);
        {
            EF l[NIN], r[NIN];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                u32 a[NIN], b[NIN];
                load_lr<NIN>(d_in + (u64)k * plane + at, a, b);
#pragma unroll
                for (int e = 0; e < NIN; e++) l[e].v[k] = a[e], r[e].v[k] = b[e];
            }
            x[2] = fold_n<NIN>(l, r0, r1), x[3] = fold_n<NIN>(r, r0, r1);
        }
        if constexpr (MODE == 0) {
            u32 a[NIN], b[NIN];
            load_lr<NIN>(n_in + at, a, b);
            x[0] = fold_n_base<NIN>(a, r0, r1), x[1] = fold_n_base<NIN>(b, r0, r1);
        } else {
            EF l[NIN], r[NIN];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                u32 a[NIN], b[NIN];
                load_lr<NIN>(n_in + (u64)k * plane + at, a, b);
#pragma unroll
                for (int e = 0; e < NIN; e++) l[e].v[k] = a[e], r[e].v[k] = b[e];
            }
            x[0] = fold_n<NIN>(l, r0, r1), x[1] = fold_n<NIN>(r, r0, r1);
        }
        x[1] = ef_add(x[1], ef_mul(alpha, x[3]));  // nr~ = nr + alpha dr
    }
}

// sums of one entry: its own bracket e and the quad's difference forms of its lane class (see above); w = eq weight of the quad
// (LA) or of the pair (!LA).  All four lanes of a quad call this together (idle lanes with x = 0).
// BASE0: x[0] (the left numerators) is a base-field value in every lane of the quad — the caller's input layer before any fold —, so
// its three products are base-by-extension (5 multiplications instead of 25) and only its plane 0 travels through the quad.
template <bool LA, bool BASE0 = false>
__device__ __forceinline__ void gkr_quad_sums(const EF (&x)[4], const EF& w, u32 cls, EF& acc_e, EF& acc_x, EF& acc_y) {
    const EF e = ef_add(BASE0 ? ef_mul_base(x[3], x[0].v[0]) : ef_mul(x[0], x[3]), ef_mul(x[1], x[2]));
    EF A[4], H[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int k = 0; k < 5; k++) {
            if (BASE0 && q == 0 && k > 0) {
                A[q].v[k] = 0, H[q].v[k] = 0;
                continue;
            }
            const u32 v = x[q].v[k];
            const u32 v0 = quad_bcast<0>(v), v1 = quad_bcast<1>(v), v2 = quad_bcast<2>(v), v3 = quad_bcast<3>(v);
            const u32 d1 = sub(v1, v0), d3 = sub(v3, v2);
            if (LA) {
                const u32 ee = sub(v2, v0), gg = sub(d3, d1);
                A[q].v[k] = cls == 0 ? gg : cls == 1 ? d1 : cls == 2 ? ee : d3;
                H[q].v[k] = sub(v3, v1);
            } else {
                A[q].v[k] = cls == 1 ? d1 : d3;
            }
        }
    const EF X = ef_add(BASE0 ? ef_mul_base(A[3], A[0].v[0]) : ef_mul(A[0], A[3]), ef_mul(A[1], A[2]));
    acc_e = ef_add(acc_e, ef_mul(e, w));
    acc_x = ef_add(acc_x, ef_mul(X, w));
    if (LA) {
        const EF Y = ef_add(BASE0 ? ef_mul_base(H[3], H[0].v[0]) : ef_mul(H[0], H[3]), ef_mul(H[1], H[2]));
        acc_y = ef_add(acc_y, ef_mul(Y, w));
    }
}

// per-class block sums (lanes of equal lane & 3) -> tot[GKR_SUM_WORDS], valid in threads < GKR_SUM_WORDS after the call
// (lds: (blockDim.x / 64) * GKR_SUM_WORDS words)
__device__ __forceinline__ void gkr_block_sums(const EF& acc_e, const EF& acc_x, const EF& acc_y, u32* lds, u32* tot) {
    u32 v[15];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = acc_e.v[k], v[5 + k] = acc_x.v[k], v[10 + k] = acc_y.v[k];
#pragma unroll
    for (int s = 0; s < 15; s++)
#pragma unroll
        for (int off = 32; off >= 4; off >>= 1) v[s] = add(v[s], (u32)__shfl_down(v[s], off, 64));
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int s = 0; s < 15; s++) lds[wave * GKR_SUM_WORDS + lane * 15 + s] = v[s];
    }
    __syncthreads();
    if (threadIdx.x < GKR_SUM_WORDS) {
        u32 s = 0;
        for (u32 wv = 0; wv < (blockDim.x >> 6); wv++) s = add(s, lds[wv * GKR_SUM_WORDS + threadIdx.x]);
        tot[threadIdx.x] = s;
    }
    __syncthreads();
}

template <int MODE, int F, bool LA>
__global__ __launch_bounds__(256) void k_gkr_step(const u32* __restrict__ n_in, const u32* __restrict__ d_in,
                                                  const u32* __restrict__ arr_in, u64 m_out, u64 n_threads, u64 valid_in, EF r0, EF r1, EF alpha,
                                                  EqSplit eq,
                                                  u32* __restrict__ out, unsigned long long* __restrict__ acc, u32* __restrict__ done_counter,
                                                  u32* __restrict__ h_res, u32 seq, const u32* __restrict__ h_mail, u32* __restrict__ d_relay,
                                                  u32 mail_no) {
    __shared__ u32 lds[4 * GKR_SUM_WORDS];
    __shared__ u32 tot[GKR_SUM_WORDS];
    if constexpr (F > 0) {
        // enqueued ahead of its challenges (lm_gkr_round): they arrive as message mail_no; a dismissed kernel publishes nothing
        if (mail_no && !lm_mail_receive(h_mail, d_relay, mail_no, lds, r0, r1)) return;
    }
    const u64 m_in = m_out << F;
    const u32 cls = threadIdx.x & 3;
    EF acc_e = ef_zero(), acc_x = ef_zero(), acc_y = ef_zero();
    // n_threads <= m_out outputs are computed (a multiple of 8 or m_out itself); outputs whose inputs all lie beyond
    // valid_in are the folded padding (0, alpha, 1, 1), synthesised without touching memory
    for (u64 base = (u64)blockIdx.x * 256; base < n_threads; base += (u64)gridDim.x * 256) {
        const u64 i = base + threadIdx.x;
        const bool active = i < n_threads;
        EF x[4];  // nl, nr~, dl, dr at output index i
        const bool padding = active && (MODE == 2 ? (i << F) : (i << (F + 1))) >= valid_in;
        if (padding) {
            x[0] = ef_zero(), x[1] = alpha, x[2] = ef_one(), x[3] = ef_one();
        }
        if (active) {
            if (!padding) gkr_entry<MODE, F>(n_in, d_in, arr_in, m_in, i, r0, r1, alpha, x);
            if constexpr (F > 0) {
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int k = 0; k < 5; k++) out[((u64)q * 5 + k) * m_out + i] = x[q].v[k];
                if (m_out <= 4) {  // the tail of the layer goes to the host as well (lm_gkr_layer_end folds it)
#pragma unroll
                    for (int q = 0; q < 4; q++)
#pragma unroll
                        for (int k = 0; k < 5; k++) lm_store_system(h_res + GKR_FIN_AT + (q * 5 + k) * 4 + i, x[q].v[k]);
                    lm_wait_stores();
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) x[q] = ef_zero();
        }
        // ---- sums: own entry, then the quad's differences ----
        const EF w = eq_split_at(eq, active ? (LA ? i >> 2 : i >> 1) : 0);  // (n_threads is a multiple of 4: a quad is all-active or all-idle)
        gkr_quad_sums<LA, MODE == 0 && F == 0>(x, w, cls, acc_e, acc_x, acc_y);  // (padding and idle lanes: x[0] = 0, a base value too)
    }
    gkr_block_sums(acc_e, acc_x, acc_y, lds, tot);
    if (gridDim.x > 1 && !lm_grid_sum<GKR_SUM_WORDS>(tot, acc, done_counter, tot)) return;
    if (threadIdx.x < GKR_SUM_WORDS) {
        lm_store_system(h_res + threadIdx.x, tot[threadIdx.x]);
        lm_wait_stores();
    }
    __syncthreads();
    if (threadIdx.x == 0) lm_publish_flag(h_res, seq);
}

// ---- the resident tail of a layer ------------------------------------------------------------------------------------
// Once the four arrays have <= GKR_TAIL_MAX entries each, ONE workgroup keeps them in LDS and stays on the device for the
// rest of the layer: it publishes the sums of a round pair exactly as k_gkr_step does, then polls a mailbox line in pinned
// host memory for the two challenges, folds in LDS and goes on — no launch and no kernel completion per exchange
// (tools/ubench/mailbox.hip: 4.3 us per exchange against 8 us for the smallest launch + publish).  What bounds an exchange
// then is the LATENCY of one wave's instruction stream (a lone wave retires an instruction every ~8 cycles: the 21 extension
// multiplications k_gkr_step spends per entry are 9 us however few entries there are), so the work of one entry is spread
// over lanes of different WAVES, each role wave-uniform:
//   phase A: wave (j, q) folds array q of entries 64 j .. + 64 (3 multiplications), LDS -> LDS;
//   phase B: wave (j, role) reads its entry's quad of two arrays from LDS and multiplies ONE pair of operands, then the quad's
//            eq weight: role 0 / 1: (nl, dr) / (nr~, dl) at the entry itself — or, in lane class 3 whose own bracket no round
//            needs, the H = x3 - x1 forms (T3); role 2 / 3: the same two pairs on the class's difference form (k_gkr_step's A).
// Five multiplications deep instead of 21.  The sums leave in k_gkr_step's [class][slot][5] order.
// Mailbox line (16 words, lm_ctx::h_cmd): words 0..9 = r0, r1 with bit 31 (free: field words are < 2^31) carrying the parity
// of the sequence number the NEXT publication must use, word 10 = that sequence number, word 11 = GKR_TAIL_ABORT to dismiss
// the kernel.  A message is complete when all conditions hold in ONE wave-wide load of the line: no ordering between the
// host's stores is assumed.
// Above 256 entries the tail runs as W = entries / 256 workgroups, each owning a contiguous slice (LSB-first folding keeps a
// slice's outputs inside it): every workgroup publishes its own partial sums (slot w of the pinned buffer, the host adds
// them) and polls the same mailbox line; no workgroup talks to another until a slice is down to one quad or two, when the
// others hand their folded entries to workgroup 0 through device memory (agent-scope stores + a ticket) and leave.
static constexpr u32 GKR_TAIL_SLICE = 256;                 // entries per workgroup
static constexpr u32 GKR_TAIL_MAX_W = 64;      // what the buffers are sized for (LM_GKR_TAIL_W)
static constexpr u32 GKR_TAIL_DEFAULT_W = 16;  // measured (round 5, GKR stage): 16 workgroups 4.40 - 4.42 ms, 32: 4.44 - 4.57, 64: 4.81 - 4.82 — every
                                               // exchange of a wider tail waits for all its workgroups' polls and publications
static constexpr u32 GKR_TAIL_MAX = GKR_TAIL_SLICE * GKR_TAIL_MAX_W;
static constexpr u32 GKR_TAIL_THREADS = 4 * GKR_TAIL_SLICE;
static constexpr u32 GKR_TAIL_STEPS = 7;  // entries: 4096, 1024, 256, 64, 16, 4  /  2048, 512, 128, 32, 8, 2
static constexpr u32 GKR_TAIL_SLOT_AT = 1024, GKR_TAIL_SLOT_WORDS = 64;  // h_res: partial sums of workgroup w > 0 (+ 63: its flag)
static constexpr u32 GKR_TAIL_MERGE_WORDS = GKR_TAIL_MAX_W * 2 * 20;       // device scratch of the hand-over
static_assert(GKR_TAIL_SLOT_AT + GKR_TAIL_MAX_W * GKR_TAIL_SLOT_WORDS <= lm_ctx::RES_WORDS, "tail slots fit the pinned result buffer");
static constexpr u32 GKR_TAIL_ABORT = 0xdead0001u;
static constexpr unsigned long long GKR_TAIL_TIMEOUT = 300000000ull;  // wall_clock64 ticks (100 MHz): 3 s without an answer = abandoned
struct GkrTailEq {
    EqSplit e[GKR_TAIL_STEPS];
};

// array q of the entry at output index i (gkr_entry, one array)
template <int MODE, int F>
__device__ __forceinline__ EF gkr_entry_one(u32 q, const u32* __restrict__ n_in, const u32* __restrict__ d_in, const u32* __restrict__ arr_in,
                                            u64 m_in, u64 i, const EF& r0, const EF& r1, const EF& alpha) {
    constexpr int NIN = 1 << F;
    if constexpr (MODE == 2) {
        EF in[NIN];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            u32 a[NIN];
            load_n<NIN>(arr_in + ((u64)q * 5 + k) * m_in + (i << F), a);
#pragma unroll
            for (int e = 0; e < NIN; e++) in[e].v[k] = a[e];
        }
        return fold_n<NIN>(in, r0, r1);
    } else {
        const u64 plane = 2 * m_in;
        const u64 at = i << (F + 1);
        auto ext = [&](const u32* base, bool right) {
            EF s[NIN];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                u32 a[NIN], b[NIN];
                load_lr<NIN>(base + (u64)k * plane + at, a, b);
#pragma unroll
                for (int e = 0; e < NIN; e++) s[e].v[k] = right ? b[e] : a[e];
            }
            return fold_n<NIN>(s, r0, r1);
        };
        auto num = [&](bool right) {
            if constexpr (MODE == 0) {
                u32 a[NIN], b[NIN];
                load_lr<NIN>(n_in + at, a, b);
                return right ? fold_n_base<NIN>(b, r0, r1) : fold_n_base<NIN>(a, r0, r1);
            } else {
                return ext(n_in, right);
            }
        };
        if (q == 0) return num(false);
        if (q == 1) return ef_add(num(true), ef_mul(alpha, ext(d_in, true)));  // nr~ = nr + alpha dr
        return ext(d_in, q == 3);
    }
}

template <int MODE, int F>
__global__ __launch_bounds__(GKR_TAIL_THREADS) void k_gkr_tail(const u32* __restrict__ n_in, const u32* __restrict__ d_in,
                                                               const u32* __restrict__ arr_in, u32 m_out0, u64 valid_in, EF r0, EF r1, EF alpha,
                                                               GkrTailEq eqs, u32* __restrict__ h_res, u32 seq0, const u32* __restrict__ h_cmd,
                                                               u32* __restrict__ merge_buf, u32* __restrict__ merge_counter,
                                                               const u32* __restrict__ h_mail, u32* __restrict__ d_relay, u32 mail_no) {
    __shared__ __attribute__((aligned(16))) u32 arr[20 * GKR_TAIL_SLICE];  // [(array * 5 + k) * stride + i], stride = max(local length, 4)
    __shared__ u32 red[(GKR_TAIL_THREADS / 64) * 20];                      // per wave: [class][5]
    __shared__ u32 msg[16];
    if constexpr (F > 0) {
        // enqueued ahead of the two challenges of its first fold (lm_gkr_round): they arrive as message mail_no of the launch-ahead line
        if (mail_no) {
            if (!lm_mail_receive(h_mail, d_relay, mail_no, msg, r0, r1)) return;
            __syncthreads();
        }
    }
    const u32 lane = threadIdx.x & 63;
    const u32 wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32 role = wv & 3, grp = wv >> 2;  // phase A: array `role`; phase B: operand pair / form `role`
    const u32 i = grp * 64 + lane;            // local entry
    const u32 W = gridDim.x;
    u32 w = blockIdx.x;
    bool solo = W == 1;
    u32 S = m_out0 / W;  // local length
    u32 m_out = m_out0;  // global length
    u32 seq = seq0;
    for (u32 step = 0;; step++) {
        const bool la = m_out >= 4;
        const EqSplit eq = eqs.e[step];
        // ---- phase A: array `role` of local entry i ----
        EF xa = ef_zero();
        if (step == 0) {
            const u64 gi = (u64)w * S + i;
            const u64 m_in = (u64)m_out << F;
            if (i < S) {
                if ((MODE == 2 ? (gi << F) : (gi << (F + 1))) >= valid_in)
                    xa = role == 0 ? ef_zero() : role == 1 ? alpha : ef_one();  // folded padding (0, alpha, 1, 1)
                else
                    xa = gkr_entry_one<MODE, F>(role, n_in, d_in, arr_in, m_in, gi, r0, r1, alpha);
            }
        } else {
            // fold the LDS arrays by the two challenges of the message (in place: all reads, barrier, all writes)
            const u32 Sn = S >> 2;
            if (i < Sn) {
                EF c0, c1;
#pragma unroll
                for (int k = 0; k < 5; k++) c0.v[k] = msg[k], c1.v[k] = msg[5 + k];
                EF in[4];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint4 v = *reinterpret_cast<const uint4*>(arr + (role * 5 + k) * S + (i << 2));
                    in[0].v[k] = v.x, in[1].v[k] = v.y, in[2].v[k] = v.z, in[3].v[k] = v.w;
                }
                xa = fold_n<4>(in, c0, c1);
            }
            __syncthreads();
            if (!solo && Sn < 4) {
                // ---- hand-over: the slices are too short for a quad each; workgroup 0 takes everything ----
                if (w > 0) {
                    if (i < Sn) {
#pragma unroll
                        for (int k = 0; k < 5; k++) lm_store_agent(merge_buf + ((w * 2 + i) * 4 + role) * 5 + k, xa.v[k]);
                    }
                    lm_wait_stores();
                    __syncthreads();
                    if (threadIdx.x == 0) (void)lm_ticket(merge_counter);
                    return;
                }
                if (threadIdx.x == 0) {
                    const unsigned long long t0 = wall_clock64();
                    bool ok = true;
                    while (lm_load_agent(merge_counter) != W - 1)
                        if (wall_clock64() - t0 > GKR_TAIL_TIMEOUT) {
                            ok = false;
                            break;
                        }
                    lm_store_agent(merge_counter, 0);  // re-armed for the next kernel on the stream
                    msg[12] = ok ? 0 : GKR_TAIL_ABORT;
                }
                __syncthreads();
                if (msg[12] == GKR_TAIL_ABORT) return;
                const u32 Sm = W * Sn, stride = Sm < 4 ? 4 : Sm;
                // own entries, then the others' (thread = (array, plane, entry) of the gathered part), zero above Sm
                if (i < Sn) {
#pragma unroll
                    for (int k = 0; k < 5; k++) arr[(role * 5 + k) * stride + i] = xa.v[k];
                }
                for (u32 t = threadIdx.x; t < 20 * stride; t += GKR_TAIL_THREADS) {  // (up to 20 x 128 words with 64 workgroups)
                    const u32 qk = t / stride, e = t % stride;
                    if (e >= Sn) {
                        const u32 src_w = e / Sn, src_i = e % Sn;
                        arr[qk * stride + e] = e < Sm ? lm_load_agent(merge_buf + ((src_w * 2 + src_i) * 4 + qk / 5) * 5 + qk % 5) : 0;
                    }
                }
                solo = true;
                S = Sm;
            } else {
                S = Sn;
                const u32 stride = S < 4 ? 4 : S;
                if (i < stride) {  // (entries S .. 4 of a two-entry tail are zero)
#pragma unroll
                    for (int k = 0; k < 5; k++) arr[(role * 5 + k) * stride + i] = xa.v[k];
                }
            }
        }
        const u32 stride = S < 4 ? 4 : S;
        if (step == 0 && i < stride) {
#pragma unroll
            for (int k = 0; k < 5; k++) arr[(role * 5 + k) * stride + i] = xa.v[k];
        }
        __syncthreads();
        if (m_out <= 4 && threadIdx.x < 20 * m_out) {  // the layer's last entries go to the host as well (lm_gkr_layer_end folds them)
            const u32 qk = threadIdx.x / m_out, e = threadIdx.x % m_out;
            lm_store_system(h_res + GKR_FIN_AT + qk * 4 + e, arr[qk * stride + e]);
        }
        // ---- phase B: one operand pair of one form per thread ----
        EF acc = ef_zero();
        if (i < S) {
            const u32 cls = i & 3;
            const u32 qa = role & 1 ? 1 : 0, qb = role & 1 ? 2 : 3;
            EF A, B;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint4 va = *reinterpret_cast<const uint4*>(arr + (qa * 5 + k) * stride + (i & ~3u));
                const uint4 vb = *reinterpret_cast<const uint4*>(arr + (qb * 5 + k) * stride + (i & ~3u));
                if (role < 2) {
                    A.v[k] = cls == 0 ? va.x : cls == 1 ? va.y : cls == 2 ? va.z : sub(va.w, va.y);
                    B.v[k] = cls == 0 ? vb.x : cls == 1 ? vb.y : cls == 2 ? vb.z : sub(vb.w, vb.y);
                } else {
                    const u32 a1 = sub(va.y, va.x), a3 = sub(va.w, va.z), b1 = sub(vb.y, vb.x), b3 = sub(vb.w, vb.z);
                    A.v[k] = cls == 0 ? sub(a3, a1) : cls == 1 ? a1 : cls == 2 ? sub(va.z, va.x) : a3;
                    B.v[k] = cls == 0 ? sub(b3, b1) : cls == 1 ? b1 : cls == 2 ? sub(vb.z, vb.x) : b3;
                }
            }
            const EF wgt = eq_split_at(eq, la ? ((u64)(solo ? 0 : w) * S + i) >> 2 : 0);
            acc = ef_mul(ef_mul(A, B), wgt);
        }
        // ---- sums per lane class inside the wave, then across the waves of a role pair ----
        {
            u32 v[5];
#pragma unroll
            for (int k = 0; k < 5; k++) v[k] = acc.v[k];
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
                for (int off = 32; off >= 4; off >>= 1) v[k] = add(v[k], (u32)__shfl_down(v[k], off, 64));
            if (lane < 4) {
#pragma unroll
                for (int k = 0; k < 5; k++) red[wv * 20 + lane * 5 + k] = v[k];
            }
        }
        __syncthreads();
        u32* const out = h_res + (w == 0 ? 0 : GKR_TAIL_SLOT_AT + w * GKR_TAIL_SLOT_WORDS);
        if (threadIdx.x < GKR_SUM_WORDS) {
            const u32 cls = threadIdx.x / 15, sl = (threadIdx.x % 15) / 5, k = threadIdx.x % 5;
            // slot 0 (e): roles 0, 1 of classes 0..2; slot 1 (X): roles 2, 3; slot 2 (Y, class 0 only): roles 0, 1 of class 3
            const u32 src_cls = sl == 2 ? 3 : cls;
            const bool used = sl == 0 ? cls < 3 : sl == 1 ? true : cls == 0;
            u32 t = 0;
            if (used)
                for (u32 g4 = 0; g4 < GKR_TAIL_THREADS / 256; g4++)
                    for (u32 r = 0; r < 2; r++) t = add(t, red[(g4 * 4 + (sl == 1 ? 2 : 0) + r) * 20 + src_cls * 5 + k]);
            lm_store_system(out + threadIdx.x, t);
        }
        lm_wait_stores();
        __syncthreads();
        if (threadIdx.x == 0) lm_publish_flag_word(w == 0 ? h_res + lm_ctx::RES_FLAG : out + GKR_TAIL_SLOT_WORDS - 1, seq);
        if (m_out <= 4) return;  // the layer's last launch-equivalent: lm_gkr_layer_end folds the rest on the host
        // ---- wait for the two challenges ----
        seq++;
        if (threadIdx.x < 64) {
            const unsigned long long t0 = wall_clock64();
            u32 v = 0;
            bool done = false, dismissed = false;
            while (!done) {
                v = lane < 16 ? __hip_atomic_load(h_cmd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
                const bool ok = lane < 10 ? (v >> 31) == (seq & 1) : lane == 10 ? v == seq : true;
                done = __ballot(ok) == ~0ull;
                dismissed = __ballot(lane == 11 && v == GKR_TAIL_ABORT) != 0 || wall_clock64() - t0 > GKR_TAIL_TIMEOUT;
                if (dismissed) break;
            }
            if (lane < 16) msg[lane] = dismissed ? GKR_TAIL_ABORT : (v & 0x7fffffffu);
        }
        __syncthreads();
        if (msg[11] == GKR_TAIL_ABORT) return;
        m_out >>= 2;
    }
}

namespace {
double gkr_now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
EF host_ef(const u32* p) {
    EF r;
    memcpy(r.v, p, 20);
    return r;
}
bool gkr_two_levels() {  // LM_GKR_ONE_LEVEL=1: one k_gkr_layer_up pass per level (A/B measurements)
    static const bool on = getenv("LM_GKR_ONE_LEVEL") == nullptr;
    return on;
}
bool gkr_tail_enabled() {
    static const bool on = getenv("LM_GKR_NO_TAIL") == nullptr;
    return on;
}
// Resident workgroups hold their CU slots while they poll, and a tail makes progress only when ALL its workgroups are
// resident: the tails of every prover ON ONE DEVICE together must fit the chip (256 CUs x 2 workgroups of 1024 threads), or tails
// could wait for each other's slots until their timeouts.  Above the cap a layer falls back to launches.
// The count is kept PER DEVICE, across processes: a small shared-memory file keyed by the device's PCI bus id
// (/dev/shm/leanmultisig_tail_<bus id>) holds the total and one (pid, count) slot per process, so that several ranks on one GPU
// (tests, LM_BENCH_SINGLE_DEVICE) or several provers of a node divide the chip without being told how.  A process that died with
// workgroups reserved is noticed by the next process that attaches (its pid no longer exists) and its share is given back.  If the
// file cannot be created the counter is per process, as before.  LM_GKR_TAIL_MAX_WORKGROUPS lowers the cap (default 256 = the whole
// chip; a tail that does not get its slots still ends by its own 3 s timeout and the layer is reported as failed, never hung).
// LM_GKR_TAIL_W: workgroups of ONE tail (default 16: layers enter the tail at 2^12 entries per array; up to GKR_TAIL_MAX_W = 64 for
// A/B measurements)
u64 gkr_tail_max_entries() {
    static const u64 v = [] {
        const char* e = getenv("LM_GKR_TAIL_W");
        u32 w = e ? (u32)atoi(e) : GKR_TAIL_DEFAULT_W;
        if (w < 1) w = 1;
        if (w > GKR_TAIL_MAX_W) w = GKR_TAIL_MAX_W;
        return (u64)GKR_TAIL_SLICE * w;
    }();
    return v;
}
int gkr_tail_max_live() {
    static const int v = [] {
        const char* e = getenv("LM_GKR_TAIL_MAX_WORKGROUPS");
        const int x = e ? atoi(e) : 256;
        return x < 0 ? 0 : x;
    }();
    return v;
}
struct TailShared {  // the mapped file
    std::atomic<int> total;
    std::atomic<int> lock;
    struct Slot {
        std::atomic<int> pid, count;
    } slots[64];
};
struct TailCounter {
    TailShared* sh = nullptr;  // nullptr: process-local fallback
    int slot = -1;
    std::atomic<int> local{0};
};
TailCounter* tail_counter(int device) {
    static TailCounter counters[16];
    static std::atomic<int> ready[16];
    device = device < 0 || device >= 16 ? 0 : device;
    TailCounter& c = counters[device];
    int st = ready[device].load(std::memory_order_acquire);
    if (st == 2) return &c;
    int expect = 0;
    if (!ready[device].compare_exchange_strong(expect, 1)) {
        while (ready[device].load(std::memory_order_acquire) != 2) {
        }
        return &c;
    }
    char bus[64] = "";
    if (!getenv("LM_GKR_TAIL_PROCESS_LOCAL") && hipDeviceGetPCIBusId(bus, sizeof bus, device) == hipSuccess) {
        for (char* q = bus; *q; q++)
            if (*q == ':' || *q == '.') *q = '_';
        // one file per user and device, readable and writable by that user only (the counts steer scheduling decisions: another
        // local user must not be able to edit them; provers of different users do not see each other and are "foreign" to each
        // other, which the launch-ahead gate below treats like any other tenant: the layer falls back if a resident kernel starves)
        char path[160];
        snprintf(path, sizeof path, "/leanmultisig_tail_%u_%s", (unsigned)geteuid(), bus);
        const int fd = shm_open(path, O_RDWR | O_CREAT | O_NOFOLLOW, 0600);
        struct stat st;
        if (fd >= 0 && (fstat(fd, &st) != 0 || st.st_uid != geteuid() || (st.st_mode & 077) != 0)) {
            close(fd);  // not ours alone: count per process
        } else if (fd >= 0) {
            if (ftruncate(fd, sizeof(TailShared)) == 0) {  // (a new file is zero-filled: total 0, every slot free)
                void* m = mmap(nullptr, sizeof(TailShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (m != MAP_FAILED) c.sh = (TailShared*)m;
            }
            close(fd);
        }
    }
    if (c.sh) {
        TailShared* sh = c.sh;
        int z = 0;
        for (int spins = 0; !sh->lock.compare_exchange_weak(z, 1, std::memory_order_acquire); z = 0)
            if (++spins > (1 << 22)) break;  // (a holder that died: proceed; the worst case is a share returned twice, i.e. a cap too lax)
        const int me = (int)getpid();
        int live_sum = 0;
        for (auto& sl : sh->slots) {  // shares of processes that no longer exist go back
            const int pid = sl.pid.load();
            if (pid && pid != me && kill(pid, 0) != 0 && errno == ESRCH) {
                sh->total.fetch_sub(sl.count.exchange(0));
                sl.pid.store(0);
            }
            const int cnt = sl.count.load();
            if (cnt < 0 || cnt > 65536 || (cnt && !sl.pid.load())) sl.count.store(0);  // (nonsense left by a crash: a count without an owner)
            live_sum += sl.count.load();
        }
        // the total is what the slots add up to (checked whenever a process attaches: a file damaged by a crash heals here)
        if (sh->total.load() != live_sum) sh->total.store(live_sum);
        for (int k = 0; k < 64 && c.slot < 0; k++) {
            int free_pid = 0;
            if (sh->slots[k].pid.load() == me || sh->slots[k].pid.compare_exchange_strong(free_pid, me)) c.slot = k;
        }
        sh->lock.store(0, std::memory_order_release);
        if (c.slot < 0) c.sh = nullptr;  // 64 live processes on one device: count locally
    }
    ready[device].store(2, std::memory_order_release);
    return &c;
}
bool gkr_tail_reserve(lm_ctx* ctx, u32 W) {
    TailCounter* c = tail_counter(ctx->device);
    std::atomic<int>& total = c->sh ? c->sh->total : c->local;
    if (total.fetch_add((int)W, std::memory_order_acq_rel) + (int)W > gkr_tail_max_live()) {
        total.fetch_sub((int)W, std::memory_order_acq_rel);
        return false;
    }
    if (c->sh) c->sh->slots[c->slot].count.fetch_add((int)W, std::memory_order_acq_rel);
    return true;
}
void gkr_tail_unreserve(lm_ctx* ctx, int n) {
    TailCounter* c = tail_counter(ctx->device);
    (c->sh ? c->sh->total : c->local).fetch_sub(n, std::memory_order_acq_rel);
    if (c->sh) c->sh->slots[c->slot].count.fetch_sub(n, std::memory_order_acq_rel);
}
void gkr_tail_release(lm_ctx* ctx, lm_gkr* g) {
    gkr_tail_unreserve(ctx, (int)g->tail_W);
    g->tail_live = false;
}
// a resident workgroup whose layer is abandoned (error path, early free) is told to leave; it would otherwise poll until its
// own timeout
void gkr_tail_dismiss(lm_ctx* ctx, lm_gkr* g) {
    if (!g->tail_live) return;
    ((volatile u32*)ctx->h_cmd)[11] = GKR_TAIL_ABORT;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemsetAsync(ctx->d_sync + 1, 0, 4, ctx->stream);  // the hand-over ticket of an interrupted multi-workgroup tail
    ctx->h_cmd[11] = 0;
    gkr_tail_release(ctx, g);
}
// a + r (b + r c)
EF quad_at(const EF& a, const EF& b, const EF& c, const EF& r) { return ef_add(a, ef_mul(r, ef_add(b, ef_mul(r, c)))); }
}  // namespace
static void gkr_ahead_drop(lm_ctx* ctx, lm_gkr* g);
static bool gkr_launch_ahead_enabled(lm_ctx* ctx);

extern "C" {

void lm_gkr_free(lm_ctx* ctx, lm_gkr* g) {
    if (!g) return;
    gkr_ahead_drop(ctx, g);
    gkr_tail_dismiss(ctx, g);
    for (u32* p : g->nums) lm_pool_free(ctx, p);
    for (u32* p : g->dens) lm_pool_free(ctx, p);
    for (int i = 0; i < 2; i++) lm_pool_free(ctx, g->work[i]);
    lm_pool_free(ctx, g->eqt.d_buf);
    lm_pool_free(ctx, g->d_merge);
    delete g;
}

static u64 round_up8(u64 x, u64 cap) { return std::min<u64>((x + 7) & ~7ull, cap); }

int lm_gkr_build(lm_ctx* ctx, const uint32_t* d_nums, const uint32_t* d_dens, uint32_t n_vars, lm_gkr** out) {
    return lm_gkr_build_active(ctx, d_nums, d_dens, n_vars, 1ull << (n_vars <= 30 ? n_vars : 0), out);
}
int lm_gkr_build_active(lm_ctx* ctx, const uint32_t* d_nums, const uint32_t* d_dens, uint32_t n_vars, uint64_t active_len, lm_gkr** out) {
    LM_REQUIRE(ctx && d_nums && d_dens && out && n_vars > 5 && n_vars <= 30 && active_len >= 1 && active_len <= (1ull << n_vars));
    lm_gkr* g = new lm_gkr();
    g->no_resident = gkr_now_s() < ctx->no_resident_until_s;  // (a recent fallback on this context: lm_gkr_round)
    g->n_vars = n_vars;
    g->d_nums0 = d_nums;
    g->d_dens0 = d_dens;
    g->valid0 = round_up8(active_len, 1ull << n_vars);
    const u32* n_in = d_nums;
    const u32* d_in = d_dens;
    u64 valid_in = g->valid0;
    auto alloc_layer = [&](u64 m, u32** nn, u32** dd) {
        *nn = *dd = nullptr;
        if (lm_pool_alloc_t(ctx, nn, 5 * m * 4) != hipSuccess || lm_pool_alloc_t(ctx, dd, 5 * m * 4) != hipSuccess) {
            lm_set_error("lm_gkr_build: device allocation failed");
            lm_pool_free(ctx, *nn);
            return false;
        }
        g->nums.push_back(*nn);
        g->dens.push_back(*dd);
        return true;
    };
    for (u32 v = n_vars - 1; v >= 5;) {
        const u64 m = 1ull << v;
        const u64 valid_out = round_up8((valid_in + 1) / 2, m);
        u32 *nn, *dd;
        if (!alloc_layer(m, &nn, &dd)) {
            lm_gkr_free(ctx, g);
            return LM_E_NOMEM;
        }
        g->valid.push_back(valid_out);
        const bool base = v == n_vars - 1;
        if (v >= 6 && (valid_in & 7) == 0 && gkr_two_levels()) {  // two levels in one pass (k_gkr_layer_up2)
            const u64 m2 = m / 2, valid_out2 = round_up8((valid_out + 1) / 2, m2);
            u32 *nn2, *dd2;
            if (!alloc_layer(m2, &nn2, &dd2)) {
                lm_gkr_free(ctx, g);
                return LM_E_NOMEM;
            }
            g->valid.push_back(valid_out2);
            const u32 blocks = (u32)std::min<u64>((valid_out2 + 255) / 256, 4096);
            if (base)
                LM_LAUNCH(ctx, k_gkr_layer_up2<true>, dim3(blocks), dim3(256), 0, n_in, d_in, m, nn, dd, nn2, dd2, valid_in, valid_out, valid_out2);
            else
                LM_LAUNCH(ctx, k_gkr_layer_up2<false>, dim3(blocks), dim3(256), 0, n_in, d_in, m, nn, dd, nn2, dd2, valid_in, valid_out, valid_out2);
            n_in = nn2, d_in = dd2, valid_in = valid_out2;
            v -= 2;
            continue;
        }
        const u32 blocks = (u32)std::min<u64>((valid_out + 255) / 256, 4096);
        if (base)
            LM_LAUNCH(ctx, k_gkr_layer_up<true>, dim3(blocks), dim3(256), 0, n_in, d_in, m, nn, dd, valid_in, valid_out);
        else
            LM_LAUNCH(ctx, k_gkr_layer_up<false>, dim3(blocks), dim3(256), 0, n_in, d_in, m, nn, dd, valid_in, valid_out);
        n_in = nn;
        d_in = dd;
        valid_in = valid_out;
        v -= 1;
    }
    // work buffers: the first launch that writes folds the biggest layer (2^(n_vars-1) per array) by two challenges
    g->work_words = std::max<u64>(20ull << (n_vars - 3), 256);
    const u64 w1 = std::max<u64>(g->work_words / 4, 256);
    if (lm_pool_alloc_t(ctx, &g->work[0], g->work_words * 4) != hipSuccess ||
        lm_pool_alloc_t(ctx, &g->work[1], w1 * 4) != hipSuccess ||
        lm_pool_alloc_t(ctx, &g->eqt.d_buf, PrefixEqTables::words_needed(n_vars) * 4) != hipSuccess ||
        lm_pool_alloc_t(ctx, &g->d_merge, GKR_TAIL_MERGE_WORDS * 4) != hipSuccess) {
        lm_set_error("lm_gkr_build: device allocation failed (work)");
        lm_gkr_free(ctx, g);
        return LM_E_NOMEM;
    }
    g->eqt.buf_words = PrefixEqTables::words_needed(n_vars);
    if (hipGetLastError() != hipSuccess) {
        lm_set_error("lm_gkr_build: kernel launch failed");
        lm_gkr_free(ctx, g);
        return LM_E_DEVICE;
    }
    *out = g;
    return LM_OK;
}

int lm_gkr_top(lm_ctx* ctx, const lm_gkr* g, uint32_t* nums32, uint32_t* dens32) {
    LM_REQUIRE(ctx && g && nums32 && dens32);
    u32 soa[2][160];
    int rc = lm_fetch_words(ctx, -1, g->nums.back(), 160, g->dens.back(), 160, 0, &soa[0][0]);
    if (rc) return rc;
    const u64 valid = g->valid.back();
    for (int i = 0; i < 32; i++)
        for (int k = 0; k < 5; k++) {
            const bool pad = (u64)i >= valid;  // never written: the neutral pair (0, 1)
            nums32[i * 5 + k] = pad ? 0u : soa[0][k * 32 + i];
            dens32[i * 5 + k] = pad ? (k == 0 ? ONE : 0u) : soa[1][k * 32 + i];
        }
    return LM_OK;
}

// Start the sumcheck of the layer with 2^(K+1) entries (K = number of coordinates of the claim point, 5 <= K < n_vars).
int lm_gkr_layer_begin(lm_ctx* ctx, lm_gkr* g, uint32_t K, const uint32_t* point, const uint32_t alpha[5]) {
    LM_REQUIRE(ctx && g && point && alpha && K >= 5 && K < g->n_vars);
    gkr_ahead_drop(ctx, g);
    gkr_tail_dismiss(ctx, g);
    g->ahead_ok = gkr_launch_ahead_enabled(ctx);
    g->K = K;
    g->round = 0;
    g->cur = -1;
    g->m = 1ull << K;  // length of each of n_l, n_r, d_l, d_r
    memcpy(g->alpha.v, alpha, 20);
    g->point.resize(K);
    for (u32 j = 0; j < K; j++) memcpy(g->point[j].v, point + 5 * j, 20);
    g->pending.clear();
    g->history.clear();
    g->la_valid = false;
    g->fin_m = 0;
    // round t uses eq over point[0 .. p), p = K-1-t
    return g->eqt.build(ctx, point, K);
}


// One round.  prev_r = NULL on the first round of the layer; afterwards the challenge of the previous round.
// out = (c0_raw, c2_raw) of finalize_round (sumcheck_utils.rs:90-109), padding included because the vectors are fully
// materialised.  Every other call is answered on the host from the look-ahead sums of the previous launch (see k_gkr_step);
// the challenges are folded into the arrays two at a time by the next launch.
// One launch of a layer's schedule (a step over the arrays, or the first launch of a resident tail): everything but the two challenges
// it folds is known one launch earlier — the state the previous launch leaves (cur, m, arr_valid) and the round index —, so it can be
// enqueued AHEAD, behind that launch, and receive (r0, r1) as a message (lm_mail_*; mail_no != 0).
static int gkr_shoot(lm_ctx* ctx, lm_gkr* g, u32 t, u32 F, int cur, u64 m, u64 arr_valid, const EF& r0, const EF& r1, u32 mail_no, GkrShot* out) {
    LM_REQUIRE(F == 0 || F == 2);  // the schedule: first launch of a layer has nothing to fold, later ones two challenges
    LM_REQUIRE(mail_no == 0 || F == 2);
    const u64 m_out = m >> F;
    LM_REQUIRE(m_out >= 2 && m_out == (2ull << (g->K - 1 - t)));
    const bool la = m_out >= 4;
    const u32 p = g->K - 1 - t;  // round t: 2^p pairs, eq over point[0..p)
    const EqSplit eq = g->eqt.at(la ? p - 1 : p);
    // valid inputs: storage entries of the layer (array index y <-> entries 2y, 2y + 1) or entries of the work arrays
    const bool input_layer = g->K == g->n_vars - 1;
    const u64 valid_in = cur < 0 ? (input_layer ? g->valid0 : g->valid[g->n_vars - g->K - 2]) : arr_valid;
    const u64 arr_in_valid = cur < 0 ? valid_in / 2 : valid_in;                      // in array entries
    u64 n_threads = round_up8((arr_in_valid + (1ull << F) - 1) >> F, m_out);     // outputs that are computed (rest: closed form)
    const int dst = cur < 0 ? 0 : 1 - cur;
    // the layer being proven has 2^(K+1) entries: the caller's input when K + 1 == n_vars, else owned layer
    // nums[i] (2^(n_vars-1-i) entries) with i = n_vars - K - 2
    const u32* n_st = input_layer ? g->d_nums0 : g->nums[g->n_vars - g->K - 2];
    const u32* d_st = input_layer ? g->d_dens0 : g->dens[g->n_vars - g->K - 2];
    const u32* nul = nullptr;
    const u32* h_mail = lm_mail_line(ctx);
    u32* d_relay = ctx->d_relay;
    GkrShot sh;
    sh.t = t, sh.F = F, sh.m_out = m_out, sh.la = la, sh.p = p, sh.dst = dst, sh.tail = false, sh.mail_no = mail_no;
    sh.tail_W = 1, sh.tail_S = 0, sh.tail_solo = true;
    u32 seq;
    if (gkr_tail_enabled() && !g->no_resident && ctx->h_cmd && m_out <= gkr_tail_max_entries() && m_out >= 8 &&
        gkr_tail_reserve(ctx, (u32)std::max<u64>(1, m_out / GKR_TAIL_SLICE))) {
        // the reservation is given back on every early return below (a leaked one would silently push later layers onto launches)
        struct TailReservation {
            lm_ctx* ctx;
            int n;
            bool keep = false;
            ~TailReservation() {
                if (!keep) gkr_tail_unreserve(ctx, n);
            }
        } reservation{ctx, (int)std::max<u64>(1, m_out / GKR_TAIL_SLICE)};
        seq = ++ctx->res_seq;
        GkrTailEq eqs;
        {
            u32 s = 0;
            for (u64 mo = m_out; s < GKR_TAIL_STEPS; mo >>= 2, s++) {
                // step s answers round t + 2 s: 2^(p - 2 s) pairs
                const u32 ps = p - 2 * s;
                eqs.e[s] = g->eqt.at(mo >= 4 ? ps - 1 : ps);
                if (mo <= 4) break;
            }
            LM_REQUIRE(s < GKR_TAIL_STEPS);
        }
        // no stale message, no stale dismissal; the payload words carry the parity the FIRST message will NOT have (that of this
        // launch's own sequence number), so a poll that sees part of the line before and part after the host's stores is refused
        // (enqueued ahead, this happens while the previous launch — a step, never a tail — is still running: nothing polls line 0 then)
        for (u32 i = 0; i < lm_ctx::CMD_LINE_WORDS; i++) ((volatile u32*)ctx->h_cmd)[i] = i < 10 ? (seq & 1) << 31 : 0;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        sh.tail_W = (u32)std::max<u64>(1, m_out / GKR_TAIL_SLICE);
        sh.tail_S = (u32)(m_out / sh.tail_W);
        sh.tail_solo = sh.tail_W == 1;
#define GKR_TAIL(MODE, FF, NI, DI, AI) \
    LM_LAUNCH(ctx, (k_gkr_tail<MODE, FF>), dim3(sh.tail_W), dim3(GKR_TAIL_THREADS), 0, NI, DI, AI, (u32)m_out, valid_in, r0, r1, g->alpha, eqs, ctx->h_res, seq, (const u32*)ctx->h_cmd, g->d_merge, ctx->d_sync + 1, h_mail, d_relay, mail_no)
        if (cur < 0) {
            if (F == 0) {
                if (input_layer)
                    GKR_TAIL(0, 0, n_st, d_st, nul);
                else
                    GKR_TAIL(1, 0, n_st, d_st, nul);
            } else {
                if (input_layer)
                    GKR_TAIL(0, 2, n_st, d_st, nul);
                else
                    GKR_TAIL(1, 2, n_st, d_st, nul);
            }
        } else {
            LM_REQUIRE(F == 2);
            GKR_TAIL(2, 2, nul, nul, (const u32*)g->work[cur]);
        }
#undef GKR_TAIL
        LM_HIP(hipGetLastError());
        reservation.keep = true;  // from here gkr_tail_release / gkr_tail_dismiss / gkr_ahead_drop give it back
        sh.tail = true;
        n_threads = m_out;  // the resident workgroups materialise every entry (padding included)
    } else {
    const u32 blocks = (u32)std::min<u64>((n_threads + 255) / 256, 1024);
    seq = ++ctx->res_seq;
    u32* counter = ctx->d_sync + 1;
    // HBM-bound launches (>= 2^20 outputs) are recorded under their own profile name (bench.py's live HBM line): algorithmic bytes =
    // the valid inputs read once (input layer: 4 + 20 bytes per entry, owned layer: 20 + 20, work arrays: 4 x 20 per array entry) + the
    // four folded arrays written once
    const bool big = n_threads >= (1ull << 20);
    const u64 alg_bytes = (cur < 0 ? (input_layer ? 24ull : 40ull) * valid_in : 80ull * std::min<u64>(arr_in_valid, m_out << F)) + (F ? 80ull * n_threads : 0);
#define k_gkr_step_big k_gkr_step
#define GKR_STEP(MODE, FF, LL, NI, DI, AI)                                                                                                                     \
    do {                                                                                                                                                      \
        if (big) {                                                                                                                                            \
            /* (LM_LAUNCH_ON directly: one more macro level would expand the alias before it is stringified) */                                            \
            LM_LAUNCH_ON(ctx, (ctx)->stream, (k_gkr_step_big<MODE, FF, LL>), dim3(blocks), dim3(256), 0, NI, DI, AI, m_out, n_threads, valid_in, r0, r1, g->alpha, eq, g->work[dst], ctx->d_acc, counter, ctx->h_res, seq, h_mail, d_relay, mail_no); \
            LM_PROF_BYTES(ctx, k_gkr_step_big, alg_bytes);                                                                                                    \
        } else                                                                                                                                                \
            LM_LAUNCH(ctx, (k_gkr_step<MODE, FF, LL>), dim3(blocks), dim3(256), 0, NI, DI, AI, m_out, n_threads, valid_in, r0, r1, g->alpha, eq, g->work[dst], ctx->d_acc, counter, ctx->h_res, seq, h_mail, d_relay, mail_no); \
    } while (0)
    if (cur < 0) {
        LM_REQUIRE(la);  // K >= 5: the launches that read layer storage always cover two rounds
        if (F == 0) {
            if (input_layer)
                GKR_STEP(0, 0, true, n_st, d_st, nul);
            else
                GKR_STEP(1, 0, true, n_st, d_st, nul);
        } else {
            if (input_layer)
                GKR_STEP(0, 2, true, n_st, d_st, nul);
            else
                GKR_STEP(1, 2, true, n_st, d_st, nul);
        }
    } else {
        LM_REQUIRE(F == 2);
        if (la)
            GKR_STEP(2, 2, true, nul, nul, (const u32*)g->work[cur]);
        else
            GKR_STEP(2, 2, false, nul, nul, (const u32*)g->work[cur]);
    }
#undef GKR_STEP
#undef k_gkr_step_big
    LM_HIP(hipGetLastError());
    }
    sh.seq = seq;
    sh.n_threads = n_threads;
    *out = sh;
    return LM_OK;
}
// a launch enqueued ahead whose challenges will never be posted (error paths, an abandoned layer): dismiss it, give its tail slots back
static void gkr_ahead_drop(lm_ctx* ctx, lm_gkr* g) {
    if (!g->ahead) return;
    (void)lm_mail_abort(ctx);
    if (g->ahead_shot.tail) {
        (void)hipMemsetAsync(ctx->d_sync + 1, 0, 4, ctx->stream);
        gkr_tail_unreserve(ctx, (int)g->ahead_shot.tail_W);
    }
    g->ahead = false;
}
// Launch-ahead is for a prover that has the device to itself.  A launch that waits for its message holds its wave slots while it
// waits; with ten provers in flight the chip never drains, and a 1024-thread workgroup of a resident tail — which needs sixteen free
// wave slots on ONE compute unit — can be starved of a slot until the rest of its tail gives up (seen once in 1200 proofs of ten
// concurrent provers: "sequence never published", after the 3 s timeout; never with one launch per exchange).  So: only when this is
// the process's only context and no other process is registered on the device (the tails' shared counter knows them).
// LM_GKR_NO_AHEAD=1 switches it off altogether (A/B measurements).
static bool gkr_launch_ahead_enabled(lm_ctx* ctx) {
    static const bool on = getenv("LM_GKR_NO_AHEAD") == nullptr;
    if (!on || lm_ctx_live_count() != 1) return false;
    // LM_GKR_AHEAD_ASSUME_ALONE=1: the operator's word that no other prover process shares the device (also what the tests use to
    // exercise launch-ahead from a child process while the parent holds a context)
    static const bool assume = getenv("LM_GKR_AHEAD_ASSUME_ALONE") != nullptr;
    if (assume) return true;
    TailCounter* c = tail_counter(ctx->device);
    // No shared counter (shm unavailable or not private to this user, LM_GKR_TAIL_PROCESS_LOCAL): other provers cannot be seen, so this
    // one does not assume it is alone (round-5 advisor finding).
    if (!c->sh) return false;
    const int me = (int)getpid();
    for (auto& sl : c->sh->slots) {
        const int pid = sl.pid.load(std::memory_order_relaxed);
        if (pid && pid != me && !(kill(pid, 0) != 0 && errno == ESRCH)) return false;  // (a slot of a process that is gone does not count)
    }
    return true;
}
// lm_ctx_create: a process takes its slot when it creates its first context on the device, not at its first GKR layer — another prover's
// gate sees it from then on.  (The gate is evaluated per layer AND before every launch that is enqueued ahead.)
}  // extern "C"
void lm_gkr_register_process(int device) { (void)tail_counter(device); }
int lm_gkr_foreign_processes(int device) {
    TailCounter* c = tail_counter(device);
    if (!c->sh) return 0;
    const int me = (int)getpid();
    int n = 0;
    for (auto& sl : c->sh->slots) {
        const int pid = sl.pid.load(std::memory_order_relaxed);
        if (pid && pid != me && !(kill(pid, 0) != 0 && errno == ESRCH)) n++;
    }
    return n;
}
extern "C" {

// LM_GKR_FAULT=tail:<n> / ahead:<n> (tests): the n-th message of that kind of this object is NOT sent — the resident kernel that waits for
// it gives up after its 3 s, exactly what a kernel that never got its wave slots looks like to the host
static bool gkr_fault(const char* kind, u32 count) {
    static const char* e = getenv("LM_GKR_FAULT");
    if (!e) return false;
    const size_t n = strlen(kind);
    return strncmp(e, kind, n) == 0 && e[n] == ':' && (u32)atoi(e + n + 1) == count;
}
static int gkr_round_impl(lm_ctx* ctx, lm_gkr* g, const uint32_t* prev_r, uint32_t out_c0_c2[10]) {
    LM_REQUIRE(ctx && g && out_c0_c2 && g->round < g->K);
    LM_REQUIRE((g->round == 0) == (prev_r == nullptr));
    if (prev_r) g->pending.push_back(host_ef(prev_r));
    if (g->la_valid) {
        const EF& r = g->pending.back();
        const EF c0 = quad_at(g->la[0], g->la[1], g->la[2], r), c2 = quad_at(g->la[3], g->la[4], g->la[5], r);
        memcpy(out_c0_c2, c0.v, 20);
        memcpy(out_c0_c2 + 5, c2.v, 20);
        g->la_valid = false;
        g->round++;
        return LM_OK;
    }
    const u32 t = g->round;
    const u32 F = (u32)g->pending.size();
    LM_REQUIRE(F == 0 || F == 2);
    const EF r0 = F ? g->pending[0] : ef_zero(), r1 = F ? g->pending[1] : ef_zero();
    int rc;
    GkrShot sh;
    if (g->tail_live) {
        // the resident workgroup holds the arrays in LDS: hand it the two challenges (bit 31 = parity of the sequence number
        // of its next publication, see k_gkr_tail) and wait for that publication
        LM_REQUIRE(F == 2 && ctx->res_seq == g->tail_seq && !g->ahead);
        sh.t = t, sh.F = F, sh.m_out = g->m >> F, sh.la = sh.m_out >= 4, sh.p = g->K - 1 - t, sh.dst = g->cur < 0 ? 0 : 1 - g->cur;
        sh.tail = true, sh.mail_no = 0;
        LM_REQUIRE(sh.m_out >= 2 && sh.m_out == (2ull << (g->K - 1 - t)));
        sh.seq = ++ctx->res_seq;
        volatile u32* cmd = ctx->h_cmd;
        const u32 tag = (sh.seq & 1) << 31;
        for (int k = 0; k < 5; k++) cmd[k] = r0.v[k] | tag, cmd[5 + k] = r1.v[k] | tag;
        if (!gkr_fault("tail", ++g->exchanges_tail)) cmd[10] = sh.seq;
        sh.n_threads = sh.m_out;  // the resident workgroup materialises every entry (padding included)
        const u32 Sn = g->tail_S >> 2;
        if (!g->tail_solo && Sn < 4) {
            g->tail_solo = true;
            g->tail_S = g->tail_W * Sn;
        } else {
            g->tail_S = Sn;
        }
    } else if (g->ahead) {
        // this launch is already on the stream, behind the previous one: its two challenges go out as a message
        sh = g->ahead_shot;
        g->ahead = false;
        if (sh.t != t || sh.F != F) {
            g->ahead = true;
            gkr_ahead_drop(ctx, g);
            lm_set_error("lm_gkr_round: the launch enqueued ahead does not match the round being asked for");
            return LM_E_INVALID;
        }
        if (!gkr_fault("ahead", ++g->exchanges_ahead)) lm_mail_post(ctx, sh.mail_no, r0.v, r1.v);
        if (sh.tail) {
            g->tail_live = true;
            g->tail_W = sh.tail_W, g->tail_S = sh.tail_S, g->tail_solo = sh.tail_solo;
        }
    } else {
        if ((rc = gkr_shoot(ctx, g, t, F, g->cur, g->m, g->arr_valid, r0, r1, 0, &sh))) return rc;
        if (sh.tail) {
            g->tail_live = true;
            g->tail_W = sh.tail_W, g->tail_S = sh.tail_S, g->tail_solo = sh.tail_solo;
        }
    }
    // The launch after this one (two rounds on: F = 2) reads what this one leaves; enqueue it now, behind this one, when this one is a
    // launch (a live tail takes messages instead) and the layer has rounds left for it
    if (!sh.tail && t + 2 < g->K && g->ahead_ok && !g->no_resident && gkr_launch_ahead_enabled(ctx)) {
        const u32 no = lm_mail_reserve(ctx);
        rc = gkr_shoot(ctx, g, t + 2, 2, F ? sh.dst : g->cur, sh.m_out, F ? sh.n_threads : g->arr_valid, ef_zero(), ef_zero(), no, &g->ahead_shot);
        if (rc) {
            (void)lm_mail_abort(ctx);
            return rc;
        }
        g->ahead = true;
    }
    const u32 seq = sh.seq, p = sh.p;
    const u64 m_out = sh.m_out, n_threads = sh.n_threads;
    const bool la = sh.la;
    const int dst = sh.dst;
    if ((rc = lm_wait_result(ctx, seq))) {
        gkr_ahead_drop(ctx, g);
        gkr_tail_dismiss(ctx, g);
        return rc;
    }
    u32 h_sum[GKR_SUM_WORDS];
    const u32* h = ctx->h_res;
    if (g->tail_live) {
        g->tail_seq = seq;
        if (m_out <= 4) gkr_tail_release(ctx, g);  // the workgroup returned after this publication
        if (!g->tail_solo) {
            // every workgroup published its own partial sums: wait for the other flags, add the slots
            memcpy(h_sum, ctx->h_res, sizeof h_sum);
            for (u32 w = 1; w < g->tail_W; w++) {
                volatile u32* slot = ctx->h_res + GKR_TAIL_SLOT_AT + w * GKR_TAIL_SLOT_WORDS;
                u64 spins = 0;
                while (slot[GKR_TAIL_SLOT_WORDS - 1] != seq) {
                    if (++spins > (1ull << 28)) {
                        lm_set_error("lm_gkr_round: workgroup %u of the resident tail never published sequence %u", w, seq);
                        gkr_tail_dismiss(ctx, g);
                        return LM_E_DEVICE;
                    }
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                for (u32 t = 0; t < GKR_SUM_WORDS; t++) h_sum[t] = add(h_sum[t], slot[t]);
            }
            h = h_sum;
        }
    }
    auto S = [&](u32 cls, u32 slot) { return host_ef(h + cls * 15 + slot * 5); };  // slot 0: e, 1: X, 2: Y
    EF c0, c2;
    if (la) {
        const EF pt = g->point[p - 1], omp = ef_sub(ef_one(), pt);
        // quads beyond n_threads are pure padding: every entry is (0, alpha, 1, 1), so e = alpha and all differences vanish;
        // their weights add up to the MLE of "zeros then ones" at the eq point (sumcheck_utils.rs:136,331 use the same closed form)
        EF pad = ef_zero();
        if (n_threads < m_out) {
            // sum_{j >= n_zeros} eq(point[0..p-1), j) = mle_of_zeros_then_ones (poly/src/mle/mle_custom.rs:4-19), iteratively:
            // walking n_zeros' bits from the most significant, every index that agrees so far and has a 1 where n_zeros has a 0 is larger
            const u64 n_zeros = n_threads / 4;
            const u32 nb = p - 1;
            EF acc = ef_zero(), prefix = ef_one();
            for (u32 b = 0; b < nb; b++) {
                const EF x = g->point[b];
                if ((n_zeros >> (nb - 1 - b)) & 1) {
                    prefix = ef_mul(prefix, x);
                } else {
                    acc = ef_add(acc, ef_mul(prefix, x));
                    prefix = ef_mul(prefix, ef_sub(ef_one(), x));
                }
            }
            acc = ef_add(acc, prefix);  // the index n_zeros itself
            pad = ef_mul(g->alpha, acc);
        }
        const EF E0 = ef_add(S(0, 0), pad), E1 = ef_add(S(1, 0), pad), E2 = ef_add(S(2, 0), pad), C01 = S(1, 1), C23 = S(3, 1), T2 = S(0, 1),
                 T0 = S(2, 1), T3 = S(0, 2);
        c0 = ef_add(ef_mul(omp, E0), ef_mul(pt, E2));
        c2 = ef_add(ef_mul(omp, C01), ef_mul(pt, C23));
        g->la[0] = E0, g->la[1] = ef_sub(ef_sub(E1, E0), C01), g->la[2] = C01;
        g->la[3] = T0, g->la[4] = ef_sub(ef_sub(T3, T0), T2), g->la[5] = T2;
        g->la_valid = true;
    } else {
        c0 = ef_add(S(0, 0), S(2, 0));
        c2 = ef_add(S(1, 1), S(3, 1));
    }
    if (F) {
        g->cur = dst;
        g->m = m_out;
        g->arr_valid = n_threads;
        g->pending.clear();
        if (m_out <= 4) {
            g->fin_m = (u32)m_out;
            for (int q = 0; q < 4; q++)
                for (int k = 0; k < 5; k++)
                    for (u32 i = 0; i < m_out; i++) g->fin[q][i].v[k] = ctx->h_res[GKR_FIN_AT + (q * 5 + k) * 4 + i];
        }
    }
    memcpy(out_c0_c2, c0.v, 20);
    memcpy(out_c0_c2 + 5, c2.v, 20);
    g->round++;
    return LM_OK;
}

// Fail soft.  A resident tail needs all its workgroups on the chip at once and a launch enqueued ahead holds its wave slots while it
// waits: on a device shared with kernels this library knows nothing about (another tenant, a profiler, RCCL) either can be starved
// until its own timeout, and the host then sees "sequence never published".  That is a scheduling hiccup, not a prover error
// (errors are for invalid witnesses): the resident kernels are dismissed, the layer is re-run from its storage — which
// no launch overwrites — with one launch per exchange and the challenges it has already received (same arithmetic, same sums: the
// transcript does not notice), and the event is counted (lm_soft_fallbacks).  A second failure is reported.
int lm_gkr_round(lm_ctx* ctx, lm_gkr* g, const uint32_t* prev_r, uint32_t out_c0_c2[10]) {
    LM_REQUIRE(ctx && g && out_c0_c2 && g->round < g->K);
    LM_REQUIRE((g->round == 0) == (prev_r == nullptr));
    if (prev_r) g->history.push_back(host_ef(prev_r));
    const bool resident_in_play = g->tail_live || g->ahead || (gkr_tail_enabled() && !g->no_resident) || (g->ahead_ok && !g->no_resident);
    int rc = gkr_round_impl(ctx, g, prev_r, out_c0_c2);
    if (rc != LM_E_DEVICE || g->no_resident || !resident_in_play) return rc;
    static const bool soft = getenv("LM_GKR_NO_SOFT_FALLBACK") == nullptr;
    if (!soft) return rc;
    // (gkr_round_impl has dismissed the tail / the launch enqueued ahead and synchronised the stream on its error path)
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return rc;  // a real device error: nothing to re-run on
    g->no_resident = true;
    ctx->soft_fallbacks++;
    {
        const double now = gkr_now_s();
        ctx->no_resident_backoff_s = now < ctx->no_resident_until_s + 10.0 && ctx->no_resident_backoff_s > 0 ? std::min(64.0, 2 * ctx->no_resident_backoff_s) : 2.0;
        ctx->no_resident_until_s = now + ctx->no_resident_backoff_s;
    }
    lm_mail_reset(ctx);
    if (hipMemsetAsync(ctx->d_acc, 0, LM_ACC_WORDS * sizeof(unsigned long long), ctx->stream) != hipSuccess ||
        hipMemsetAsync(ctx->d_sync + 1, 0, 4, ctx->stream) != hipSuccess)
        return rc;
    g->round = 0, g->cur = -1, g->m = 1ull << g->K, g->arr_valid = 0, g->fin_m = 0;
    g->pending.clear();
    g->la_valid = false;
    const size_t n = g->history.size();  // the call that failed asked for round n
    u32 scratch[10];
    for (size_t i = 0; i <= n; i++) {
        const int rc2 = gkr_round_impl(ctx, g, i ? g->history[i - 1].v : nullptr, i == n ? out_c0_c2 : scratch);
        if (rc2) return rc2;
    }
    return LM_OK;
}

// After the last round: fold by the remaining challenges and return [n_l, n_r, d_l, d_r] (4 EF, mod.rs:129).  The last
// launch left the <= 4 remaining entries of each array with the host; folding them is two EF multiplications per array.
int lm_gkr_layer_end(lm_ctx* ctx, lm_gkr* g, const uint32_t last_r[5], uint32_t inner_evals[20]) {
    LM_REQUIRE(ctx && g && last_r && inner_evals && g->round == g->K);
    g->pending.push_back(host_ef(last_r));
    LM_REQUIRE(g->fin_m == (1u << g->pending.size()) && g->fin_m == g->m);
    EF v[4];
    for (int q = 0; q < 4; q++) {
        EF a[4];
        u32 n = g->fin_m;
        for (u32 i = 0; i < n; i++) a[i] = g->fin[q][i];
        for (const EF& r : g->pending) {
            n /= 2;
            for (u32 i = 0; i < n; i++) a[i] = ef_add(a[2 * i], ef_mul(r, ef_sub(a[2 * i + 1], a[2 * i])));
        }
        v[q] = a[0];
    }
    v[1] = ef_sub(v[1], ef_mul(g->alpha, v[3]));  // the arrays hold nr + alpha dr
    for (int q = 0; q < 4; q++) memcpy(inner_evals + 5 * q, v[q].v, 20);
    g->pending.clear();
    g->m = 1;
    g->fin_m = 0;
    return LM_OK;
}

}  // extern "C"
