#!/usr/bin/env python3
"""A SECOND, independent restatement of the pieces of the hot path above Poseidon — pure Python integers, written from the
reference's Rust (not from oracle/*.hpp), run ONLY in the build container (it parses the Poseidon constants out of
/root/reference) by `python tests/golden/twin_r03.py`, which writes tests/golden/vectors_r03.json.  tests/test_golden_r03.py
checks BOTH the C++ oracle and the device against those vectors (SURVEY.md §8(c) "fixtures to commit").

What is restated, each in its textbook form (the reference's optimised forms are equivalent — SURVEY.md §8(c) checked the
Poseidon schedule against the KAT):
  field / extension     koala-bear: p = 2^31 - 2^24 + 1, F_p[X]/(X^5 + X^2 - 1) (quintic_extension/extension.rs:531-548)
  Poseidon1-16          4 + 20 + 4 rounds, x^3, dense circulant MDS, constants parsed from poseidon1_koalabear_16.rs:22,699-815;
                        checked against the KAT :1083-1091 before anything else is produced
  compress / sponge     perm(x) + x; hash_rtl_iter / precompute_zero_suffix_state (symetric/src/sponge.rs:27-108)
  commit matrix         prepare_evals_for_fft_unpacked (whir/src/utils.rs:128-150)
  evals-DFT             by its DEFINITION, the identity of the reference's own test (whir/src/dft.rs:583-603): row i of a column =
                        the multilinear extension of the column evaluated at (g^i, g^2i, g^4i, ...), g = two_adic_generator(log h)
  Merkle tree           first_digest_layer{,_with_initial_state} (whir/src/merkle.rs:215-287), compress_layer, open_siblings
                        (symetric/src/merkle.rs:21-90)
  challenger            overwrite-mode duplex, sample, sample_vec, sample_in_range, add_sumcheck_polynomial with eq factor,
                        pow_grinding with the smallest witness (fiat-shamir/src/{challenger,prover,utils}.rs)
  GKR                   prove_gkr_quotient (sub_protocols/src/quotient_gkr/mod.rs:31-141) in natural order: layers, top 32 + 32
                        values, per layer alpha, K rounds LSB first (bare polynomial from three evaluations), inner evaluations, beta
  product sumcheck      c0, c1, c2 per round, MSB-first folds (sumcheck/src/product_computation.rs:127-169,242-315)
"""
import json
import os
import re
import sys

P = 0x7F000001
R = 1 << 32
REF = "/root/reference/crates/backend/koala-bear/src"


def to_monty(x):
    return (x % P) * R % P


# ---------------------------------------------------------------------------------------------------------------- extension
def ef(*c):
    c = list(c) + [0] * (5 - len(c))
    return tuple(x % P for x in c)


EF0, EF1 = ef(0), ef(1)


def ef_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def ef_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def ef_scale(a, s):
    return tuple(x * s % P for x in a)


def ef_mul(a, b):  # schoolbook product reduced by X^5 = 1 - X^2
    t = [0] * 9
    for i in range(5):
        for j in range(5):
            t[i + j] += a[i] * b[j]
    for k in range(8, 4, -1):  # X^k = X^(k-5) - X^(k-3)
        t[k - 5] += t[k]
        t[k - 3] -= t[k]
        t[k] = 0
    return tuple(x % P for x in t[:5])


def ef_pow(a, e):
    r = EF1
    while e:
        if e & 1:
            r = ef_mul(r, a)
        a = ef_mul(a, a)
        e >>= 1
    return r


def ef_inv(a):
    return ef_pow(a, P ** 5 - 2)


def ef_monty(a):
    return [to_monty(x) for x in a]


# ---------------------------------------------------------------------------------------------------------------- Poseidon1-16
def parse_constants():
    src = open(os.path.join(REF, "poseidon1_koalabear_16.rs")).read()
    col = [int(x) for x in re.search(r"const MDS_CIRC_COL[^=]*=\s*KoalaBear::new_array\(\[([^\]]*)\]", src).group(1).split(",") if x.strip()]
    body = src[src.index("const POSEIDON1_RC:"):]
    body = body[:body.index("]);")]
    words = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", body)]
    assert len(words) == 28 * 16, len(words)
    rc = [words[16 * r:16 * r + 16] for r in range(28)]
    assert len(col) == 16
    return col, rc


MDS_COL, RC = parse_constants() if os.path.exists(REF) else (None, None)


def mds(s):
    return [sum(MDS_COL[(i - j) % 16] * s[j] for j in range(16)) % P for i in range(16)]


def permute(s):
    s = list(s)
    r = 0
    for _ in range(4):
        s = mds([pow((x + c) % P, 3, P) for x, c in zip(s, RC[r])])
        r += 1
    for _ in range(20):
        s = [(x + c) % P for x, c in zip(s, RC[r])]
        s[0] = pow(s[0], 3, P)
        s = mds(s)
        r += 1
    for _ in range(4):
        s = mds([pow((x + c) % P, 3, P) for x, c in zip(s, RC[r])])
        r += 1
    return s


def compress(s):
    return [(a + b) % P for a, b in zip(permute(s), s)]


def check_kat():
    src = open(os.path.join(REF, "poseidon1_koalabear_16.rs")).read()
    tail = src[src.rindex("assert_eq!("):]
    nums = [int(x) for x in re.findall(r"\b\d{6,10}\b", tail)][:16]
    assert len(nums) == 16
    got = permute(list(range(16)))
    assert got == nums, "the textbook Poseidon1 schedule does not reproduce the reference KAT"


# ---------------------------------------------------------------------------------------------------------------- sponge, Merkle
def hash_rtl(row, full_width):
    """hash_rtl_iter over a row zero-padded to full_width (a multiple of 8, >= 16): the state is the LAST 16 words, then the
    chunks of 8 to the left overwrite state[8..16], each followed by a compression"""
    data = list(row) + [0] * (full_width - len(row))
    state = compress(data[-16:])
    for off in range(full_width - 24, -1, -8):
        state = compress(state[:8] + data[off:off + 8])
    return state[:8]


def zero_suffix_state(n_zero_chunks):
    state = compress([0] * 16)
    for _ in range(n_zero_chunks - 2):
        state = compress(state[:8] + [0] * 8)
    return state


def hash_rtl_with_initial_state(row, effective_width, state):
    """first_digest_layer_with_initial_state: the row's first effective_width words, padded to a multiple of 8, absorbed right to left"""
    data = list(row[:effective_width]) + [0] * ((-effective_width) % 8)
    for off in range(len(data) - 8, -1, -8):
        state = compress(state[:8] + data[off:off + 8])
    return state[:8]


def merkle_tree(rows, full_width, effective_width):
    """build_merkle_tree_koalabear (whir/src/merkle.rs:59-88) -> layers bottom-up (leaf digests first)"""
    n_zero = (full_width - effective_width) // 8
    if n_zero >= 2:
        st = zero_suffix_state(n_zero)
        layer = [hash_rtl_with_initial_state(r, effective_width, st) for r in rows]
    else:
        layer = [hash_rtl(r, full_width) for r in rows]
    layers = [layer]
    while len(layer) > 1:
        layer = [compress(layer[2 * i] + layer[2 * i + 1])[:8] for i in range(len(layer) // 2)]
        layers.append(layer)
    return layers


def open_siblings(layers, index):
    out = []
    for lv in layers[:-1]:
        out.append(lv[index ^ 1])
        index >>= 1
    return out


# ---------------------------------------------------------------------------------------------------------------- commit matrix + DFT
TWO_ADIC_GENERATORS = None


def two_adic_generator(bits):
    src = open(os.path.join(REF, "koala_bear.rs")).read()
    m = re.search(r"const TWO_ADIC_GENERATORS[^=]*=\s*&KoalaBear::new_array\(\[([^\]]*)\]", src, re.S)
    vals = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
    g = vals[bits]
    assert pow(g, 1 << bits, P) == 1 and (bits == 0 or pow(g, 1 << (bits - 1), P) != 1)
    return g


def eq_weights(point):
    """eq(point, i) for all i, point[0] <-> most significant bit of i (poly/src/eq_mle.rs, evals.rs:142-347)"""
    w = [1]
    for x in point:
        w = [u for v in w for u in (v * ((1 - x) % P) % P, v * x % P)]
    return w


def mle_eval(vals, point):
    w = eq_weights(point)
    return sum(a * b for a, b in zip(vals, w)) % P


def commit_matrix(evals, folding_factor, log_inv_rate, dft_n_cols):
    """prepare_evals_for_fft_unpacked + the evals-DFT of every column -> rows (h x dft_n_cols)"""
    n_blocks = 1 << folding_factor
    full_len = len(evals) << log_inv_rate
    block = full_len // n_blocks
    log_block = block.bit_length() - 1
    mat = [[evals[((c << log_block) + r) >> log_inv_rate] for c in range(dft_n_cols)] for r in range(block)]
    g = two_adic_generator(log_block)
    out = [[0] * dft_n_cols for _ in range(block)]
    for i in range(block):
        y = pow(g, i, P)
        pt = [pow(y, 1 << k, P) for k in range(log_block)]   # expand_from_univariate
        w = eq_weights(pt)
        for c in range(dft_n_cols):
            out[i][c] = sum(mat[r][c] * w[r] for r in range(block)) % P
    return out


# ---------------------------------------------------------------------------------------------------------------- transcript
class Prover:
    def __init__(self):
        self.state = [0] * 16
        self.fresh = False
        self.transcript = []

    def observe(self, v8):
        self.state = permute(self.state[:8] + list(v8))
        self.fresh = True

    def observe_many(self, xs):
        xs = list(xs)
        for i in range(0, len(xs), 8):
            ch = xs[i:i + 8]
            self.observe(ch + [0] * (8 - len(ch)))

    def duplex(self):
        self.observe([0] * 8)

    def sample(self):
        assert self.fresh, "stale rate"
        self.fresh = False
        return self.state[8:]

    def sample_many(self, n):
        out = []
        for i in range(n):
            if i:
                self.duplex()
            out.append(self.sample())
        return out

    def sample_vec(self, n):
        flat = [x for b in self.sample_many((5 * n + 7) // 8) for x in b][:5 * n]
        return [tuple(flat[5 * i:5 * i + 5]) for i in range(n)]

    def sample_ef(self):
        return self.sample_vec(1)[0]

    def sample_in_range(self, bits, n):
        flat = [x for b in self.sample_many((n + 7) // 8) for x in b][:n]
        return [x & ((1 << bits) - 1) for x in flat]

    def add_base(self, xs):
        self.observe_many(xs)
        self.transcript += list(xs)

    def add_ext(self, es):
        self.add_base([x for e in es for x in e])

    def add_sumcheck_polynomial(self, coeffs, eq_alpha=None):
        flat = [x for e in coeffs for x in e]
        if eq_alpha is None:
            self.observe_many(flat)
        else:  # expand_bare_to_full: eq(alpha, X) * h(X), eq(alpha, X) = (1 - alpha) + (2 alpha - 1) X
            oma, tam = ef_sub(EF1, eq_alpha), ef_sub(ef_add(eq_alpha, eq_alpha), EF1)
            d = len(coeffs) - 1
            full = [ef_mul(oma, coeffs[0])]
            for k in range(1, d + 1):
                full.append(ef_add(ef_mul(oma, coeffs[k]), ef_mul(tam, coeffs[k - 1])))
            full.append(ef_mul(tam, coeffs[d]))
            self.observe_many([x for e in full for x in e])
        self.transcript += flat[5:]

    def pow_grinding(self, bits):
        if bits == 0:
            return None
        w = 0
        while permute(self.state[:8] + [w] + [0] * 7)[8] & ((1 << bits) - 1):
            w += 1
        self.observe_many([w])
        self.transcript.append(w)
        return w


# ---------------------------------------------------------------------------------------------------------------- GKR
def gkr_prove(nums, dens):
    """nums: base values, dens: EF values, natural order, power-of-two length -> (Prover, quotient, point, claim_num, claim_den)"""
    pr = Prover()
    layers = [([ef(n) for n in nums], list(dens))]
    while len(layers[-1][0]) > 32:
        n, d = layers[-1]
        layers.append(([ef_add(ef_mul(n[2 * j], d[2 * j + 1]), ef_mul(n[2 * j + 1], d[2 * j])) for j in range(len(n) // 2)],
                       [ef_mul(d[2 * j], d[2 * j + 1]) for j in range(len(d) // 2)]))
    top_n, top_d = layers.pop()
    pr.add_ext(top_n)
    pr.add_ext(top_d)
    quotient = EF0
    for n, d in zip(top_n, top_d):
        quotient = ef_add(quotient, ef_mul(n, ef_inv(d)))
    point = pr.sample_vec(5)

    def mle_ef(vals, pt):
        w = [EF1]
        for x in pt:
            w = [u for v in w for u in (ef_mul(v, ef_sub(EF1, x)), ef_mul(v, x))]
        acc = EF0
        for a, b in zip(vals, w):
            acc = ef_add(acc, ef_mul(a, b))
        return acc

    claim_num, claim_den = mle_ef(top_n, point), mle_ef(top_d, point)
    for n, d in reversed(layers):
        K = len(point)
        pr.duplex()
        alpha = pr.sample_ef()
        nl, nr, dl, dr = n[0::2], n[1::2], d[0::2], d[1::2]      # even_odd_split
        mmf = EF1
        challenges = []
        for t in range(K):
            eq_alpha = point[K - 1 - t]
            rest = point[:K - 1 - t]
            w = [EF1]
            for x in rest:
                w = [u for v in w for u in (ef_mul(v, ef_sub(EF1, x)), ef_mul(v, x))]
            vals = []
            for X in (0, 1, 2):
                acc = EF0
                for j in range(len(w)):
                    a = [ef_add(arr[2 * j], ef_scale(ef_sub(arr[2 * j + 1], arr[2 * j]), X)) for arr in (nl, nr, dl, dr)]
                    e = ef_add(ef_mul(alpha, ef_mul(a[2], a[3])), ef_add(ef_mul(a[0], a[3]), ef_mul(a[1], a[2])))
                    acc = ef_add(acc, ef_mul(w[j], e))
                vals.append(ef_mul(acc, mmf))
            inv2 = pow(2, P - 2, P)
            c0 = vals[0]
            c2 = ef_scale(ef_add(ef_sub(vals[2], ef_add(vals[1], vals[1])), vals[0]), inv2)
            c1 = ef_sub(ef_sub(vals[1], c0), c2)
            pr.add_sumcheck_polynomial([c0, c1, c2], eq_alpha)
            r = pr.sample_ef()
            eq_eval = ef_add(ef_mul(ef_sub(EF1, eq_alpha), ef_sub(EF1, r)), ef_mul(eq_alpha, r))
            mmf = ef_mul(mmf, eq_eval)
            nl, nr, dl, dr = ([ef_add(arr[2 * j], ef_mul(r, ef_sub(arr[2 * j + 1], arr[2 * j]))) for j in range(len(arr) // 2)]
                              for arr in (nl, nr, dl, dr))
            challenges.append(r)
        inner = [nl[0], nr[0], dl[0], dr[0]]
        pr.add_ext(inner)
        beta = pr.sample_ef()
        omb = ef_sub(EF1, beta)
        claim_num = ef_add(ef_mul(omb, inner[0]), ef_mul(beta, inner[1]))
        claim_den = ef_add(ef_mul(omb, inner[2]), ef_mul(beta, inner[3]))
        point = challenges[::-1] + [beta]
    return pr, quotient, point, claim_num, claim_den


# ---------------------------------------------------------------------------------------------------------------- product sumcheck
def product_sumcheck(f, W, challenges):
    """f: base values, W: EF values; -> per round (c0, c1, c2), and the final folded (f, W)"""
    f = [ef(x) for x in f]
    rounds = []
    S = EF0
    for a, b in zip(f, W):
        S = ef_add(S, ef_mul(a, b))
    for r in challenges:
        h = len(f) // 2
        c0 = c2 = EF0
        for i in range(h):
            c0 = ef_add(c0, ef_mul(f[i], W[i]))
            c2 = ef_add(c2, ef_mul(ef_sub(f[i + h], f[i]), ef_sub(W[i + h], W[i])))
        c1 = ef_sub(ef_sub(S, ef_add(c0, c0)), c2)
        rounds.append((c0, c1, c2))
        f = [ef_add(f[i], ef_mul(r, ef_sub(f[i + h], f[i]))) for i in range(h)]
        W = [ef_add(W[i], ef_mul(r, ef_sub(W[i + h], W[i]))) for i in range(h)]
        S = ef_add(c0, ef_mul(r, ef_add(c1, ef_mul(r, c2))))
    return rounds, f, W


# ---------------------------------------------------------------------------------------------------------------- the fixture
def splitmix(seed):
    x = seed

    def nxt():
        nonlocal x
        x = (x + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        return (z ^ (z >> 31)) % P
    return nxt


def main():
    check_kat()
    rnd = splitmix(0x1EA7)
    M = lambda xs: [to_monty(x) for x in xs]  # noqa: E731 — vectors are stored as Montgomery words (the boundary's representation)
    out = {"note": "generated by tests/golden/twin_r03.py (pure-int Python restatement written from the reference's Rust); Montgomery u32 words"}
    # ---- commits of a 2^10 polynomial: folding 6 (64 columns of 32 rows at rate 1/2); with and without a zero suffix
    for name, actual_len in (("commit_zero_suffix", 600), ("commit_dense", 1024)):
        evals = [rnd() if i < actual_len else 0 for i in range(1024)]
        fold, rate = 6, 1
        block = (1024 << rate) >> fold
        dft_cols = -(-actual_len // (1024 >> fold))      # columns that are not entirely zero (commit.rs:70-83)
        rows = commit_matrix(evals, fold, rate, dft_cols)
        layers = merkle_tree(rows, 1 << fold, dft_cols)
        idx = [0, 5, block - 1]
        out[name] = dict(evals=M(evals), actual_len=actual_len, folding_factor=fold, log_inv_rate=rate, dft_n_cols=dft_cols,
                         matrix=[M(r) for r in rows], root=M(layers[-1][0]), leaf_digests=[M(d) for d in layers[0]],
                         open_indices=idx, siblings=[[M(s) for s in open_siblings(layers, i)] for i in idx])
    # ---- sponge on short rows (both paths)
    row = [rnd() for _ in range(21)]
    out["sponge"] = dict(row=M(row), full_width=40, digest_plain=M(hash_rtl(row, 40)),
                         digest_zero_suffix=M(hash_rtl_with_initial_state(row + [0] * 3, 21, zero_suffix_state((64 - 24) // 8))), zero_suffix_full_width=64,
                         zero_suffix_state=M(zero_suffix_state(5)))
    assert hash_rtl(row, 64) == hash_rtl_with_initial_state(row, 21, zero_suffix_state(5))
    # ---- transcript primitives
    pr = Prover()
    pr.add_base([rnd() for _ in range(11)])
    s1 = pr.sample_vec(3)
    pr.duplex()
    q = pr.sample_in_range(9, 13)
    poly, poly_alpha = [ef(*[rnd() for _ in range(5)]) for _ in range(3)], ef(*[rnd() for _ in range(5)])
    pr.add_sumcheck_polynomial(poly, poly_alpha)
    s2 = pr.sample_ef()
    cap = list(pr.state[:8])
    w = pr.pow_grinding(9)
    pr.duplex()
    s3 = pr.sample_ef()
    out["transcript"] = dict(seed_note="splitmix64(0x1EA7) stream continues from the commits above", sample_vec_3=[M(e) for e in s1],
                             sample_in_range_9x13=q, poly_coeffs=[M(c) for c in poly], poly_eq_alpha=M(poly_alpha), sample_after_sumcheck_poly=M(s2), pow_capacity=M(cap), pow_bits=9, pow_witness=to_monty(w),
                             sample_after_pow=M(s3), transcript=M(pr.transcript), final_state=M(pr.state))
    # ---- GKR at log_n = 11 (the reference's smallest test shape), a consistent-free instance: random numerators / denominators
    n = 1 << 11
    nums = [rnd() for _ in range(n)]
    dens = [ef(*[rnd() for _ in range(5)]) for _ in range(n)]
    g, quotient, point, cn, cd = gkr_prove(nums, dens)
    out["gkr"] = dict(log_n=11, nums=M(nums), dens=[M(d) for d in dens], transcript=M(g.transcript), quotient=M(quotient),
                      point=[M(x) for x in point], claim_num=M(cn), claim_den=M(cd))
    # ---- product sumcheck with fixed challenges, 2^8 -> 2^3
    f = [rnd() for _ in range(256)]
    W = [ef(*[rnd() for _ in range(5)]) for _ in range(256)]
    ch = [ef(*[rnd() for _ in range(5)]) for _ in range(5)]
    rounds, ff, Wf = product_sumcheck(f, W, ch)
    out["product_sumcheck"] = dict(f=M(f), W=[M(x) for x in W], challenges=[M(x) for x in ch],
                                   rounds=[[M(c) for c in r] for r in rounds], f_final=[M(x) for x in ff], W_final=[M(x) for x in Wf])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors_r03.json")
    with open(path, "w") as fh:
        json.dump(out, fh)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("this script parses the reference's constants: run it in the build container (/root/reference)")
    main()
