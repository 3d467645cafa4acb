#!/usr/bin/env python3
"""Generate poseidon16_consts.inc: Montgomery-form constant tables for the device/host Poseidon1-16.

The permutation is defined by the reference parameters (round constants in params/poseidon1_rc.inc, circulant
MDS column, 4+20+4 rounds, x^3; reference crates/backend/koala-bear/src/poseidon1_koalabear_16.rs:11-22,699-815).
The device evaluates the 20 partial rounds in the standard sparse form (Poseidon paper, App. B; the reference
does the same at :399-480,:873-912).  This script derives that form from first principles:

  original partial round r :  x += c_r ; x0 = x0^3 ; x = M x
  (1) constants: lanes 1..15 never see an S-box inside the partial block, so c_r is pulled backwards through
      M^{-1}; only a scalar on lane 0 remains after each S-box, plus one full vector before the first S-box.
  (2) matrices: M = P * Q with Q = diag(1, Mhat) (commutes with the lane-0 S-box and with lane-0 constants) and
      P = [[m00, (Mhat^{-T} v)^T], [w, I]] sparse.  Q is merged into the previous round's M, repeatedly;
      the last Q is the dense matrix D applied once before the first partial S-box.
  D is further fused with the MDS of the 4th initial full round:  x -> D (M x + first_rc) = (D M) x + D first_rc.

The script self-checks the derived tables against the textbook schedule on random states and on the reference
known-answer vector (:1083-1091) before writing anything.
"""
import os, re, random, sys

P = 0x7F000001
R = 1 << 32
HERE = os.path.dirname(os.path.abspath(__file__))
MDS_COL = [1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1]
RF_HALF, RP, W = 4, 20, 16


def load_rc():
    txt = open(os.path.join(HERE, "params", "poseidon1_rc.inc")).read()
    txt = txt[txt.index("*/") + 2:]
    vals = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", txt)]
    assert len(vals) == 28 * 16
    return [vals[16 * r:16 * r + 16] for r in range(28)]


def inv(a):
    return pow(a, P - 2, P)


def mat_mul(A, B):
    n, m, k = len(A), len(B[0]), len(B)
    return [[sum(A[i][t] * B[t][j] for t in range(k)) % P for j in range(m)] for i in range(n)]


def mat_vec(A, x):
    return [sum(a * b for a, b in zip(row, x)) % P for row in A]


def mat_inv(A):
    n = len(A)
    a = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c] % P)
        a[c], a[piv] = a[piv], a[c]
        iv = inv(a[c][c])
        a[c] = [x * iv % P for x in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % P for x, y in zip(a[r], a[c])]
    return [row[n:] for row in a]


def transpose(A):
    return [list(r) for r in zip(*A)]


def textbook(state, rc, M):
    s = state[:]
    r = 0
    for _ in range(RF_HALF):
        s = [pow((x + c) % P, 3, P) for x, c in zip(s, rc[r])]
        s = mat_vec(M, s)
        r += 1
    for _ in range(RP):
        s = [(x + c) % P for x, c in zip(s, rc[r])]
        s[0] = pow(s[0], 3, P)
        s = mat_vec(M, s)
        r += 1
    for _ in range(RF_HALF):
        s = [pow((x + c) % P, 3, P) for x, c in zip(s, rc[r])]
        s = mat_vec(M, s)
        r += 1
    return s


def derive(rc, M):
    prc = rc[RF_HALF:RF_HALF + RP]
    Minv = mat_inv(M)
    # (1) constants
    scalar = [0] * RP  # scalar[r] is added to lane 0 right after the S-box of partial round r (r < RP-1)
    tmp = prc[RP - 1][:]
    for i in range(RP - 2, -1, -1):
        back = mat_vec(Minv, tmp)
        scalar[i] = back[0]
        tmp = prc[i][:]
        for j in range(1, W):
            tmp[j] = (tmp[j] + back[j]) % P
    first_rc = tmp
    # (2) matrices
    rows, cols = [None] * RP, [None] * RP
    Mcur = [r[:] for r in M]
    Q = None
    for r in range(RP - 1, -1, -1):
        m00 = Mcur[0][0]
        v = Mcur[0][1:]
        w = [Mcur[i][0] for i in range(1, W)]
        Mhat = [row[1:] for row in Mcur[1:]]
        MhatInvT = transpose(mat_inv(Mhat))
        pv = mat_vec(MhatInvT, v)
        rows[r] = [m00] + pv          # s0' = rows . s
        cols[r] = w                   # s[i] += cols[i-1] * s0_old
        Q = [[1] + [0] * (W - 1)] + [[0] + Mhat[i] for i in range(W - 1)]
        Mcur = mat_mul(Q, M)
    D = Q
    DM = mat_mul(D, M)
    Dbias = mat_vec(D, first_rc)
    return dict(scalar=scalar, rows=rows, cols=cols, DM=DM, Dbias=Dbias, D=D)


def sparse_eval(state, rc, M, T):
    s = state[:]
    for r in range(RF_HALF - 1):
        s = [pow((x + c) % P, 3, P) for x, c in zip(s, rc[r])]
        s = mat_vec(M, s)
    s = [pow((x + c) % P, 3, P) for x, c in zip(s, rc[RF_HALF - 1])]
    s = [(a + b) % P for a, b in zip(mat_vec(T["DM"], s), T["Dbias"])]
    for r in range(RP):
        s0 = pow(s[0], 3, P)
        if r < RP - 1:
            s0 = (s0 + T["scalar"][r]) % P
        new0 = (T["rows"][r][0] * s0 + sum(a * b for a, b in zip(T["rows"][r][1:], s[1:]))) % P
        s = [new0] + [(s[i] + T["cols"][r][i - 1] * s0) % P for i in range(1, W)]
    for r in range(RF_HALF + RP, 2 * RF_HALF + RP):
        s = [pow((x + c) % P, 3, P) for x, c in zip(s, rc[r])]
        s = mat_vec(M, s)
    return s


def linearise(T):
    """Affine forms of the partial block over u = (t_0..t_15, q_0..q_19, 1): t = state entering the block (the AIR's
    beginning_full_rounds[1] columns), q_r = committed post-S-box lane-0 value of partial round r (partial_rounds[r]).
    Lanes 1..15 never see an S-box inside the block and lane 0 is re-based on q_r each round, so the value cubed in
    round r (y_r) and the state leaving the block (fin) are affine in u.  Returns (Y[20][37], F[16][37])."""
    NU = W + RP + 1

    def unit(j):
        v = [0] * NU
        v[j] = 1
        return v

    def axpy(a, x, y):  # a*x + y
        return [(a * xi + yi) % P for xi, yi in zip(x, y)]

    s = []
    for i in range(W):
        v = [0] * NU
        for j in range(W):
            v[j] = T["D"][i][j]
        v[NU - 1] = T["Dbias"][i]
        s.append(v)
    Y = []
    for r in range(RP):
        Y.append(s[0][:])
        s0 = unit(W + r)
        if r < RP - 1:
            s0[NU - 1] = T["scalar"][r]
        n0 = [T["rows"][r][0] * x % P for x in s0]
        for j in range(1, W):
            n0 = axpy(T["rows"][r][j], s[j], n0)
        for i in range(1, W):
            s[i] = axpy(T["cols"][r][i - 1], s0, s[i])
        s[0] = n0
    for r in range(RP):
        assert all(Y[r][W + k] == 0 for k in range(r, RP)), "y_r must only depend on q_0..q_{r-1}"
    return Y, s


def linear_eval(state, rc, M, T, Y, F):
    """Full permutation through the affine forms: q_r = y_r^3 (honest trace), cross-check against the textbook."""
    s = state[:]
    for r in range(RF_HALF):
        s = [pow((x + c) % P, 3, P) for x, c in zip(s, rc[r])]
        s = mat_vec(M, s)
    u = s + [0] * RP + [1]
    for r in range(RP):
        y = sum(a * b for a, b in zip(Y[r], u)) % P
        u[W + r] = pow(y, 3, P)
    s = [sum(a * b for a, b in zip(row, u)) % P for row in F]
    for r in range(RF_HALF + RP, 2 * RF_HALF + RP):
        s = [pow((x + c) % P, 3, P) for x, c in zip(s, rc[r])]
        s = mat_vec(M, s)
    return s


def monty(x):
    return (x % P) * R % P


def fmt(vals):
    return "{ " + ", ".join("0x%08xu" % monty(v) for v in vals) + " }"


def main():
    rc = load_rc()
    M = [[MDS_COL[(16 + i - j) % 16] for j in range(16)] for i in range(16)]
    T = derive(rc, M)
    kat_in = list(range(16))
    kat_out = [610090613, 935319874, 1893335292, 796792199, 356405232, 552237741, 55134556, 1215104204,
               1823723405, 1133298033, 1780633798, 1453946561, 710069176, 1128629550, 1917333254, 1175481618]
    assert textbook(kat_in, rc, M) == kat_out, "textbook schedule does not reproduce the reference KAT"
    assert sparse_eval(kat_in, rc, M, T) == kat_out, "sparse form does not reproduce the reference KAT"
    rng = random.Random(1)
    for _ in range(20):
        st = [rng.randrange(P) for _ in range(16)]
        assert textbook(st, rc, M) == sparse_eval(st, rc, M, T)
    Y, F = linearise(T)
    assert linear_eval(kat_in, rc, M, T, Y, F) == kat_out, "linearised partial block does not reproduce the KAT"
    for _ in range(5):
        st = [rng.randrange(P) for _ in range(16)]
        assert textbook(st, rc, M) == linear_eval(st, rc, M, T, Y, F)
    lin = ["// GENERATED by gen_poseidon_consts.py — do not edit.  Montgomery form.  Affine forms of the 20 partial rounds over",
           "// u = (t_0..t_15, q_0..q_19, 1); layout must match struct PoseidonLinear in air_tables.h", "{",
           "  /* y[20][37] */ { " + ",\n    ".join(fmt(row) for row in Y) + " },",
           "  /* fin[16][37] */ { " + ",\n    ".join(fmt(row) for row in F) + " },", "}"]
    open(os.path.join(HERE, "poseidon16_linear.inc"), "w").write("\n".join(lin) + "\n")
    # hashing variant: t = M c with c = the S-box outputs of the 4th full round, so the MDS of that round is folded in
    def fuse(rows):
        return [[sum(row[i] * M[i][j] for i in range(W)) % P for j in range(W)] + row[W:] for row in rows]
    YM, FM = fuse(Y), fuse(F)
    for _ in range(5):
        st = [rng.randrange(P) for _ in range(16)]
        s4 = st[:]
        for r in range(RF_HALF - 1):
            s4 = mat_vec(M, [pow((x + c) % P, 3, P) for x, c in zip(s4, rc[r])])
        u = [pow((x + c) % P, 3, P) for x, c in zip(s4, rc[RF_HALF - 1])] + [0] * RP + [1]
        for r in range(RP):
            u[W + r] = pow(sum(a * b for a, b in zip(YM[r], u)) % P, 3, P)
        s4 = [sum(a * b for a, b in zip(row, u)) % P for row in FM]
        for r in range(RF_HALF + RP, 2 * RF_HALF + RP):
            s4 = mat_vec(M, [pow((x + c) % P, 3, P) for x, c in zip(s4, rc[r])])
        assert s4 == textbook(st, rc, M), "fused linearised partial block is wrong"
    linh = ["// GENERATED by gen_poseidon_consts.py — do not edit.  Montgomery form.  Partial block as affine forms over",
            "// u = (c_0..c_15, q_0..q_19, 1), c = S-box outputs of the 4th full round (its MDS is folded in).", "{",
            "  /* y[20][37] */ { " + ",\n    ".join(fmt(row) for row in YM) + " },",
            "  /* fin[16][37] */ { " + ",\n    ".join(fmt(row) for row in FM) + " },", "}"]
    open(os.path.join(HERE, "poseidon16_linear_hash.inc"), "w").write("\n".join(linh) + "\n")
    out = []
    out.append("// GENERATED by gen_poseidon_consts.py — do not edit.  All values Montgomery form (R = 2^32).")
    out.append("// layout must match struct PoseidonConsts in poseidon16.h")
    out.append("{")
    out.append("  /* rc_init[4][16] */ { " + ",\n    ".join(fmt(rc[r]) for r in range(4)) + " },")
    out.append("  /* rc_term[4][16] */ { " + ",\n    ".join(fmt(rc[r]) for r in range(24, 28)) + " },")
    out.append("  /* dm[16][16] = D*MDS */ { " + ",\n    ".join(fmt(row) for row in T["DM"]) + " },")
    out.append("  /* dbias[16] */ " + fmt(T["Dbias"]) + ",")
    out.append("  /* dmat[16][16] = D (entry map of the partial block, used by the Poseidon AIR) */ { " +
               ",\n    ".join(fmt(row) for row in T["D"]) + " },")
    out.append("  /* prow[20][16] */ { " + ",\n    ".join(fmt(row) for row in T["rows"]) + " },")
    out.append("  /* pcol[20][16] (index 15 unused) */ { " + ",\n    ".join(fmt(c + [0]) for c in T["cols"]) + " },")
    out.append("  /* pscalar[20] (index 19 unused) */ " + fmt(T["scalar"]) + ",")
    out.append("}")
    path = os.path.join(HERE, "poseidon16_consts.inc")
    open(path, "w").write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
