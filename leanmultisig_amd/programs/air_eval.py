"""evaluate_air_constraints of the recursion program (recursion.py:777-787), emitted at the ISA level: the constraint polynomial of each
leanVM table, sum_k alpha^k C_k, at the column evaluations the prover sends behind the batched AIR sumcheck.

The reference GENERATES these three functions from its symbolic constraint systems (crates/lean_compiler); here they are written out
against the assembler (programs/asm.py), constraint by constraint in the reference's order:
    execution      crates/lean_vm/src/tables/execution/air.rs:56-129         13 constraints
    extension_op   crates/lean_vm/src/tables/extension_op/air.rs:59-163      34 constraints (two quintic products)
    poseidon_16    crates/lean_vm/src/tables/poseidon_16/mod.rs:316-548     100 constraints (eval_poseidon1_16: 8 full + 20 partial rounds)
    bus column     crates/lean_vm/src/tables/utils.rs:5-21
Lowering: a table's constraint values are stored side by side and weighted with ONE dot product against the powers of alpha; the MDS
layer is 16 base-by-extension dot products against a sliding window over the circulant's column (31 words written once per program);
a product by a small constant is a one-term base-by-extension dot product.  The values must equal what the library's verifier
computes (csrc/air_tables.h through lmh_verify_execution_raw: lm_pcs_statement_claim::air_constraint_evals) — tests/test_whir_verify_program.py.
"""
from ..vm import FP, K, M
from .asm import DIM, absolute, fp

P = 0x7F000001
MDS_COL = [1, 3, 13, 22, 67, 2, 15, 63, 101, 1, 2, 17, 11, 1, 51, 1]   # poseidon1_koalabear_16.rs:22
N_CONSTRAINTS = {0: 13, 1: 34, 2: 100}
MAX_ALPHA = 101


def load_round_constants():
    """POSEIDON1_RC (poseidon1_koalabear_16.rs:699-815) from the parameter file the device tables are generated from: 28 x 16 canonical"""
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "params", "poseidon1_rc.inc")
    txt = open(path).read()
    txt = txt[txt.index("*/") + 2:]
    vals = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", txt)]
    assert len(vals) == 28 * 16
    return [vals[16 * r:16 * r + 16] for r in range(28)]


class Alg:
    """extension-field expressions over a function body `f` (asm.Fn); every value is a location of 5 words"""

    def __init__(self, f, one, zero):
        self.f, self.one, self.zero = f, one, zero

    def add(self, a, b):
        return self.f.add(a, b)

    def sub(self, a, b):
        return self.f.sub(a, b)

    def mul(self, a, b):
        return self.f.mul(a, b)

    def neg(self, a):
        return self.f.sub(self.zero, a)

    def mulc(self, a, c):                       # a * (base constant)
        out = self.f.new_ef()
        self.f.dot(fp(self.f.const(c)), a, out, 1, be=True)
        return out

    def addc(self, a, c):                       # a + (base constant)
        out = self.f.new_ef()
        self.f.ext("add", fp(self.f.const(c)), a, out, be=True)
        return out

    def cube(self, a):
        return self.mul(self.mul(a, a), a)

    def store(self, value, dst):
        self.f.copy5(value, dst)


class Folder:
    """ConstraintFolder: the constraint values side by side, weighted by the powers of alpha in one dot product"""

    def __init__(self, A, n):
        self.A, self.n, self.k = A, n, 0
        self.values = A.f.new_ef(n)

    def assert_zero(self, value):
        self.A.store(value, self.values + DIM * self.k)
        self.k += 1

    def slot(self):
        """the location of the next constraint value, for an operation that can write it directly"""
        loc = self.values + DIM * self.k
        self.k += 1
        return loc

    def result(self, alpha_powers):
        assert self.k == self.n, (self.k, self.n)
        out = self.A.f.new_ef()
        self.A.f.dot(self.values, alpha_powers, out, self.n)
        return out


def bus_column(A, X, flag, data):
    """(sum_{i<4} eq[i] data[i] + eq[15] * LOGUP_PRECOMPILE_DOMAINSEP) * beta + flag   (tables/utils.rs:5-21; the separator is 1)"""
    f = A.f
    buf = f.new_ef(4)
    for i, d in enumerate(data):
        f.copy5(d, buf + DIM * i)
    s = f.new_ef()
    f.dot(buf, X["aeq"], s, 4)
    return A.add(A.mul(A.add(s, X["aeq"] + DIM * 15), X["bus_beta"]), flag)


def bool_check(A, v):
    return A.mul(A.sub(A.one, v), v)


def eval_execution(A, flat, shift, X):
    c = lambda i: flat + DIM * i  # noqa: E731
    pc, fp_, addr_a, addr_b, addr_c = c(0), c(1), c(2), c(3), c(4)
    value_a, value_b, value_c = c(5), c(6), c(7)
    operand_a, operand_b, operand_c = c(8), c(9), c(10)
    flag_a, flag_b, flag_c, flag_c_fp, flag_ab_fp = c(11), c(12), c(13), c(14), c(15)
    mul, jump, aux, precompile_data = c(16), c(17), c(18), c(19)
    pc_shift, fp_shift = shift, shift + DIM
    one = A.one
    omfa = A.sub(one, A.add(flag_a, flag_ab_fp))
    omfb = A.sub(one, A.add(flag_b, flag_ab_fp))
    omfc = A.sub(one, A.add(flag_c, flag_c_fp))
    fpa, fpb, fpc = A.add(fp_, operand_a), A.add(fp_, operand_b), A.add(fp_, operand_c)
    nu_a = A.add(A.add(A.mul(flag_a, operand_a), A.mul(omfa, value_a)), A.mul(flag_ab_fp, fpa))
    nu_b = A.add(A.add(A.mul(flag_b, operand_b), A.mul(omfb, value_b)), A.mul(flag_ab_fp, fpb))
    nu_c = A.add(A.add(A.mul(flag_c, operand_c), A.mul(omfc, value_c)), A.mul(flag_c_fp, fpc))
    add_ = A.sub(A.add(aux, aux), A.mul(aux, aux))
    deref = A.mulc(A.mul(aux, A.sub(aux, one)), (P + 1) // 2)
    is_precompile = A.sub(one, A.add(A.add(A.add(add_, mul), deref), jump))
    F = Folder(A, N_CONSTRAINTS[0])
    F.assert_zero(bus_column(A, X, is_precompile, [precompile_data, nu_a, nu_b, nu_c]))
    F.assert_zero(A.mul(omfa, A.sub(addr_a, fpa)))
    F.assert_zero(A.mul(omfb, A.sub(addr_b, fpb)))
    F.assert_zero(A.mul(omfc, A.sub(addr_c, fpc)))
    F.assert_zero(A.mul(add_, A.sub(nu_b, A.add(nu_a, nu_c))))
    F.assert_zero(A.mul(mul, A.sub(nu_b, A.mul(nu_a, nu_c))))
    F.assert_zero(A.mul(deref, A.sub(addr_b, A.add(value_a, operand_b))))
    F.assert_zero(A.mul(deref, A.sub(value_b, nu_c)))
    jc = A.mul(jump, nu_a)
    F.assert_zero(A.mul(jc, A.sub(nu_a, one)))
    F.assert_zero(A.mul(jc, A.sub(pc_shift, nu_b)))
    F.assert_zero(A.mul(jc, A.sub(fp_shift, nu_c)))
    njc = A.sub(one, jc)
    F.assert_zero(A.mul(njc, A.sub(pc_shift, A.add(pc, one))))
    F.assert_zero(A.mul(njc, A.sub(fp_shift, fp_)))
    return F.result(X["apw"])


def quintic_mul(A, a, b):
    """quintic_mul with plain dot products (extension_op/air.rs:37-42): a, b lists of 5 values -> list of 5"""
    f = A.f
    b0m3, b1m4, b4m2 = A.sub(b[0], b[3]), A.sub(b[1], b[4]), A.sub(b[4], b[2])
    b3m14 = A.sub(b[3], b1m4)
    rows = [[b[0], b[4], b[3], b[2], b1m4], [b[1], b[0], b[4], b[3], b[2]], [b[2], b1m4, b0m3, b4m2, b3m14],
            [b[3], b[2], b1m4, b0m3, b4m2], [b[4], b[3], b[2], b1m4, b0m3]]
    av = f.new_ef(5)
    for i in range(5):
        f.copy5(a[i], av + DIM * i)
    out = []
    for r in rows:
        rv = f.new_ef(5)
        for i in range(5):
            f.copy5(r[i], rv + DIM * i)
        o = f.new_ef()
        f.dot(av, rv, o, 5)
        out.append(o)
    return out


def eval_extension_op(A, flat, shift, X):
    c = lambda i: flat + DIM * i   # noqa: E731
    sh = lambda i: shift + DIM * i  # noqa: E731
    one = A.one
    is_be, start, ln, flag_add, flag_mul, flag_poly_eq = c(0), c(1), c(2), c(3), c(4), c(5)
    idx_a, idx_b, idx_r = c(6), c(7), c(13)
    comp, va, vb, vres = [c(8 + k) for k in range(5)], [c(14 + k) for k in range(5)], [c(19 + k) for k in range(5)], [c(24 + k) for k in range(5)]
    comp_shift = [sh(8 + k) for k in range(5)]
    start_shift = sh(1)
    activation_flag = A.mul(start, A.add(A.add(flag_add, flag_mul), flag_poly_eq))
    aux = A.add(A.add(A.add(A.add(A.mulc(is_be, 4), A.mulc(flag_add, 8)), A.mulc(flag_mul, 16)), A.mulc(flag_poly_eq, 32)), A.mulc(ln, 64))
    F = Folder(A, N_CONSTRAINTS[1])
    F.assert_zero(bus_column(A, X, activation_flag, [aux, idx_a, idx_b, idx_r]))
    is_ee = A.sub(one, is_be)
    nss = A.sub(one, start_shift)
    vaf = [va[0]] + [A.mul(va[k], is_ee) for k in range(1, 5)]
    comp_tail = [A.mul(comp_shift[k], nss) for k in range(5)]
    for v in (is_be, start, flag_add, flag_mul, flag_poly_eq):
        F.assert_zero(bool_check(A, v))
    for k in range(5):
        F.assert_zero(A.mul(A.sub(comp[k], A.add(A.add(vaf[k], vb[k]), comp_tail[k])), flag_add))
    vavb = quintic_mul(A, vaf, vb)
    for k in range(5):
        F.assert_zero(A.mul(A.sub(comp[k], A.add(vavb[k], comp_tail[k])), flag_mul))
    pev, csoo = [], []
    for k in range(5):
        base = A.sub(A.sub(A.add(vavb[k], vavb[k]), vaf[k]), vb[k])
        pev.append(A.add(base, one) if k == 0 else base)
        csoo.append(A.add(comp_tail[0], start_shift) if k == 0 else comp_tail[k])
    per = quintic_mul(A, pev, csoo)
    for k in range(5):
        F.assert_zero(A.mul(A.sub(comp[k], per[k]), flag_poly_eq))
    for k in range(5):
        F.assert_zero(A.mul(A.sub(comp[k], vres[k]), start))
    F.assert_zero(A.mul(nss, A.sub(A.sub(ln, sh(2)), one)))
    F.assert_zero(A.mul(nss, A.sub(is_be, sh(0))))
    F.assert_zero(A.mul(nss, A.sub(flag_add, sh(3))))
    F.assert_zero(A.mul(nss, A.sub(flag_mul, sh(4))))
    F.assert_zero(A.mul(nss, A.sub(flag_poly_eq, sh(5))))
    a_inc = A.add(is_be, A.mulc(is_ee, 5))
    F.assert_zero(A.mul(nss, A.sub(A.sub(sh(6), idx_a), a_inc)))
    F.assert_zero(A.mul(nss, A.addc(A.sub(sh(7), idx_b), P - 5)))
    F.assert_zero(A.mul(start_shift, A.sub(ln, one)))
    return F.result(X["apw"])


def eval_poseidon16(A, col, X, mds_window):
    """eval_poseidon1_16 on the committed state columns (poseidon_16/mod.rs:316-548).  mds_window: location of 31 base words
    W[m] = col[(15 - m) mod 16], so that row i of the circulant is the 16 words from W + 15 - i."""
    f, p = A.f, A.f.p
    RC = load_round_constants()
    c = lambda i: col + DIM * i  # noqa: E731
    one = A.one
    flag_active, index_b, index_res, flag_half, flag_left = c(0), c(1), c(2), c(3), c(4)
    offset_left, eff_first, eff_second, flag_permute = c(5), c(6), c(7), c(8)
    inputs, bfr, partial, efr, out_left, out_right = 9, 25, 57, 77, 93, 101
    pdr = A.add(A.add(A.add(A.addc(A.mulc(flag_half, 4), 1), A.mulc(flag_left, 8)), A.mulc(A.mul(flag_left, offset_left), 16)), A.mulc(flag_permute, 2))
    omfl = A.sub(one, flag_left)
    index_a = A.sub(eff_second, A.mulc(omfl, 4))
    F = Folder(A, N_CONSTRAINTS[2])
    F.assert_zero(bus_column(A, X, flag_active, [pdr, index_a, index_b, index_res]))
    for v in (flag_active, flag_half, flag_left, flag_permute):
        F.assert_zero(bool_check(A, v))
    F.assert_zero(A.mul(flag_permute, A.add(flag_half, flag_left)))
    F.assert_zero(A.mul(flag_left, A.sub(offset_left, eff_first)))
    F.assert_zero(A.mul(omfl, A.sub(index_a, eff_first)))

    def mds(state):
        """state: 16 contiguous values -> 16 contiguous values"""
        out = f.new_ef(16)
        for i in range(16):
            f.dot(mds_window + (15 - i), state, out + DIM * i, 16, be=True)
        return out

    def full_round(state, r):
        cubes = f.new_ef(16)
        for i in range(16):
            t = A.addc(state + DIM * i, RC[r][i])
            f.mul(f.mul(t, t), t, cubes + DIM * i)
        return mds(cubes)

    s = c(inputs)
    r = 0
    for blk in range(2):
        s = full_round(full_round(s, r), r + 1)
        r += 2
        for i in range(16):
            f.sub(s + DIM * i, c(bfr + 16 * blk + i), F.slot())
        s = c(bfr + 16 * blk)
    for pr in range(20):
        u = f.new_ef(16)
        t0 = A.addc(s, RC[r][0])
        for i in range(1, 16):
            f.ext("add", fp(f.const(RC[r][i])), s + DIM * i, u + DIM * i, be=True)
        f.sub(A.cube(t0), c(partial + pr), F.slot())
        f.copy5(c(partial + pr), u)
        s = mds(u)
        r += 1
    s = full_round(full_round(s, r), r + 1)
    r += 2
    for i in range(16):
        f.sub(s + DIM * i, c(efr + i), F.slot())
    s = full_round(full_round(c(efr), r), r + 1)
    assert r + 2 == 28
    not_permute = A.sub(one, flag_permute)
    comp_last4 = A.sub(not_permute, flag_half)
    for i in range(8):
        gate = not_permute if i < 4 else comp_last4
        F.assert_zero(A.mul(gate, A.sub(A.add(s + DIM * i, c(inputs + i)), c(out_left + i))))
        F.assert_zero(A.mul(flag_permute, A.sub(s + DIM * i, c(out_left + i))))
        F.assert_zero(A.mul(flag_permute, A.sub(s + DIM * (i + 8), c(out_right + i))))
    return F.result(X["apw"])


def write_mds_window(f):
    """-> location of the 31 words W[m] = col[(15 - m) mod 16] in the function's own frame"""
    w = f.alloc(31)
    for m in range(31):
        f.p.add(K(0), K(MDS_COL[(15 - m) % 16]), M(w + m))
    return fp(w)
