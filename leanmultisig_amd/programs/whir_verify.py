"""The PCS-opening part of the recursion program — `whir_open` of the reference's in-VM verifier — assembled by hand at the ISA
level, and the witness of one run on genuine proofs.

Reference: crates/rec_aggregation/zkdsl_implem/whir.py (whir_open :18-164, whir_round :316-368, sample_stir_indexes_and_fold
:266-313, sumcheck_verify* :167-220, parse_commitment :377-391), fiat_shamir.py (the sponge over the raw transcript: _absorb_chunks,
fs_grinding + assert_trailing_bits_are_zeros, fs_duplex, fs_sample_chunks, fs_sample_queries, fs_receive_*), hashing.py
(slice_hash_rtl :54-60, whir_do_{4,3,2,1}_merkle_levels :114-203), utils.py (decompose_and_verify_merkle_query :537-626, powers_const,
compute_eq_mle_extension, expand_from_univariate_*, univariate_eval_on_base, eval_multilinear_coeffs_rev, univariate_polynomial_eval);
the call site is recursion.py:470-532 and its results are consumed at :534-654.  The zkDSL compiler is out of scope:
the statements are lowered by hand with leanmultisig_amd/programs/asm.py, for ONE WhirConfig (the reference's program dispatches over
every admissible configuration with match_range; here the configuration of the child proofs is an assembly-time constant).

What the program proves.  Public input = slice_hash_with_iv of a claims buffer holding, per child proof, the arguments and the
expected results of whir_open as lm_whir_opening_claim (include/leanmultisig_host.h) has them: the sponge state at the opening, the
commitment root, the OOD points and answers of the commitment, the statement's share of the initial sum and of the final weight, the
folding randomness.  For every child the program replays the Fiat-Shamir transcript of WhirConfig::verify (whir/src/verify.rs:83-232)
over the raw proof transcript, checks every sumcheck polynomial, grinding witness, Merkle opening (leaf sponge + path), folded leaf,
the final polynomial at the final queries, recomputes the constraint weights of the OOD and STIR points at the folding randomness, and
asserts (s + statement_weights) * final_value == end_sum and folding randomness == claim.  What stays outside is everything
recursion.py does BEFORE the opening (GKR, logup, AIR, statement assembly): the claims buffer stands for it.

MI355X-first lowering: the reference's loops over the queries of a round (decompose_and_verify_merkle_batch_const, the fold loops,
the s6s loop of whir_open) are `range` loops run once per child; here the children advance in lock step and each such loop is ONE
parallel loop over (child, query) pairs — 4 x 118 / 56 / 28 segments per batch at the reference's recursion configuration — which
leanVM's runner hands to the device (csrc/lm_vm_device.hip, one wavefront per segment).  A segment finds its child's pointers
through a jump table indexed by the iteration (match_range), verifies its opening and folds its leaf in the same iteration.
"""
import numpy as np

from ..vm import FP, K, M, Label, Program, Witness, to_monty
from .asm import DIM, DIGEST_LEN, Fn, absolute, at, fp, loop_epilogue, loop_prologue

P = 0x7F000001
TWO_ADICITY = 24
ROOT_24 = 0x6AC49F88           # generator of the 2^24-th roots of unity (koala_bear.rs:50-54), canonical
PUBLIC_INPUT_LEN = DIGEST_LEN
ZERO_VEC_PTR, ZERO_VEC_LEN = PUBLIC_INPUT_LEN, 16
SDS_PTR = ZERO_VEC_PTR + ZERO_VEC_LEN
ONE_EF_PTR = SDS_PTR + DIGEST_LEN
REPEATED_ONES_PTR, NUM_REPEATED_ONES = ONE_EF_PTR + DIM, 32
PREAMBLE_MEMORY_LEN = ZERO_VEC_LEN + DIGEST_LEN + DIM + NUM_REPEATED_ONES
MAIN_FP = (PUBLIC_INPUT_LEN + PREAMBLE_MEMORY_LEN + 4) // 5 * 5    # the runner's starting fp (runner.rs:253-255)


def ceil_div(a, b):
    return -(-a // b)


TABLE_COLUMNS = {0: (20, 2), 1: (29, 13), 2: (109, 0)}   # table id -> (n_columns, n_shift): execution, extension_op, poseidon16 (tables/*/air.rs)


class Statement:
    """The static layout of the PCS statement of one child shape (capi.PcsStatementClaim + WhirOpeningClaim): table heights and order,
    where the claimed evaluations lie in the raw transcript.  Two children with equal layouts run the same program."""

    def __init__(self, stmt, claim, public_input_len=None):
        self.public_input_len = public_input_len if public_input_len is not None else 1 << int(stmt.log_public_memory)
        self.log_rows = [int(x) for x in stmt.log_rows]
        self.order = [int(x) for x in stmt.table_order]
        self.log_memory, self.log_bytecode = int(stmt.log_memory), int(stmt.log_bytecode)
        self.gkr_n_vars, self.n_max, self.ending_pc, self.lpm = int(stmt.gkr_n_vars), int(stmt.n_max), int(stmt.ending_pc), int(stmt.log_public_memory)
        self.off_whir = int(claim.transcript_offset)
        self.off_value_memory_acc, self.off_value_memory = int(stmt.off_value_memory_acc), int(stmt.off_value_memory)
        self.off_value_bytecode_acc = int(stmt.off_value_bytecode_acc)
        self.off_inner = [int(x) for x in stmt.off_inner_evals]
        self.logup = {t: sorted((int(stmt.logup_col[t][k]), int(stmt.logup_off[t][k])) for k in range(int(stmt.n_logup_values[t]))) for t in range(3)}
        self.n_values = 6 + sum(len(self.logup[t]) + sum(TABLE_COLUMNS[t]) for t in range(3))
        self.rate, self.n_ood = int(claim.log_inv_rate), int(claim.n_ood)
        # the batched AIR sumcheck in front of the statement
        self.air_off, self.air_degree = int(stmt.air_offset), int(stmt.air_degree)
        self.off_bus_selector, self.off_bus_data = [int(x) for x in stmt.off_bus_selector], [int(x) for x in stmt.off_bus_data]

    def key(self):
        return (self.log_rows, self.order, self.log_memory, self.log_bytecode, self.gkr_n_vars, self.n_max, self.ending_pc, self.off_whir,
                self.off_value_memory_acc, self.off_value_memory, self.off_value_bytecode_acc, self.off_inner, sorted(self.logup.items()),
                self.air_off, self.air_degree, self.off_bus_selector, self.off_bus_data, self.lpm)


class Shape:
    """everything the assembly depends on, derived from a WhirConfig dict (capi.WhirConfig.to_dict) and, optionally, a Statement"""

    def __init__(self, cfg, n_children, statement=None, air=False, head=False):
        assert (statement is not None or not air) and (air or not head)
        self.cfg, self.n_children, self.statement, self.air, self.head = cfg, n_children, statement, air, head
        self.n, self.rate, self.n_rounds = cfg["num_variables"], cfg["starting_log_inv_rate"], cfg["n_rounds"]
        assert self.n_rounds >= 1, "a configuration without a WHIR round has base-field leaves in the final round (not assembled)"
        self.fold = [cfg["fold_first"]] + [cfg["fold_sub"]] * self.n_rounds
        self.n_final = cfg["final_sumcheck_rounds"]
        assert self.n == sum(self.fold) + self.n_final
        self.queries = [r["num_queries"] for r in cfg["rounds"]] + [cfg["final_queries"]]
        self.oods = [cfg["commitment_ood_samples"]] + [r["ood_samples"] for r in cfg["rounds"]]
        self.query_grinding = [r["query_pow_bits"] for r in cfg["rounds"]] + [cfg["final_query_pow_bits"]]
        self.folding_grinding = [cfg["starting_folding_pow_bits"]] + [r["folding_pow_bits"] for r in cfg["rounds"]]
        dom = self.n + self.rate
        self.height, self.leaf_words = [], []
        for r in range(self.n_rounds + 1):
            self.height.append(dom - self.fold[r])
            self.leaf_words.append((1 << self.fold[r]) * (1 if r == 0 else DIM))
            dom -= cfg["rs_red"] if r == 0 else 1
        assert all(5 <= h <= TWO_ADICITY for h in self.height) and all(1 <= o <= 4 for o in self.oods)
        self.n_rem = [self.n - sum(self.fold[:r + 1]) for r in range(self.n_rounds + 1)]   # variables left after round r's fold
        # claims buffer of one child (words); mirrors lm_whir_opening_claim
        o0 = self.oods[0]
        self.c_fs, self.c_root = 0, 16
        self.c_ood_points = 24
        self.c_ood_evals = self.c_ood_points + DIM * o0
        if statement is None:   # the statement's share of the initial sum and of the final weight are claims
            self.c_stmt_sum = self.c_ood_evals + DIM * o0
            self.c_stmt_weights = self.c_stmt_sum + DIM
            self.c_rand = self.c_stmt_weights + DIM
        elif not air:           # the statement itself: the child's public input and the three points; the values are in the transcript
            self.c_public_input = self.c_ood_evals + DIM * o0
            self.c_gkr_point = self.c_public_input + ceil_div(1 << statement.lpm, DIGEST_LEN) * DIGEST_LEN
            self.c_air_point = self.c_gkr_point + DIM * statement.gkr_n_vars     # all_challenges: the LAST challenge first (recursion.py:416)
            self.c_pm_point = self.c_air_point + DIM * statement.n_max
            self.c_rand = self.c_pm_point + DIM * statement.lpm
        elif head:              # the whole of recursion.py but evaluate_air_constraints: the transcript is replayed from its first word;
            self.c_public_input = 0                                                    # claims: the child's public input, the digest of
            self.c_domsep = ceil_div(1 << statement.lpm, DIGEST_LEN) * DIGEST_LEN      # (bytecode hash, domain separator), the three AIR
            self.c_air_evals = self.c_domsep + DIGEST_LEN                              # constraint evaluations and the bytecode value
            self.c_bytecode_value = self.c_air_evals + 3 * DIM                         # (a hint in the reference too, recursion.py:138)
            self.c_rand = self.c_bytecode_value + DIM
        else:                   # + the batched AIR sumcheck: c_fs is the sponge when bus_beta is sampled; the sumcheck's challenges and the
            self.c_public_input = self.c_ood_evals + DIM * o0      # public-memory point are SAMPLED by the program; what is left of the
            self.c_gkr_point = self.c_public_input + ceil_div(1 << statement.lpm, DIGEST_LEN) * DIGEST_LEN   # verifier's earlier work: the GKR
            self.c_logup_c = self.c_gkr_point + DIM * statement.gkr_n_vars                                   # point, logup_c, and the three
            self.c_air_evals = self.c_logup_c + DIM                                                          # AIR constraint evaluations
            self.c_rand = self.c_air_evals + 3 * DIM
        self.claim_words = ceil_div(self.c_rand + DIM * self.n, DIGEST_LEN) * DIGEST_LEN
        # raw transcript words whir_open reads (every absorbed slice padded to the rate)
        t = 0
        for r in range(self.n_rounds + 1):
            t += self.fold[r] * (16 + (8 if self.folding_grinding[r] else 0))
            if r < self.n_rounds:
                t += 8 + ceil_div(DIM * self.oods[r + 1], 8) * 8 + (8 if self.query_grinding[r] else 0)
        t += ceil_div(DIM << self.n_final, 8) * 8 + (8 if self.query_grinding[self.n_rounds] else 0) + 16 * self.n_final
        self.whir_transcript_words = t
        self.transcript_words = t + (statement.off_whir if statement else 0)   # with a statement the program is given the whole raw transcript
        if air:   # what the AIR part reads in front of the opening: round polynomials and the tables' column evaluations
            T = statement
            a = T.n_max * ceil_div(DIM * (T.air_degree + 1), 8) * 8 + sum(ceil_div(DIM * sum(TABLE_COLUMNS[t]), 8) * 8 for t in range(3))
            assert T.air_off + a == T.off_whir, (T.air_off, a, T.off_whir)


class Fs:
    """fiat_shamir.py: the sponge [capacity | rate] and the transcript cursor, tracked at assembly time"""

    def __init__(self, f, state, transcript_cell):
        self.f, self.st, self.tb, self.off = f, state, transcript_cell, 0

    def _absorb(self, data):
        new = fp(self.f.alloc(16))
        self.f.permute(self.st, data, new)   # poseidon16_permute(fs, data, chain) (_absorb_chunks :20-27)
        self.st = new

    def duplex(self):
        self._absorb(absolute(ZERO_VEC_PTR))

    def rate(self):
        return self.st + 8

    def receive_chunks(self, n):
        loc = at(self.tb, self.off)
        for i in range(n):
            self._absorb(loc + DIGEST_LEN * i)
        self.off += DIGEST_LEN * n
        return loc

    def receive_ef(self, n):
        """fs_receive_ef_inlined :168-172: the padding of the last chunk is checked to be zero"""
        nch = ceil_div(n * DIM, DIGEST_LEN)
        loc = self.receive_chunks(nch)
        for i in range(n * DIM, nch * DIGEST_LEN):
            self.f.assert_zero(loc + i)
        return loc

    def sample_chunks(self, n):
        """fs_sample_chunks :109-129: n x 8 squeezed words, contiguous"""
        if n == 1:
            return self.rate()
        f = self.f
        samples = fp(f.alloc(DIGEST_LEN * n))
        f.copy8(self.rate(), samples)
        for i in range(1, n):
            self.duplex()
            f.copy8(self.rate(), samples + DIGEST_LEN * i)
        return samples

    def grinding(self, bits):
        """fs_grinding :49-60"""
        if bits == 0:
            return
        self.receive_chunks(1)
        assert self.st.kind == "fp"
        assert_trailing_bits_are_zeros(self.f, self.st.a + 8, bits)


def assert_trailing_bits_are_zeros(f, value, bits):
    """fiat_shamir.py:63-96: `value` (a cell) has its `bits` low bits zero; canonical decomposition 12 + 12 + 7 bits"""
    p = f.p
    ch = f.alloc(2)
    p.hint_decompose_bits_merkle_whir(FP(ch), M(value), K(12))
    f.range_check(ch, 4095)
    f.range_check(ch + 1, 4095)
    t, ps, diff, top7 = f.alloc(), f.alloc(), f.alloc(), f.alloc()
    p.mul(M(ch + 1), K(4096), M(t))
    p.add(M(ch), M(t), M(ps))
    p.add(M(diff), M(value), M(ps))          # diff = partial_sum - value
    p.mul(M(diff), K(127), M(top7))          # inv(2^24) = -127
    f.range_check(top7, 127)
    assert_if_127_then_zero(f, top7, ps)
    if bits < 12:
        q = f.alloc()
        p.mul(M(q), K(1 << bits), M(ch))     # q = chunks[0] / 2^bits
        f.range_check(q, (1 << (12 - bits)) - 1)
    elif bits < 24:
        p.add(M(ch), K(0), K(0))
        q = f.alloc()
        p.mul(M(q), K(1 << (bits - 12)), M(ch + 1))
        f.range_check(q, (1 << (24 - bits)) - 1)
    else:
        p.add(M(ch), K(0), K(0))
        p.add(M(ch + 1), K(0), K(0))


def assert_if_127_then_zero(f, top7, low):
    """`if top7 == 2**7 - 1: assert low == 0` (the 31-bit decomposition is the canonical one)"""
    p = f.p
    t, tinv, nz, omnz = f.alloc(), f.alloc(), f.alloc(), f.alloc()
    skip = f.fresh("canon")
    p.add(M(t), K(127), M(top7))
    p.hint_inverse(M(t), tinv)
    p.mul(M(t), M(tinv), M(nz))                  # nz = (top7 != 127), a boolean a jump accepts
    p.add(M(omnz), M(nz), K(1))
    p.mul(M(omnz), M(t), K(0))
    p.jump(M(nz), K(Label(skip)), FP(0))
    p.add(M(low), K(0), K(0))
    p.label(skip)
    f.forget_pointers()


def sumcheck_round(f, fs, claimed, grinding_bits, challenge_dst):
    """one round of sumcheck_verify_with_grinding (whir.py:210-220), degree 2 -> the new claimed sum"""
    poly = fs.receive_ef(3)
    tmp = f.new_ef()
    f.dot(absolute(REPEATED_ONES_PTR), poly, tmp, 3, be=True)     # sum_continuous_ef
    f.add(tmp, poly, claimed)                                     # polynomial_sum_at_0_and_1 == claimed_sum
    fs.grinding(grinding_bits)
    rand = fs.rate()                                              # fs_sample_ef
    f.copy5(rand, challenge_dst)
    pw = f.powers(rand, 3)
    out = f.new_ef()
    f.dot(poly, pw, out, 3)                                       # univariate_polynomial_eval
    return out


def statement_values(T):
    """the values of the PCS statement in statement order (recursion.py:478-518 = stacked_pcs_global_statements): (kind, data) with kind
    'at' (raw-transcript offset of one value), 'run' (offset, count: consecutive values), 'pm', 'zero', 'const' (a base constant)"""
    out = [("at", T.off_value_memory), ("at", T.off_value_memory_acc), ("pm", None), ("at", T.off_value_bytecode_acc), ("zero", None),
           ("const", T.ending_pc)]
    for t in T.order:
        n_flat, n_shift = TABLE_COLUMNS[t]
        out += [("at", off) for _, off in T.logup[t]]
        if n_shift:
            out.append(("run", (T.off_inner[t] + DIM * n_flat, n_shift)))
        out.append(("run", (T.off_inner[t], n_flat)))
    return out


def statement_sum(f, S, T, cl, tr, pw, acc, pm_point):
    """whir_sum (recursion.py:473-518): the OOD part `acc` + sum_i value_i * gen^(n_ood + i); consecutive values are one dot product"""
    k = S.oods[0]
    vals = statement_values(T)
    i = 0
    while i < len(vals):
        kind, d = vals[i]
        if kind == "at":        # merge values that lie 5 words apart into a run
            n = 1
            while i + n < len(vals) and vals[i + n][0] == "at" and vals[i + n][1] == d + DIM * n:
                n += 1
            term = f.new_ef()
            f.dot(tr + d, pw + DIM * k, term, n)
            acc, k, i = f.add(acc, term), k + n, i + n
            continue
        if kind == "run":
            off, n = d
            term = f.new_ef()
            f.dot(tr + off, pw + DIM * k, term, n)
            acc, k = f.add(acc, term), k + n
        elif kind == "pm":      # public_memory_eval = <public input, eq(public_memory_random_point, .)> (recursion.py:464-467)
            eq = f.eq_mle(pm_point, T.lpm)
            pm = f.new_ef()
            f.dot(cl + S.c_public_input, eq, pm, 1 << T.lpm, be=True)
            acc, k = f.add(acc, f.mul(pm, pw + DIM * k)), k + 1
        elif kind == "const":   # embed_in_ef(ENDING_PC) * randomness
            cell = f.const(d)
            term = f.new_ef()
            f.dot(fp(cell), pw + DIM * k, term, 1, be=True)
            acc, k = f.add(acc, term), k + 1
        else:                   # STARTING_PC = 0: nothing to add, the power is consumed
            k += 1
        i += 1
    assert k == S.oods[0] + T.n_values
    return acc


def next_mle(f, x, y, n):
    """next_mle_const (utils.py:726-767)"""
    one = absolute(ONE_EF_PTR)
    eq_prefix = f.new_ef(n + 1)
    f.set_one(eq_prefix)
    for i in range(n):
        eq_i = f.new_ef()
        f.poly_eq(x + DIM * i, y + DIM * i, eq_i, 1)
        f.mul(eq_prefix + DIM * i, eq_i, eq_prefix + DIM * (i + 1))
    low = f.new_ef(n + 1)
    f.set_one(low + DIM * n)
    for i in range(n):
        idx = n - 1 - i
        omy = f.sub(one, y + DIM * idx)
        f.mul(low + DIM * (idx + 1), f.mul(x + DIM * idx, omy), low + DIM * idx)
    total = absolute(ZERO_VEC_PTR)
    for a in range(n):
        omx = f.sub(one, x + DIM * a)
        term = f.mul(f.mul(eq_prefix + DIM * a, f.mul(omx, y + DIM * a)), low + DIM * (a + 1))
        total = f.add(total, term)
    px, py = f.new_ef(), f.new_ef()
    f.poly_eq(absolute(REPEATED_ONES_PTR), x, px, n, be=True)          # product_first_n_const
    f.poly_eq(absolute(REPEATED_ONES_PTR), y, py, n, be=True)
    return f.add(total, f.mul(px, py))


def statement_weights(f, S, T, cl, rand, pw, air_point, pm_point, gkr_point):
    """recursion.py:534-652: sum_i gen^(n_ood + i) * weight_i(folding randomness), weight = eq(top coordinates, selector) x eq / next of the
    statement's point at the inner coordinates.  The reference evaluates one location prefix per value (bit decomposition + poly_eq_be);
    here the values of a statement have consecutive selectors, so their prefixes are a slice of ONE eq table over the low q <= 6 selector
    bits, shared by the table, times a prefix over the remaining high bits per aligned block: a dot product per block."""
    p, n = f.p, S.n
    bits_cache, low_cache = {}, {}

    def const_bits(value, nbits):
        key = (value, nbits)
        if key not in bits_cache:
            c = f.alloc(max(nbits, 1))
            for j in range(nbits):
                p.add(K(0), K((value >> (nbits - 1 - j)) & 1), M(c + j))
            bits_cache[key] = c
        return bits_cache[key]

    def prefix(value, nbits):
        """eq(rand[:nbits], bits of value) (multilinear_location_prefix, recursion.py:658-661)"""
        if nbits == 0:
            return absolute(ONE_EF_PTR)
        out = f.new_ef()
        f.poly_eq(fp(const_bits(value, nbits)), rand, out, nbits, be=True)
        return out

    def run_weight(k, sel0, m, n_top):
        """sum_{j < m} pw[k + j] * eq(rand[:n_top], sel0 + j)"""
        q = min(6, n_top)
        if (n_top, q) not in low_cache:
            low_cache[(n_top, q)] = f.eq_mle(rand + DIM * (n_top - q), q)
        low = low_cache[(n_top, q)]
        acc, j = None, 0
        while j < m:
            blk, lo = (sel0 + j) >> q, (sel0 + j) & ((1 << q) - 1)
            ln = min(m - j, (1 << q) - lo)
            d = f.new_ef()
            f.dot(pw + DIM * (k + j), low + DIM * lo, d, ln)
            term = f.mul(d, prefix(blk, n_top - q)) if n_top > q else d
            acc = term if acc is None else f.add(acc, term)
            j += ln
        return acc

    k = S.oods[0]
    mem = 1 << T.log_memory
    gkr_tail = lambda nv: gkr_point + DIM * (T.gkr_n_vars - nv)  # noqa: E731 — from_end(point_gkr, nv)
    inner = lambda nv: rand + DIM * (n - nv)                               # noqa: E731
    # memory and its accumulator (selectors 0, 1), the public memory, the bytecode accumulator
    eqf = f.new_ef()
    f.poly_eq(inner(T.log_memory), gkr_tail(T.log_memory), eqf, T.log_memory)
    s = f.mul(run_weight(k, 0, 2, n - T.log_memory), eqf)
    k += 2
    eqf = f.new_ef()
    f.poly_eq(inner(T.lpm), pm_point, eqf, T.lpm)
    s = f.add(s, f.mul(f.mul(pw + DIM * k, prefix(0, n - T.lpm)), eqf))
    k += 1
    eqf = f.new_ef()
    f.poly_eq(inner(T.log_bytecode), gkr_tail(T.log_bytecode), eqf, T.log_bytecode)
    s = f.add(s, f.mul(f.mul(pw + DIM * k, prefix((2 * mem) >> T.log_bytecode, n - T.log_bytecode)), eqf))
    k += 1
    soff = 2 * mem + (1 << max(T.log_bytecode, T.log_rows[T.order[0]]))
    for t in T.order:
        nv = T.log_rows[t]
        n_flat, n_shift = TABLE_COLUMNS[t]
        if t == 0:              # pc at the first and at the last row of the execution table: full-index prefixes
            s = f.add(s, f.mul(pw + DIM * k, prefix(soff, n)))
            s = f.add(s, f.mul(pw + DIM * (k + 1), prefix(soff + (1 << nv) - 1, n)))
            k += 2
        base = soff >> nv
        # the logup column evaluations at from_end(point_gkr, nv): runs of consecutive columns
        eqf = f.new_ef()
        f.poly_eq(gkr_tail(nv), inner(nv), eqf, nv)
        cols = [c for c, _ in T.logup[t]]
        w, i = None, 0
        while i < len(cols):
            m = 1
            while i + m < len(cols) and cols[i + m] == cols[i] + m:
                m += 1
            rw = run_weight(k, base + cols[i], m, n - nv)
            w = rw if w is None else f.add(w, rw)
            k, i = k + m, i + m
        s = f.add(s, f.mul(w, eqf))
        # the column evaluations behind the AIR sumcheck at all_challenges[:nv]: shifted columns (next_mle), then all flat columns
        point = air_point
        if n_shift:
            s = f.add(s, f.mul(run_weight(k, base, n_shift, n - nv), next_mle(f, point, inner(nv), nv)))
            k += n_shift
        eqf = f.new_ef()
        f.poly_eq(point, inner(nv), eqf, nv)
        s = f.add(s, f.mul(run_weight(k, base, n_flat, n - nv), eqf))
        k += n_flat
        soff += n_flat << nv
    assert k == S.oods[0] + T.n_values
    return s


LOOKUPS = {0: [(2, 5, 1), (3, 6, 1), (4, 7, 1)], 1: [(6, 14, 5), (7, 19, 5), (13, 24, 5)], 2: [(6, 9, 4), (7, 13, 4), (1, 17, 8), (2, 93, 16)]}
# (index column, first value column, number of value columns) per lookup into memory and table (LOOKUPS_INDEXES / LOOKUPS_VALUES)
N_INSTRUCTION_COLUMNS, LOGUP_BYTECODE_DOMAINSEP = 12, 2


def sumcheck_reversed(f, fs, claimed, n_rounds, degree, challenges):
    """sumcheck_verify_reversed_helper_const (whir.py:199-207): the challenge of round r lands on challenges[n_rounds - 1 - r]"""
    n_co = degree + 1
    for r in range(n_rounds):
        poly = fs.receive_ef(n_co)
        tmp = f.new_ef()
        f.dot(absolute(REPEATED_ONES_PTR), poly, tmp, n_co, be=True)
        f.add(tmp, poly, claimed)
        rand = fs.rate()
        f.copy5(rand, challenges + DIM * (n_rounds - 1 - r))
        nxt = f.new_ef()
        f.dot(poly, f.powers(rand, n_co), nxt, n_co)
        claimed = nxt
    return claimed


def head_verify(f, S, T, cl, fs):
    """recursion.py:48-378: the transcript from its first word — public input and domain separator observed, the dimensions checked
    against the ones this program is assembled for, the stacked commitment, logup_c and the alphas, verify_gkr_quotient, and the logup
    statement (the GKR claims rebuilt from the column evaluations the prover sends).
    -> dict(root, ood_points, ood_evals, logup_c, gkr_point)"""
    p = f.p
    one, zero = absolute(ONE_EF_PTR), absolute(ZERO_VEC_PTR)
    n_pi_chunks = ceil_div(1 << T.lpm, DIGEST_LEN)
    for j in range(n_pi_chunks):                        # fs_observe(inner_public_memory), fs_observe(bytecode_hash_domsep)
        fs._absorb(cl + S.c_public_input + DIGEST_LEN * j)
    fs._absorb(cl + S.c_domsep)
    dims = fs.receive_chunks(1)
    for i, want in enumerate([T.rate, T.log_memory, T.public_input_len] + T.log_rows + [0, 0]):
        f.store(dims + i, K(want))                      # the dimensions are assembly-time constants of this program: asserted
    root = fs.receive_chunks(1)                         # parse_commitment
    ood_points = fs.sample_chunks(ceil_div(DIM * T.n_ood, 8))
    ood_evals = fs.receive_ef(T.n_ood)
    logup_c = fs.rate()
    fs.duplex()
    alphas = fs.sample_chunks(ceil_div(DIM * 4, 8))
    aeq = f.eq_mle(alphas, 4)                           # logup_alphas_eq_poly: 16 values
    # ---- verify_gkr_quotient (recursion.py:684-749)
    ng = T.gkr_n_vars
    nums, dens = fs.receive_ef(32), fs.receive_ef(32)
    quots = f.new_ef(32)
    for k in range(32):
        f.ext("mul", dens + DIM * k, quots + DIM * k, nums + DIM * k)      # div_extension: the unknown factor is solved for
    f.dot(absolute(REPEATED_ONES_PTR), quots, zero, 32, be=True)          # the quotient sum is zero (set_to_5_zeros(quotient_gkr), :108)
    point = fs.sample_chunks(ceil_div(DIM * 5, 8))
    eq32 = f.eq_mle(point, 5)
    cnum, cden = f.new_ef(), f.new_ef()
    f.dot(nums, eq32, cnum, 32)
    f.dot(dens, eq32, cden, 32)
    for i in range(5, ng):                              # verify_gkr_quotient_step
        fs.duplex()
        alpha = fs.rate()
        target = f.add(cnum, f.mul(alpha, cden))
        pp = f.new_ef(i + 1)
        value = sumcheck_reversed(f, fs, target, i, 3, pp)
        ie = fs.receive_ef(4)
        a_num, b_num, a_den, b_den = ie, ie + DIM, ie + 2 * DIM, ie + 3 * DIM
        sum_num = f.add(f.mul(a_num, b_den), f.mul(b_num, a_den))
        sum_den = f.mul(a_den, b_den)
        eqf = f.new_ef()
        f.poly_eq(point, pp, eqf, i)
        f.mul(f.add(sum_num, f.mul(sum_den, alpha)), eqf, value)
        beta = fs.rate()
        eqb = f.eq_mle(beta, 1)
        cnum, cden = f.new_ef(), f.new_ef()
        f.dot(ie, eqb, cnum, 2)
        f.dot(ie + 2 * DIM, eqb, cden, 2)
        f.copy5(beta, pp + DIM * i)
        point = pp
    gp = point
    # ---- the logup statement (recursion.py:110-378)
    bits_cache = {}

    def prefix(value, nbits):
        if nbits == 0:
            return one
        key = (value, nbits)
        if key not in bits_cache:
            c = f.alloc(nbits)
            for j in range(nbits):
                p.add(K(0), K((value >> (nbits - 1 - j)) & 1), M(c + j))
            bits_cache[key] = c
        out = f.new_ef()
        f.poly_eq(fp(bits_cache[key]), gp, out, nbits, be=True)
        return out

    def mle_of_index(pt, n):                            # mle_of_01234567_etc (utils.py:206-217), from the last coordinate up
        e = zero
        for k in range(n - 1, -1, -1):
            x = pt + DIM * k
            c = f.new_ef()
            f.ext("add", fp(f.const(1 << (n - 1 - k))), e, c, be=True)
            e = f.add(f.mul(f.sub(one, x), e), f.mul(x, c))
        return e

    def base_times(value, x):                           # mul_base_extension_ret
        out = f.new_ef()
        f.dot(fp(f.const(value)), x, out, 1, be=True)
        return out

    tail = lambda nv: gp + DIM * (ng - nv)              # noqa: E731
    mem_prefix = prefix(0, ng - T.log_memory)
    value_acc, value_memory = fs.receive_ef(1), fs.receive_ef(1)
    assert value_acc.k == T.off_value_memory_acc and value_memory.k == T.off_value_memory
    r_num = f.sub(zero, f.mul(mem_prefix, value_acc))
    buf = f.new_ef(2)
    f.copy5(value_memory, buf)
    f.copy5(mle_of_index(tail(T.log_memory), T.log_memory), buf + DIM)
    fp_mem = f.new_ef()
    f.dot(buf, aeq, fp_mem, 2)                          # fingerprint_2 with the memory domain separator 0
    r_den = f.mul(mem_prefix, f.sub(logup_c, fp_mem))
    offset = 1 << T.log_memory
    lb, n_cycles_log = T.log_bytecode, T.log_rows[0]
    lbp = max(lb, n_cycles_log)
    bc_prefix, bcp_prefix = prefix(offset >> lb, ng - lb), prefix(offset >> lbp, ng - lbp)
    value_bc_acc = fs.receive_ef(1)
    assert value_bc_acc.k == T.off_value_bytecode_acc
    r_num = f.sub(r_num, f.mul(bc_prefix, value_bc_acc))
    dom = base_times(LOGUP_BYTECODE_DOMAINSEP, aeq + DIM * 15)
    inner = f.add(cl + S.c_bytecode_value, f.add(f.mul(mle_of_index(tail(lb), lb), aeq + DIM * N_INSTRUCTION_COLUMNS), dom))
    r_den = f.add(r_den, f.mul(bc_prefix, f.sub(logup_c, inner)))
    if lbp > lb:                                        # mle_of_zeros_then_ones_pow2 (utils.py:700-709)
        pt = tail(lbp)
        prod = f.sub(one, pt)
        for k in range(1, lbp - lb):
            prod = f.mul(prod, f.sub(one, pt + DIM * k))
        r_den = f.add(r_den, f.mul(bcp_prefix, f.sub(one, prod)))
    offset += 1 << lbp
    for t in T.order:
        nv = T.log_rows[t]
        if t == 0:
            pre = prefix(offset >> nv, ng - nv)
            on_pc, instr = fs.receive_ef(1), fs.receive_ef(N_INSTRUCTION_COLUMNS)
            r_num = f.add(r_num, pre)
            fpb = f.new_ef()
            f.dot(instr, aeq, fpb, N_INSTRUCTION_COLUMNS)                 # fingerprint_bytecode (:674-681)
            fpb = f.add(f.add(fpb, f.mul(on_pc, aeq + DIM * N_INSTRUCTION_COLUMNS)), dom)
            r_den = f.add(r_den, f.mul(pre, f.sub(logup_c, fpb)))
            offset += 1 << nv
        pre = prefix(offset >> nv, ng - nv)
        on_sel, on_data = fs.receive_ef(1), fs.receive_ef(1)
        assert on_sel.k == T.off_bus_selector[t] and on_data.k == T.off_bus_data[t]
        r_num = f.add(r_num, f.mul(pre, on_sel))
        r_den = f.add(r_den, f.mul(pre, on_data))
        offset += 1 << nv
        for _, _, n_val in LOOKUPS[t]:
            index_eval = fs.receive_ef(1)
            for i in range(n_val):
                value_eval = fs.receive_ef(1)
                pre = prefix(offset >> nv, ng - nv)
                r_num = f.add(r_num, pre)
                buf = f.new_ef(2)
                f.copy5(value_eval, buf)
                if i == 0:
                    f.copy5(index_eval, buf + DIM)
                else:
                    f.ext("add", fp(f.const(i)), index_eval, buf + DIM, be=True)   # add_base_extension_ret(i, index_eval)
                fpv = f.new_ef()
                f.dot(buf, aeq, fpv, 2)
                r_den = f.add(r_den, f.mul(pre, f.sub(logup_c, fpv)))
                offset += 1 << nv
    # + mle_of_zeros_then_ones(point_gkr, offset, n_vars) (utils.py:670-697): the padding of the logup vector
    assert offset < (1 << ng)
    res = one
    for i in range(ng):
        x = gp + DIM * (ng - 1 - i)
        res = f.mul(x, res) if (offset >> i) & 1 else f.add(f.mul(f.sub(one, x), res), x)
    r_den = f.add(r_den, res)
    f.copy5(r_num, cnum)                                # the GKR claims are what the column evaluations give (:376-377)
    f.copy5(r_den, cden)
    assert fs.off == T.air_off
    return dict(root=root, ood_points=ood_points, ood_evals=ood_evals, logup_c=logup_c, gkr_point=gp, aeq=aeq)


def air_sumcheck_verify(f, S, T, cl, tr, fs, logup_c, gkr_point, evaluators=None):
    """recursion.py:383-467: bus_beta / air_alpha / eta, the initial sum from the bus evaluations, the batched AIR sumcheck
    (sumcheck_verify_reversed, whir.py:184-207), the tables' column evaluations and the back-loaded check against the claimed constraint
    evaluations, then the public-memory point.  -> (all_challenges, public_memory_random_point)"""
    zero = absolute(ZERO_VEC_PTR)
    bus_beta = fs.rate()
    fs.duplex()
    air_alpha = fs.rate()           # (its powers only enter evaluate_air_constraints)
    fs.duplex()
    eta_pw = f.powers(fs.rate(), 3)
    claimed = zero
    for k, t in enumerate(T.order):
        num, den = tr + T.off_bus_selector[t], tr + T.off_bus_data[t]
        bfv = num if t == 0 else f.sub(zero, num)                              # opposite_extension_ret for the tables that pull
        bfv = f.add(bfv, f.mul(bus_beta, f.sub(den, logup_c)))
        claimed = f.add(claimed, f.mul(eta_pw + DIM * k, bfv))
    n_co = T.air_degree + 1
    all_ch = f.new_ef(T.n_max)
    for r in range(T.n_max):
        poly = fs.receive_ef(n_co)
        tmp = f.new_ef()
        f.dot(absolute(REPEATED_ONES_PTR), poly, tmp, n_co, be=True)
        f.add(tmp, poly, claimed)                                               # p(0) + p(1) == the running claim
        rand = fs.rate()
        f.copy5(rand, all_ch + DIM * (T.n_max - 1 - r))
        nxt = f.new_ef()
        f.dot(poly, f.powers(rand, n_co), nxt, n_co)
        claimed = nxt
    check = None
    for k, t in enumerate(T.order):
        nv = T.log_rows[t]
        inner = fs.receive_ef(sum(TABLE_COLUMNS[t]))
        assert inner.kind == "at" and inner.k in (T.off_inner[t] - T.air_off, T.off_inner[t])
        eq_val = f.new_ef()
        f.poly_eq(gkr_point + DIM * (T.gkr_n_vars - nv), all_ch, eq_val, nv)
        if T.n_max > nv:            # product_first_n (utils.py:70-83)
            k_t = f.new_ef()
            f.poly_eq(absolute(REPEATED_ONES_PTR), all_ch + DIM * nv, k_t, T.n_max - nv, be=True)
            lhs = f.mul(eta_pw + DIM * k, k_t)
        else:
            lhs = f.mul(eta_pw + DIM * k, absolute(ONE_EF_PTR))
        if evaluators is None:      # the constraint evaluation is a claim
            air_eval = cl + S.c_air_evals + DIM * t
        else:                       # evaluate_air_constraints (recursion.py:434, 777-787)
            from . import air_eval as ae
            if "apw" not in evaluators:
                evaluators.update(apw=f.powers(air_alpha, ae.MAX_ALPHA), bus_beta=bus_beta)
            A = ae.Alg(f, absolute(ONE_EF_PTR), zero)
            n_flat = TABLE_COLUMNS[t][0]
            if t == 0:
                air_eval = ae.eval_execution(A, inner, inner + DIM * n_flat, evaluators)
            elif t == 1:
                air_eval = ae.eval_extension_op(A, inner, inner + DIM * n_flat, evaluators)
            else:
                air_eval = ae.eval_poseidon16(A, inner, evaluators, evaluators["mds_window"])
        term = f.mul(lhs, f.mul(eq_val, air_eval))
        check = term if check is None else f.add(check, term)
    f.copy5(check, claimed)                                                     # the sumcheck's final value is what the tables give
    pm_point = fs.sample_chunks(ceil_div(DIM * T.lpm, 8))                       # fs_sample_many_ef(INNER_PUBLIC_MEMORY_LOG_SIZE)
    return all_ch, pm_point


def build_program(cfg, n_children=4, log_size=None, statement=None, air=False, head=False, evaluators=False):
    """-> vm.Bytecode for `n_children` proofs of the WhirConfig `cfg` (a dict, capi.WhirConfig.to_dict()).  statement (a Statement): the
    program also assembles the PCS statement (recursion.py:469-518, 534-652) instead of taking its two sums from the claims; air: it
    starts in front of the batched AIR sumcheck (recursion.py:383-467) and samples that sumcheck's challenges and the public-memory point
    itself; head: it replays the transcript from its first word (recursion.py:48-378: GKR quotient, logup statement) — the whole verifier
    of one child except evaluate_air_constraints, whose three results stay claims — unless evaluators: then the three constraint
    polynomials are evaluated in the VM as well (programs/air_eval.py) and the program is the reference's `recursion()` whole."""
    assert head or not evaluators
    S = Shape(cfg, n_children, statement, air, head)
    S.evaluators = evaluators
    T = statement
    p = Program()
    f = Fn(p, 0, ZERO_VEC_PTR, ONE_EF_PTR, REPEATED_ONES_PTR)
    A = lambda off: MAIN_FP + off  # noqa: E731 — absolute address of a main-frame cell
    NC, R = n_children, S.n_rounds

    # ================================================================ main ================================================================
    # build_preamble_memory (utils.py:11-29)
    zv = f.const(ZERO_VEC_PTR)
    for i in range(ZERO_VEC_LEN):
        p.deref(zv, i, K(0))
    sds = f.const(SDS_PTR)
    p.deref(sds, 0, K(1))
    for i in range(1, DIGEST_LEN):
        p.deref(sds, i, K(0))
    one = f.const(ONE_EF_PTR)
    p.deref(one, 0, K(1))
    for i in range(1, DIM):
        p.deref(one, i, K(0))
    ones = f.const(REPEATED_ONES_PTR)
    for i in range(NUM_REPEATED_ONES):
        p.deref(ones, i, K(1))
    # the claims buffer and the raw proof transcripts (hinted, as main.py:38-42 / recursion.py:49-52)
    claims = f.alloc()
    p.hint_request_memory(claims, K(NC * S.claim_words))
    p.hint_witness("claims", claims, indirect=True)
    tfull = []
    for c in range(NC):
        t = f.alloc()
        p.hint_request_memory(t, K(S.transcript_words))
        p.hint_witness("proof_transcript", t, indirect=True)
        tfull.append(t)
    tbase = []                                   # where whir_open starts reading
    for c in range(NC):
        t = f.alloc()
        p.add(M(tfull[c]), K((0 if head else T.air_off if air else T.off_whir) if T else 0), M(t))
        tbase.append(t)

    # per-round arrays shared by the children: iteration i = child * q + j of a round's loop owns entry i
    folds_all = [f.alloc(DIM * NC * S.queries[r]) for r in range(R)]           # the folded leaves (whir.py:302-311)
    circle_all = [f.alloc(NC * S.queries[r]) for r in range(R)]                # all_circle_values[r]
    s6s_all = [f.alloc(DIM * NC * S.queries[r]) for r in range(R)]             # whir.py:143-147
    # per child and round: the sampled query words (padded to whole chunks) and the descriptor its segments read
    sampled = [[f.alloc(ceil_div(S.queries[r], 8) * 8) for r in range(R + 1)] for c in range(NC)]
    DESC_ROOT, DESC_EQ, DESC_COEFFS, DESC_RAND = 0, 1, 2, 3
    desc = [[f.alloc(4) for r in range(R + 1)] for c in range(NC)]

    mds_window = None
    if evaluators:
        from .air_eval import write_mds_window
        mds_window = write_mds_window(f)
    ch = []                                                                    # assembly-time state of every child
    for c in range(NC):
        cl = at(claims, c * S.claim_words)
        st = dict(cl=cl, rand=fp(f.alloc(DIM * S.n)))                          # folding_randomness_global
        st["fs"] = Fs(f, absolute(ZERO_VEC_PTR) if head else cl + S.c_fs, tbase[c])     # (fs_new: the zero state)
        st["air_point"], st["pm_point"] = (cl + S.c_air_point, cl + S.c_pm_point) if T is not None and not air else (None, None)
        hd = dict(root=cl + S.c_root, ood_points=cl + S.c_ood_points, ood_evals=cl + S.c_ood_evals) if not head else None
        if head:
            hd = head_verify(f, S, T, cl, st["fs"])
        elif air:
            hd.update(logup_c=cl + S.c_logup_c, gkr_point=cl + S.c_gkr_point)
        elif T is not None:
            hd.update(gkr_point=cl + S.c_gkr_point)
        st["hd"] = hd
        if air:
            ev = dict(aeq=hd["aeq"], mds_window=mds_window) if evaluators else None
            st["air_point"], st["pm_point"] = air_sumcheck_verify(f, S, T, cl, at(tfull[c], 0), st["fs"], hd["logup_c"], hd["gkr_point"], ev)
            assert st["fs"].off == T.off_whir - (0 if head else T.air_off)
            st["fs"].duplex()                                                  # recursion.py:470
        # recursion.py:472-475: the combination randomness of the first constraint set; only its OOD powers are needed here
        gen = st["fs"].rate()
        st["pw0"] = f.powers(gen, S.oods[0] if T is None else 1 << (S.oods[0] + T.n_values - 1).bit_length())
        ood_sum = f.new_ef()
        f.dot(hd["ood_evals"], st["pw0"], ood_sum, S.oods[0])
        if T is None:
            st["claimed"] = f.add(ood_sum, cl + S.c_stmt_sum)                  # whir_sum
        else:
            st["claimed"] = statement_sum(f, S, T, cl, at(tfull[c], 0), st["pw0"], ood_sum, st["pm_point"])
        st["root"] = hd["root"]
        st["ood_points"], st["comb"], st["roots"] = [], [], []
        ch.append(st)

    def sumcheck_rounds(st, count, bits, first_var):
        for k in range(count):
            st["claimed"] = sumcheck_round(f, st["fs"], st["claimed"], bits, st["rand"] + DIM * (first_var + k))

    def stir_prepare(c, st, r):
        """sample_stir_indexes_and_fold (whir.py:266-313) up to its loops: grinding, the query words, the eq table of the folding
        randomness, and the descriptor the (child, query) segments of round r read"""
        fs = st["fs"]
        fs.grinding(S.query_grinding[r])
        nch = ceil_div(S.queries[r], 8)
        smp = fs.sample_chunks(nch)                                            # fs_sample_queries :197-209
        for j in range(nch):
            f.copy8(smp + 8 * j, fp(sampled[c][r] + 8 * j))
        var0 = sum(S.fold[:r])
        eq = f.eq_mle(st["rand"] + DIM * var0, S.fold[r])
        p.add(K(0), M(f.ptr(st["root"])), M(desc[c][r] + DESC_ROOT))
        p.add(K(0), M(f.ptr(eq)), M(desc[c][r] + DESC_EQ))

    # ---- whir_round x n_rounds (whir.py:316-368), children in lock step ---------------------------------------------------------------
    var = 0
    for r in range(R):
        for c, st in enumerate(ch):
            fs = st["fs"]
            sumcheck_rounds(st, S.fold[r], S.folding_grinding[r], var)
            new_root = fs.receive_chunks(1)                                    # parse_commitment :377-391
            ood_points = fs.sample_chunks(ceil_div(DIM * S.oods[r + 1], 8))
            st["ood_evals"] = fs.receive_ef(S.oods[r + 1])
            st["ood_points"].append(ood_points)
            stir_prepare(c, st, r)
            st["roots"].append(new_root)
        f.call_loop(f"merkle_loop_{r}", f"@merkle_frame_{r}", [K(NC * S.queries[r]), K(A(folds_all[r])), K(A(circle_all[r]))])
        for c, st in enumerate(ch):
            fs = st["fs"]
            st["root"] = st["roots"][-1]
            fs.duplex()
            gen = fs.rate()
            q, o = S.queries[r], S.oods[r + 1]
            comb = f.powers(gen, 1 << (q + o - 1).bit_length())                # powers(): next power of two (utils.py:38-46)
            st["comb"].append(comb)
            s0, s1 = f.new_ef(), f.new_ef()
            f.dot(st["ood_evals"], comb, s0, o)
            f.dot(fp(folds_all[r] + DIM * c * q), comb + DIM * o, s1, q)
            st["claimed"] = f.add(st["claimed"], f.add(s0, s1))
        var += S.fold[r]
    # ---- the final round (whir.py:71-101) --------------------------------------------------------------------------------------------------
    for c, st in enumerate(ch):
        fs = st["fs"]
        sumcheck_rounds(st, S.fold[R], S.folding_grinding[R], var)
        st["coeffs"] = fs.receive_ef(1 << S.n_final)
        stir_prepare(c, st, R)
        p.add(K(0), M(f.ptr(st["coeffs"])), M(desc[c][R] + DESC_COEFFS))
    f.call_loop(f"merkle_loop_{R}", f"@merkle_frame_{R}", [K(NC * S.queries[R]), K(0), K(0)])
    var += S.fold[R]
    for c, st in enumerate(ch):
        sumcheck_rounds_plain = st["fs"]
        for k in range(S.n_final):                                             # sumcheck_verify (no grinding)
            st["claimed"] = sumcheck_round(f, sumcheck_rounds_plain, st["claimed"], 0, st["rand"] + DIM * (var + k))
        st["end_sum"] = st["claimed"]
        for r in range(R):                                                     # what the s6s segments of round r read
            p.add(K(0), FP(st["rand"].a + DIM * sum(S.fold[:r + 1])), M(desc[c][r] + DESC_RAND))
    # ---- the constraint weights at the folding randomness (whir.py:113-156) ------------------------------------------------------------------
    for r in range(R):
        f.call_loop(f"s6s_loop_{r}", f"@s6s_frame_{r}", [K(NC * S.queries[r]), K(A(circle_all[r])), K(A(s6s_all[r]))])
    for c, st in enumerate(ch):
        cl = st["cl"]
        rec = f.new_ef(S.oods[0])
        for i in range(S.oods[0]):
            ex = expand_from_univariate_ext(f, st["hd"]["ood_points"] + DIM * i, S.n)
            f.poly_eq(ex, st["rand"], rec + DIM * i, S.n)
        s = f.new_ef()
        f.dot(rec, st["pw0"], s, S.oods[0])
        for r in range(R):
            my_rand = st["rand"] + DIM * sum(S.fold[:r + 1])
            q, o, nr = S.queries[r], S.oods[r + 1], S.n_rem[r]
            rec = f.new_ef(o)
            for j in range(o):
                ex = expand_from_univariate_ext(f, st["ood_points"][r] + DIM * j, nr)
                f.poly_eq(ex, my_rand, rec + DIM * j, nr)
            summed_ood, s7 = f.new_ef(), f.new_ef()
            f.dot(rec, st["comb"][r], summed_ood, o)
            f.dot(fp(s6s_all[r] + DIM * c * q), st["comb"][r] + DIM * o, s7, q)
            s = f.add(summed_ood, f.add(s, s7))
        # eval_multilinear_coeffs_rev (utils.py:181-192) at the final sumcheck's challenges
        nf = S.n_final
        basis = f.new_ef(1 << nf)
        f.set_one(basis)
        final_rand = st["rand"] + DIM * (S.n - nf)
        for k in range(nf):
            pt = f.new_ef()
            f.copy5(final_rand + DIM * k, pt)
            for j in range(1 << k):
                f.mul(basis + DIM * j, pt, basis + DIM * (j + (1 << k)))
        final_value = f.new_ef()
        f.dot(st["coeffs"], basis, final_value, 1 << nf)
        # recursion.py:534-654 with the statement's share taken from the claim: (s + statement_weights) * final_value == end_sum
        total = f.add(s, cl + S.c_stmt_weights) if T is None else f.add(s, statement_weights(f, S, T, cl, st["rand"], st["pw0"], st["air_point"], st["pm_point"], st["hd"]["gkr_point"]))
        f.mul(total, final_value, st["end_sum"])
        for k in range(S.n):
            f.copy5(st["rand"] + DIM * k, cl + S.c_rand + DIM * k)             # folding_randomness_global == the claim's
    # ---- slice_hash_with_iv(claims, n_chunks, public input) (hashing.py:81-89) -------------------------------------------------------------------
    nch = NC * S.claim_words // DIGEST_LEN
    states = f.alloc((nch - 1) * DIGEST_LEN)
    f.compress(absolute(ZERO_VEC_PTR), at(claims, 0), fp(states))
    for j in range(1, nch):
        dst = absolute(0) if j == nch - 1 else fp(states + 8 * j)
        f.compress(fp(states + 8 * (j - 1)), at(claims, 8 * j), dst)
    p.return_from_main(f.alloc())
    p.starting_frame_memory = f.top

    # ================================================================ the loops ============================================================
    info = dict(shape=S, frames={}, main_frame_size=f.top)
    for r in range(R + 1):
        info["frames"][f"merkle_{r}"] = emit_merkle_loop(p, S, r, [A(sampled[c][r]) for c in range(NC)], [A(desc[c][r]) for c in range(NC)])
    for r in range(R):
        info["frames"][f"s6s_{r}"] = emit_s6s_loop(p, S, r, [A(desc[c][r]) for c in range(NC)])
    info["n_instructions"] = p.here()
    bc = p.finalize(log_size)
    bc.info = info
    return bc


def expand_from_univariate_ext(f, alpha, n):
    """expand_from_univariate_ext_const (utils.py:160-165): alpha, alpha^2, alpha^4, ..."""
    res = f.new_ef(n)
    f.copy5(alpha, res)
    for i in range(n - 1):
        f.mul(res + DIM * i, res + DIM * i, res + DIM * (i + 1))
    return res


def emit_child_table(p, g, S, r, label, cells):
    """match_range over the iteration: entry i sets, for cell, per_child, per_query in cells: m[fp + cell] = per_child[c] + per_query * j
    with (c, j) = divmod(i, q).  Entries are len(cells) + 1 instructions."""
    q = S.queries[r]
    block = len(cells) + 1
    after = g.dispatch(2, block, label)      # frame cell 2 = the iteration
    table = []
    for i in range(S.n_children * q):
        c, j = divmod(i, q)
        table.append([(cell, per_child[c] + per_query * j) for cell, per_child, per_query in cells])
    return after, table, block


def emit_merkle_loop(p, S, r, sampled_abs, desc_abs):
    """One (child, query) pair of round r: decompose_and_verify_merkle_query (utils.py:537-626) + the fold of the opened leaf
    (whir.py:304-311) [+ the final polynomial at the query point, whir.py:91-99, in the last round].
    frame: [return pc, saved fp, i, end, folds_all, circle_all, locals...]"""
    N_ARGS = 4
    I, FOLDS, CIRCLE = 2, 4, 5
    final = r == S.n_rounds
    g = Fn(p, 2 + N_ARGS, ZERO_VEC_PTR, ONE_EF_PTR, REPEATED_ONES_PTR)
    label, size_label = f"merkle_loop_{r}", f"@merkle_frame_{r}"
    loop_prologue(p, g, label, N_ARGS)
    DESC, APTR = g.alloc(), g.alloc()
    after_child, child_table, child_block = emit_child_table(p, g, S, r, f"child_table_m{r}", [(DESC, desc_abs, 0), (APTR, sampled_abs, 1)])
    root_ptr, eq_ptr = g.alloc(), g.alloc()
    p.deref(DESC, 0, M(root_ptr))
    p.deref(DESC, 1, M(eq_ptr))
    a = g.alloc()
    p.deref(APTR, 0, M(a))                                        # the sampled word
    # nibbles of the canonical value (utils.py:539-555)
    nib = g.alloc(6)
    p.hint_decompose_bits_merkle_whir(FP(nib), M(a), K(4))
    for i in range(6):
        g.range_check(nib + i, 15)
    ps = nib
    for i in range(1, 6):
        t, s = g.alloc(), g.alloc()
        p.mul(M(nib + i), K(16 ** i), M(t))
        p.add(M(ps), M(t), M(s))
        ps = s
    diff, top7 = g.alloc(), g.alloc()
    p.add(M(diff), M(a), M(ps))
    p.mul(M(diff), K(127), M(top7))
    g.range_check(top7, 127)
    assert_if_127_then_zero(g, top7, ps)
    # the leaf and its digest (slice_hash_rtl, hashing.py:54-60)
    n_chunks = S.leaf_words[r] // DIGEST_LEN
    leaf = g.alloc(S.leaf_words[r])
    states = g.alloc((n_chunks - 1) * DIGEST_LEN)
    p.hint_witness("merkle_leaf", leaf)
    p.poseidon16(FP(leaf + (n_chunks - 2) * 8), FP(leaf + (n_chunks - 1) * 8), FP(states))
    for j in range(1, n_chunks - 1):
        p.poseidon16(FP(states + (j - 1) * 8), FP(leaf + (n_chunks - 2 - j) * 8), FP(states + j * 8))
    leaf_hash = states + (n_chunks - 2) * 8
    # the path, four levels per nibble through 16-entry jump tables (utils.py:561-624)
    h = S.height[r]
    path = g.alloc(h * DIGEST_LEN)
    n_nib = ceil_div(h, 4)
    mstates = g.alloc((n_nib - 1) * DIGEST_LEN)
    tables = []
    prod = None
    for k in range(n_nib):
        levels = min(4, h - 4 * k)
        temps, nibpow = g.alloc(3 * DIGEST_LEN), g.alloc()
        state_in = leaf_hash if k == 0 else mstates + (k - 1) * 8
        last = k == n_nib - 1
        if k == 0:
            p.hint_witness("merkle_path", path)
        g.dispatch(nib + k, levels + 2, f"merkle_table_{r}_{k}")
        tables.append(dict(k=k, levels=levels, temps=temps, nibpow=nibpow, state_in=state_in, out=None if last else mstates + k * 8))
        if prod is None:
            prod = nibpow
        else:
            t = g.alloc()
            p.mul(M(prod), M(nibpow), M(t))
            prod = t
    # the fold of the leaf with eq(folding randomness) (whir.py:306-311)
    leaf_ptr = g.alloc()
    p.add(K(0), FP(leaf), M(leaf_ptr))
    if not final:
        cptr = g.alloc()
        p.add(M(CIRCLE), M(I), M(cptr))
        p.deref(cptr, 0, M(prod))                                 # circle_values[i]
        i5, fptr = g.alloc(), g.alloc()
        p.mul(M(I), K(DIM), M(i5))
        p.add(M(FOLDS), M(i5), M(fptr))
        p.extension_op("dot_product", M(leaf_ptr), M(eq_ptr), M(fptr), size=1 << S.fold[r], is_be=(r == 0))
    else:
        # whir.py:91-99: the final polynomial at the query point equals the fold (univariate_eval_on_base, utils.py:168-178)
        fold = g.alloc(DIM)
        p.extension_op("dot_product", M(leaf_ptr), M(eq_ptr), FP(fold), size=1 << S.fold[r], is_be=False)
        nco = 1 << S.n_final
        pw = g.alloc(nco)
        p.add(K(0), K(1), M(pw))
        for i in range(nco - 1):
            p.mul(M(pw + i), M(prod), M(pw + i + 1))
        co_ptr, pw_ptr = g.alloc(), g.alloc()
        p.deref(DESC, 2, M(co_ptr))
        p.add(K(0), FP(pw), M(pw_ptr))
        p.extension_op("dot_product", M(pw_ptr), M(co_ptr), FP(fold), size=nco, is_be=True)
    loop_epilogue(p, g, label, size_label, N_ARGS)
    p.labels[size_label] = g.top
    # ---- tables --------------------------------------------------------------------------------------------------------------------------
    p.label(f"child_table_m{r}")
    for entry in child_table:
        for cell, value in entry:
            p.add(K(0), K(value), M(cell))
        p.jump(K(1), K(Label(after_child)), FP(0))
    for tb in tables:
        k, levels = tb["k"], tb["levels"]
        shift = 1 << (TWO_ADICITY - h + 4 * k)
        p.label(f"merkle_table_{r}_{k}")
        for v in range(16):
            start = p.here()
            cur = tb["state_in"]
            for lv in range(levels):
                sib = path + (4 * k + lv) * 8
                lastlv = lv == levels - 1
                if lastlv:
                    dst = M(root_ptr) if tb["out"] is None else FP(tb["out"])
                else:
                    dst = FP(tb["temps"] + 8 * lv)
                if (v >> lv) & 1:
                    p.poseidon16(FP(sib), FP(cur), dst)
                else:
                    p.poseidon16(FP(cur), FP(sib), dst)
                cur = tb["temps"] + 8 * lv
            p.add(K(0), K(pow(ROOT_24, shift * (v % (1 << levels)), P)), M(tb["nibpow"]))
            p.jump(K(1), K(Label(f"merkle_table_{r}_{k}@after")), FP(0))
            assert p.here() - start == levels + 2
    return g.top


def emit_s6s_loop(p, S, r, desc_abs):
    """whir.py:143-147 for one (child, query) pair of round r: eq(expand_from_univariate_base(circle value), my_folding_randomness).
    frame: [return pc, saved fp, i, end, circle_all, s6s_all, locals...]"""
    N_ARGS = 4
    I, CIRCLE, S6S = 2, 4, 5
    g = Fn(p, 2 + N_ARGS, ZERO_VEC_PTR, ONE_EF_PTR, REPEATED_ONES_PTR)
    label, size_label = f"s6s_loop_{r}", f"@s6s_frame_{r}"
    loop_prologue(p, g, label, N_ARGS)
    DESC = g.alloc()
    after_child, child_table, _ = emit_child_table(p, g, S, r, f"child_table_s{r}", [(DESC, desc_abs, 0)])
    rand_ptr, cptr = g.alloc(), g.alloc()
    p.deref(DESC, 3, M(rand_ptr))
    p.add(M(CIRCLE), M(I), M(cptr))
    nr = S.n_rem[r]
    ex = g.alloc(nr)
    p.deref(cptr, 0, M(ex))                                       # expand_from_univariate_base_const (utils.py:142-150)
    for i in range(1, nr):
        p.mul(M(ex + i - 1), M(ex + i - 1), M(ex + i))
    ex_ptr, i5, dst = g.alloc(), g.alloc(), g.alloc()
    p.add(K(0), FP(ex), M(ex_ptr))
    p.mul(M(I), K(DIM), M(i5))
    p.add(M(S6S), M(i5), M(dst))
    p.extension_op("poly_eq", M(ex_ptr), M(rand_ptr), M(dst), size=nr, is_be=True)
    loop_epilogue(p, g, label, size_label, N_ARGS)
    p.labels[size_label] = g.top
    p.label(f"child_table_s{r}")
    for entry in child_table:
        for cell, value in entry:
            p.add(K(0), K(value), M(cell))
        p.jump(K(1), K(Label(after_child)), FP(0))
    return g.top


# ================================================================================================================================================
# the witness of one run
# ================================================================================================================================================
def parse_raw_proof(words):
    """lmh_proof_copy blob -> (transcript, [(index, leaf, path)])  (include/leanmultisig_host.h: RawProof layout)"""
    w = np.asarray(words, dtype=np.uint32)
    t = int(w[0])
    transcript = w[1:1 + t]
    o = 1 + t
    m = int(w[o])
    o += 1
    openings = []
    for _ in range(m):
        idx = int(w[o]) | (int(w[o + 1]) << 32)
        ll, pl = int(w[o + 2]), int(w[o + 3])
        o += 4
        openings.append((idx, w[o:o + ll], w[o + ll:o + ll + pl]))
        o += ll + pl
    assert o == w.size
    return transcript, openings


def claim_words(S, claim, stmt=None, public_input=None):
    """one child's block of the claims buffer from an lm_whir_opening_claim (capi.WhirOpeningClaim) [+ lm_pcs_statement_claim]"""
    o0 = S.oods[0]
    assert claim.num_variables == S.n and claim.log_inv_rate == S.rate and claim.n_ood == o0
    out = np.zeros(S.claim_words, dtype=np.uint32)
    out[S.c_fs:S.c_fs + 16] = np.ctypeslib.as_array(claim.challenger_state)
    out[S.c_root:S.c_root + 8] = np.ctypeslib.as_array(claim.root)
    out[S.c_ood_points:S.c_ood_points + DIM * o0] = np.ctypeslib.as_array(claim.ood_points)[:DIM * o0]
    out[S.c_ood_evals:S.c_ood_evals + DIM * o0] = np.ctypeslib.as_array(claim.ood_answers)[:DIM * o0]
    if S.statement is None:
        out[S.c_stmt_sum:S.c_stmt_sum + DIM] = np.ctypeslib.as_array(claim.statement_sum)
        out[S.c_stmt_weights:S.c_stmt_weights + DIM] = np.ctypeslib.as_array(claim.statement_weights)
    else:
        T = S.statement
        assert Statement(stmt, claim).key() == T.key(), "a child proof of another shape than the program was assembled for"
        pi = np.asarray(public_input, dtype=np.uint32)
        out[S.c_public_input:S.c_public_input + pi.size] = pi                   # (zero padded to the public memory's power of two)
        if S.head:
            out[:] = 0
            out[S.c_public_input:S.c_public_input + pi.size] = pi
            out[S.c_domsep:S.c_domsep + 8] = np.ctypeslib.as_array(stmt.bytecode_hash_domsep)
            out[S.c_air_evals:S.c_air_evals + 3 * DIM] = np.ctypeslib.as_array(stmt.air_constraint_evals).reshape(-1)
            out[S.c_bytecode_value:S.c_bytecode_value + DIM] = np.ctypeslib.as_array(stmt.bytecode_value)
            out[S.c_rand:S.c_rand + DIM * S.n] = np.ctypeslib.as_array(claim.folding_randomness)[:DIM * S.n]
            return out
        out[S.c_gkr_point:S.c_gkr_point + DIM * T.gkr_n_vars] = np.ctypeslib.as_array(stmt.gkr_point)[:DIM * T.gkr_n_vars]
        if S.air:
            out[S.c_fs:S.c_fs + 16] = np.ctypeslib.as_array(stmt.air_challenger_state)
            out[S.c_logup_c:S.c_logup_c + DIM] = np.ctypeslib.as_array(stmt.logup_c)
            out[S.c_air_evals:S.c_air_evals + 3 * DIM] = np.ctypeslib.as_array(stmt.air_constraint_evals).reshape(-1)
        else:
            ap = np.ctypeslib.as_array(stmt.air_point)[:DIM * T.n_max].reshape(T.n_max, DIM)
            out[S.c_air_point:S.c_air_point + DIM * T.n_max] = ap[::-1].reshape(-1)
            out[S.c_pm_point:S.c_pm_point + DIM * T.lpm] = np.ctypeslib.as_array(stmt.pm_point)[:DIM * T.lpm]
    out[S.c_rand:S.c_rand + DIM * S.n] = np.ctypeslib.as_array(claim.folding_randomness)[:DIM * S.n]
    return out


def build_witness(bc, children):
    """children: per child (raw transcript words, lm_whir_opening_claim, openings [(index, leaf, path)] of the WHOLE proof in opening
    order — every opening of a proof belongs to its PCS opening [, lm_pcs_statement_claim, the child's public input: programs assembled
    with a Statement]).  -> (public_input, vm.Witness, info).
    The hint streams are what type_1_aggregation.rs:310-356 builds for a recursion: `proof_transcript` per child, and the
    `merkle_leaf` / `merkle_path` blobs of extract_merkle_hint_blobs in the order the program's segments consume them: round by
    round, within a round child by child."""
    from ..xmss import Xmss
    from .xmss_aggregate import compress_slice
    S = bc.info["shape"]
    assert len(children) == S.n_children
    claims, transcripts, per_child = [], [], []
    for child in children:
        raw, claim, openings = child[:3]
        off = int(claim.transcript_offset) if S.statement is None else 0
        t = np.asarray(raw, dtype=np.uint32)[off:]
        assert t.size == S.transcript_words, (t.size, S.transcript_words)
        transcripts.append(t)
        claims.append(claim_words(S, claim, *child[3:5]))
        assert len(openings) == sum(S.queries)
        per_child.append(openings)
    leaves, paths = [], []
    for r in range(S.n_rounds + 1):
        o0 = sum(S.queries[:r])
        for c in range(S.n_children):
            for idx, leaf, path in per_child[c][o0:o0 + S.queries[r]]:
                assert leaf.size == S.leaf_words[r] and path.size == 8 * S.height[r]
                leaves.append(leaf)
                paths.append(path)
    data = np.concatenate(claims)
    public_input = compress_slice(Xmss(), data, use_iv=True)
    hints = {"claims": [data], "proof_transcript": transcripts, "merkle_leaf": leaves, "merkle_path": paths}
    return public_input, Witness(bc, PREAMBLE_MEMORY_LEN, hints), dict(claims=data)


def expected_counts(S):
    """Poseidon16 calls and ExtensionOp rows of one run, from the protocol parameters alone (the terms tools/recursion_shape.py counts
    for the reference's program, here for this lowering): what a run must report (tests/test_whir_verify_program.py)."""
    n_sumcheck = S.n
    grind = lambda b: 1 if b else 0  # noqa: E731
    pos = ext = 0
    # Fiat-Shamir: one permutation per absorbed or squeezed rate block (fiat_shamir.py)
    for r in range(S.n_rounds + 1):
        pos += S.fold[r] * (2 + grind(S.folding_grinding[r]))
        q = S.queries[r]
        if r < S.n_rounds:
            o = S.oods[r + 1]
            pos += 1 + (ceil_div(DIM * o, 8) - 1) + ceil_div(DIM * o, 8) + 1      # root, OOD points, OOD answers, the duplex before gamma
        else:
            pos += ceil_div(DIM << S.n_final, 8)
        pos += grind(S.query_grinding[r]) + ceil_div(q, 8) - 1
        pos += q * (S.leaf_words[r] // 8 - 1 + S.height[r])                       # leaf sponge + path
    pos += 2 * S.n_final
    pos *= S.n_children
    pos += S.n_children * S.claim_words // 8                                      # the public input
    # ExtensionOp rows
    copy8 = 2
    powers = lambda k: 1 if k == 1 else k                                         # noqa: E731 — set_one, copy, k - 2 products
    eq_mle = lambda k: 1 + k + 2 * ((1 << k) - 1)                                 # noqa: E731
    ext += n_sumcheck * (3 + 1 + 1 + powers(3) + 3)
    o0 = S.oods[0]
    ext += powers(o0) + o0 + 1
    for r in range(S.n_rounds + 1):
        q = S.queries[r]
        nq = ceil_div(q, 8)
        ext += (copy8 * nq if nq > 1 else 0) + copy8 * nq + eq_mle(S.fold[r])
        if r < S.n_rounds:
            o, nr = S.oods[r + 1], S.n_rem[r]
            no = ceil_div(DIM * o, 8)
            ext += copy8 * no if no > 1 else 0
            ext += powers(1 << (q + o - 1).bit_length()) + o + q + 2
            ext += q * (1 << S.fold[r])                                           # leaf folds
            ext += o * 2 * nr + o + q + 2 + q * nr                                # OOD / STIR constraint weights
        else:
            ext += q * ((1 << S.fold[r]) + (1 << S.n_final))
    ext += o0 * 2 * S.n + o0
    ext += 1 + S.n_final + (1 << S.n_final) - 1 + (1 << S.n_final) + 2 + S.n
    ext *= S.n_children
    return dict(poseidon_calls=pos, extension_rows=ext)
