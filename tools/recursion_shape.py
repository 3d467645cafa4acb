#!/usr/bin/env python3
"""BASELINE configs[3] (`recursion --n 4 --log-inv-rate 2`): table shapes of the ROOT step, derived by counting.

The root of the reference's recursion benchmark (src/main.rs:91-115) aggregates 4 children of 775 raw signatures each: its
program is the in-VM verifier of `verify_execution` (crates/rec_aggregation/zkdsl_implem/{recursion,whir,fiat_shamir,hashing}.py)
run once per child.  The zkDSL compiler is out of scope here, so the program cannot be executed; what CAN be done is
to count what it must execute, from the protocol parameters the library derives itself (WhirConfig::new) and the loop structure of
the zkDSL sources.  Every term below cites the lines it counts.  The result replaces the survey's guessed stand-in shapes
(execution 2^21, ExtensionOp 2^19, Poseidon 2^17, memory 2^23): it is still a stand-in — the register allocation, inlining and loop
overheads of the real compiled program are unknown — but a derived one.

    python tools/recursion_shape.py            # prints the derivation as JSON
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_CHILDREN, CHILD_SIGS, LOG_INV_RATE = 4, 775, 2
# per-signature footprint of the aggregation program (leanmultisig_amd/programs/xmss_aggregate.py, measured by the runner:
# 838 738 cycles / 257 134 Poseidon calls / 30 620 ExtensionOp rows for the 1549 segments of bench.py's default workload)
CYCLES_PER_SIG, POSEIDON_PER_SIG, EXT_ROWS_PER_SIG, FRAME_WORDS_PER_SIG = 542, 166, 20, 2241


def log2_ceil(x):
    return max(0, math.ceil(math.log2(max(1, x))))


def child_shape():
    """a leaf of 775 raw signatures at rate 1/4 (the child whose proof the root verifies)"""
    cycles = CHILD_SIGS * CYCLES_PER_SIG + 7000
    poseidon = CHILD_SIGS * POSEIDON_PER_SIG + 1000
    ext = CHILD_SIGS * EXT_ROWS_PER_SIG + 200
    log_exec, log_pos, log_ext = log2_ceil(cycles), log2_ceil(poseidon), max(8, log2_ceil(ext))
    log_mem = max(16, log2_ceil(CHILD_SIGS * FRAME_WORDS_PER_SIG + 60000), log_exec)
    log_bc = 19
    # stack_polynomials (sub_protocols/src/stacked_pcs.rs:118-136)
    total = 2 * (1 << log_mem) + max(1 << log_bc, 1 << log_exec) + 20 * (1 << log_exec) + 29 * (1 << log_ext) + 109 * (1 << log_pos)
    # logup domain (logup.rs:88-199): memory + bytecode sections + per table: bus + one entry per looked-up value column
    logup = (1 << log_mem) + max(1 << log_bc, 1 << log_exec) + (1 + 1 + 3) * (1 << log_exec) + (1 + 15) * (1 << log_ext) + (1 + 32) * (1 << log_pos)
    return dict(log_exec=log_exec, log_pos=log_pos, log_ext=log_ext, log_memory=log_mem, log_bytecode=log_bc, stacked_n_vars=log2_ceil(total),
                gkr_n_vars=log2_ceil(logup))


def derive():
    import leanmultisig_amd as lm
    ch = child_shape()
    n = ch["stacked_n_vars"]
    cfg = lm.WhirConfig.new(lm.WhirBuilder.default(LOG_INV_RATE), n).to_dict()
    fold = [cfg["fold_first"]] + [cfg["fold_sub"]] * cfg["n_rounds"]
    queries = [r["num_queries"] for r in cfg["rounds"]] + [cfg["final_queries"]]
    oods = [cfg["commitment_ood_samples"]] + [r["ood_samples"] for r in cfg["rounds"]]
    n_final = cfg["final_sumcheck_rounds"]
    # Merkle trees the queries open (whir.py:266-313): round r commits 2^(domain - fold) leaves of 2^fold values
    domain = n + LOG_INV_RATE
    trees, n_rem = [], n
    for r, q in enumerate(queries):
        height = domain - fold[r]
        words = (1 << fold[r]) * (1 if r == 0 else 5)
        n_rem -= fold[r]
        trees.append(dict(round=r, queries=q, height=height, leaf_chunks=words // 8, fold=fold[r], base_field_leaves=r == 0, n_vars_remaining=n_rem))
        domain -= cfg["rs_red"] if r == 0 else 1
    terms = {}
    # ---- Poseidon16 calls per child -------------------------------------------------------------------------------------------------
    # decompose_and_verify_merkle_query (hashing.py): leaf sponge (one compression per 8-word chunk) + one compression per level
    terms["poseidon.merkle"] = sum(t["queries"] * (t["height"] + t["leaf_chunks"]) for t in trees)
    # fiat_shamir.py: one permutation per absorbed / squeezed rate block.  Absorbed words of verify_execution: header, roots, OOD
    # answers, GKR (4 coefficients per round + 4 inner evaluations per layer, 5 words each), logup column evaluations (~170), AIR
    # (12 coefficients per round + ~160 column evaluations), WHIR sumcheck polynomials, final coefficients; one squeeze per challenge
    gkr_rounds = sum(range(5, ch["gkr_n_vars"]))
    absorbed_ef = 4 * gkr_rounds + 4 * (ch["gkr_n_vars"] - 5) + 64 + 170 + 12 * ch["log_exec"] + 160 + 3 * (n + n_final) + (1 << n_final) + sum(oods)
    challenges = gkr_rounds + 2 * (ch["gkr_n_vars"] - 5) + ch["log_exec"] + n + n_final + 3 * len(trees) + 40
    terms["poseidon.fiat_shamir"] = (5 * absorbed_ef + 7) // 8 + challenges + sum(t["queries"] for t in trees) // 8 * 2
    # proof-of-work checks: one permutation each (fiat_shamir.py)
    terms["poseidon.pow"] = n + 2 * len(trees) + 2
    # ---- ExtensionOp rows per child (one row per element of a dot product / poly_eq chain, extension_op/exec.rs:106-190) -----------------
    # whir.py:304-311: every opened leaf is folded with the eq table of the folding randomness (base x ext for the first tree)
    terms["ext.leaf_folds"] = sum(t["queries"] * (1 << t["fold"]) for t in trees)
    terms["ext.eq_tables"] = sum(2 << t["fold"] for t in trees)                       # compute_eq_mle_extension_dynamic
    # whir.py:145-147: per STIR query of round r the eq polynomial of its point against the remaining folding randomness
    terms["ext.query_eq"] = sum(t["queries"] * t["n_vars_remaining"] for t in trees[:-1])
    # whir.py:92-104: the final polynomial (2^n_final coefficients) evaluated at every final query point
    terms["ext.final_poly_evals"] = trees[-1]["queries"] * (1 << n_final)
    # whir.py:114-136: OOD points: expand_from_univariate_ext (one product per variable) + poly_eq_extension
    terms["ext.ood"] = 2 * (oods[0] * n + sum(o * t["n_vars_remaining"] for o, t in zip(oods[1:], trees)))
    terms["ext.combination_dots"] = sum(t["queries"] for t in trees) + sum(oods)      # whir.py:118-154, 351-354
    terms["ext.sumcheck_verify"] = 6 * (n + n_final)                                  # whir.py:167-222: p(0) + p(1), p(r) per round
    # recursion.py:684-750: per GKR layer i: an i-round degree-3 sumcheck (eval at the challenge: 4 products), the eq factor
    # (poly_eq over i coordinates) and ~12 single products
    terms["ext.gkr_verify"] = sum(6 * i + i + 12 for i in range(5, ch["gkr_n_vars"]))
    # recursion.py:416-445: batched AIR sumcheck (degree 12: 13 products per round) + per table eq factor + the constraint
    # polynomials at the point: Poseidon16 (8 full rounds: 16 cubes + a 16 x 16 base-by-extension matrix; 20 partial rounds: a cube + 31
    # products; flags, bus), ExtensionOp (~60 products), execution (~40)
    terms["ext.air_verify"] = 14 * ch["log_exec"] + ch["log_exec"] + ch["log_ext"] + ch["log_pos"] + 8 * (32 + 256) + 20 * 33 + 100 + 60 + 40
    # recursion.py:465-650: statement assembly: ~160 column claims x 3 products, eq / next factors per table, the public-memory eq
    # table (2^INNER_PUBLIC_MEMORY_LOG_SIZE = 8 words) and location prefixes
    terms["ext.statements"] = 160 * 3 + 3 * (2 * ch["log_exec"] + 20) + 16 + 200
    poseidon = sum(v for k, v in terms.items() if k.startswith("poseidon."))
    ext_rows = sum(v for k, v in terms.items() if k.startswith("ext."))
    # ---- cycles per child: the program's own instructions around the precompile calls.  Per Merkle level a bit test, a branch and two
    # pointer updates (~8 cycles, cf. do_4_merkle_levels in xmss_aggregate.py); per Poseidon / ExtensionOp CALL ~3 cycles of operand
    # set-up; per query ~40 cycles (index decomposition into `height` bits with range checks, expand_from_univariate_base:
    # n_vars_remaining squarings as MUL instructions); ~25 k cycles of straight-line code (statement assembly is unrolled).
    ext_calls = (sum(t["queries"] for t in trees) * 2 + sum(oods) * 2 + 4 * (n + n_final) + 20 * ch["gkr_n_vars"] + 1200)
    merkle_levels = sum(t["queries"] * t["height"] for t in trees)
    # Round 5: the WHIR part of the verifier (whir_open) is assembled and EXECUTED (leanmultisig_amd/programs/whir_verify.py on four
    # genuine 775-signature child proofs, profiles/r05_bench_whir_recursion.json): 43 285 cycles, 7 412 Poseidon16 calls and 29 931
    # ExtensionOp rows per child.  Round 4's rules for that part (8 cycles per Merkle level, 3 per precompile call, 40 + 3 x height per
    # query) gave 76.4 k cycles — 77 % too many: a level is ONE cycle (the Poseidon instruction; the branch is a 16-entry jump table per
    # four levels, 7 cycles per nibble), a query 65 + 7 x nibbles + leaf chunks.  Calibrated rules for the WHIR part; the rest of the
    # verifier (GKR, logup, AIR, statements: straight-line extension-field code) keeps 3 cycles per precompile call + 25 k.
    whir_pos = terms["poseidon.merkle"] + terms["poseidon.pow"] + 60
    whir_ext_calls = sum(t["queries"] for t in trees) * 2 + sum(oods) * 2 + 4 * (n + n_final)
    whir_cycles_round4_rules = (3 * whir_pos + 3 * whir_ext_calls + 8 * merkle_levels
                                + sum(t["queries"] * (40 + 3 * t["height"] + t["n_vars_remaining"]) for t in trees))
    whir_cycles = (sum(t["queries"] * (65 + 7 * -(-t["height"] // 4) + t["leaf_chunks"]) for t in trees) + trees[-1]["queries"] * (1 << n_final)
                   + sum(t["queries"] * (22 + t["n_vars_remaining"]) for t in trees[:-1]) + 6500)
    cycles = whir_cycles + 3 * (poseidon - whir_pos) + 3 * (ext_calls - whir_ext_calls) + 25000
    # ---- memory per child: the hinted Merkle openings (leaf + 8 words per level) and the verifier's frames (~4 words per cycle)
    hint_words = sum(t["queries"] * (8 * t["leaf_chunks"] + 8 * t["height"]) for t in trees) + 5 * absorbed_ef
    memory_words = hint_words + 4 * cycles
    per_child = dict(poseidon_calls=poseidon, extension_rows=ext_rows, extension_calls=ext_calls, cycles=cycles, memory_words=memory_words, terms=terms,
                     trees=trees, whir_cycles=whir_cycles, whir_cycles_round4_rules=whir_cycles_round4_rules,
                     measured_whir_part=dict(cycles=43285, poseidon_calls=7412, extension_rows=29931, source="whir_open alone, round 5"),
                     # recursion() whole as assembled by hand (programs/whir_verify.py + air_eval.py), per child of the same shape: the
                     # counted ExtensionOp rows were 25 % low for THIS lowering (textbook MDS layers as 16-term dot products in the AIR
                     # evaluator: 8.8 k rows where the count assumed 3.5 k), the counted cycles 36 % high
                     measured_whole_recursion=dict(cycles=58148, poseidon_calls=8681, extension_rows=47215, source="profiles/r05_bench_whir_recursion.json"))
    root = dict(poseidon_calls=N_CHILDREN * poseidon + 400, extension_rows=N_CHILDREN * ext_rows + 2000, cycles=N_CHILDREN * cycles + 10000,
                memory_words=N_CHILDREN * memory_words + 50000)
    shape = dict(log_exec=log2_ceil(root["cycles"]), log_pos=max(8, log2_ceil(root["poseidon_calls"])), log_ext=max(8, log2_ceil(root["extension_rows"])),
                 log_memory=max(16, log2_ceil(root["memory_words"])), log_bytecode=19)
    shape["log_memory"] = max(shape["log_memory"], shape["log_exec"])
    # the mix of ExtensionOp calls a synthetic witness should carry (op, base-by-extension, length, count), ALL children together
    t0, rest = trees[0], trees[1:]
    mix = [("mul", True, 1 << t0["fold"], N_CHILDREN * t0["queries"])]                                            # first-tree leaf folds
    mix += [("mul", False, 1 << t["fold"], N_CHILDREN * t["queries"]) for t in rest]                              # extension-field leaf folds
    mix += [("poly_eq", True, t["n_vars_remaining"], N_CHILDREN * t["queries"]) for t in trees[:-1]]              # query eq factors
    mix += [("mul", True, 1 << n_final, N_CHILDREN * trees[-1]["queries"])]                                       # final polynomial at the final queries
    mix += [("poly_eq", False, i, N_CHILDREN) for i in range(5, ch["gkr_n_vars"])]                                # GKR eq factors
    used = sum(s * c for _, _, s, c in mix)
    singles = max(0, root["extension_rows"] - used)
    mix += [("mul", False, 1, singles // 2), ("add", False, 1, singles - singles // 2)]                           # single products / sums
    return dict(child=ch, whir=cfg, per_child=per_child, root=root, shape=shape, ext_calls=mix,
                note="derived by counting (tools/recursion_shape.py); the PCS-opening part is executed since round 5 and its cycle rules are calibrated on "
                     "that run; the compiled reference verifier's own cycle count is unknown")


if __name__ == "__main__":
    print(json.dumps(derive(), indent=1))
