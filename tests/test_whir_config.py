"""lmh_whir_config_new — the product's own WhirConfig::new (crates/whir/src/config.rs:186-334) — against the oracle's
restatement over a grid of sizes, rates, security levels and all three soundness assumptions, and against the external
sanity points of SURVEY.md §8 (query counts of the headline configurations).  CPU only: no device call."""
import numpy as np
import pytest

import leanmultisig_amd as lm
from tests import oracle_binding as ob

KEYS = ["commitment_ood_samples", "starting_folding_pow_bits", "n_rounds", "final_queries", "final_query_pow_bits",
        "final_sumcheck_rounds"]


def _same(orc, nv, **kw):
    names = dict(log_inv_rate="starting_log_inv_rate", max_send="max_num_variables_to_send_coeffs",
                 rs_red="rs_domain_initial_reduction_factor", fold_first="folding_factor_first",
                 fold_sub="folding_factor_subsequent", soundness="soundness_type", security="security_level", pow_bits="pow_bits")
    ref = ob.whir_config(orc, ob.whir_builder(**kw), nv)
    b = lm.WhirBuilder.default(kw.get("log_inv_rate", 1), **{names[k]: v for k, v in kw.items() if k != "log_inv_rate"})
    got = lm.WhirConfig.new(b, nv).to_dict()
    for k in KEYS:
        assert got[k] == ref[k], (nv, kw, k, got[k], ref[k])
    assert len(got["rounds"]) == len(ref["rounds"])
    for g, r in zip(got["rounds"], ref["rounds"]):
        for k in ("query_pow_bits", "folding_pow_bits", "num_queries", "ood_samples"):
            assert g[k] == r[k], (nv, kw, k)
    return got


def test_config_matches_oracle_on_a_grid(orc):
    for nv in range(12, 31):
        for rate in (1, 2, 3, 4):
            if nv + rate - 7 > 24:
                continue
            for soundness in (0, 1, 2):
                _same(orc, nv, log_inv_rate=rate, soundness=soundness)
    for nv in (16, 22, 26):
        for security, pow_bits in ((60, 6), (80, 10), (100, 20), (124, 0), (128, 18)):
            _same(orc, nv, log_inv_rate=1, security=security, pow_bits=pow_bits)
        for ff, fs, red in ((4, 4, 1), (6, 3, 2), (7, 5, 7), (5, 2, 3)):
            _same(orc, nv, log_inv_rate=2, fold_first=ff, fold_sub=fs, rs_red=red)


def test_headline_schedules():
    """SURVEY.md §8 size table: rate 1/2 -> queries 243/74/32/21, rate 1/4 -> 118/56/28/19 at 26 variables."""
    c = lm.WhirConfig.new(lm.WhirBuilder.default(1), 26).to_dict()
    assert [r["num_queries"] for r in c["rounds"]] + [c["final_queries"]] == [243, 74, 32, 21]
    assert c["commitment_ood_samples"] == 2 and c["final_sumcheck_rounds"] == 4 and c["n_rounds"] == 3
    c = lm.WhirConfig.new(lm.WhirBuilder.default(2), 26).to_dict()
    assert [r["num_queries"] for r in c["rounds"]] + [c["final_queries"]] == [118, 56, 28, 19]


def test_invalid_builders_are_rejected():
    with pytest.raises(lm.LmError):
        lm.WhirConfig.new(lm.WhirBuilder.default(1), 5)          # folding factor 7 > 5 variables
    with pytest.raises(lm.LmError):
        lm.WhirConfig.new(lm.WhirBuilder.default(4), 30)         # folded domain beyond the two-adicity
    with pytest.raises(lm.LmError):
        lm.WhirConfig.new(lm.WhirBuilder.default(1, rs_domain_initial_reduction_factor=8), 26)
    with pytest.raises(lm.LmError):
        lm.WhirConfig.new(lm.WhirBuilder.default(1, security_level=160), 26)


def test_table_log_rows_matches_pad_table_rule():
    """lmh_table_log_rows = log2_ceil(n + 1).max(MIN_LOG_N_ROWS_PER_TABLE = 8) (pad_table, trace_gen.rs:185): a table always
    keeps at least one padding row."""
    import leanmultisig_amd as lm
    lib = lm.load()
    for n, want in ((0, 8), (1, 8), (255, 8), (256, 9), (257, 9), (511, 9), (512, 10), ((1 << 20) - 1, 20), (1 << 20, 21), (799392, 20)):
        assert lib.lmh_table_log_rows(n) == want, n
