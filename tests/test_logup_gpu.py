"""GPU parity: logup numerator / denominator construction (prove_generic_logup fill loops) vs the CPU oracle."""
import numpy as np
import pytest

from tests import oracle_binding as ob
from tests.oracle_binding import rand_field

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_mem,log_bc,heights", [(10, 8, {0: 9, 1: 8, 2: 8}), (11, 10, {0: 9, 2: 9, 1: 8}), (9, 9, {0: 8, 1: 8, 2: 8})])
def test_logup_build_matches_oracle(ctx, orc, log_mem, log_bc, heights):
    rng = np.random.default_rng(log_mem * 31 + log_bc)
    memory, memory_acc = rand_field(rng, 1 << log_mem), rand_field(rng, 1 << log_mem)
    bytecode, bytecode_acc = rand_field(rng, 16 << log_bc), rand_field(rng, 1 << log_bc)
    order = sorted(heights, key=lambda t: -heights[t])  # sort_tables_by_height: descending, stable in enum order
    tables = [(t, rand_field(rng, (ob.VM_N_TOTAL[t], 1 << heights[t]))) for t in order]
    c, alphas = rand_field(rng, 5), rand_field(rng, (16, 5))
    total, nums, dens = ob.logup_fill(orc, memory, memory_acc, bytecode, bytecode_acc, tables, c, alphas)
    n_vars = int(nums.size).bit_length() - 1
    d_mem, d_acc = ctx.to_device(memory), ctx.to_device(memory_acc)
    d_bc, d_bca = ctx.to_device(bytecode), ctx.to_device(bytecode_acc)
    keep, d_tables = [], []
    for t, cols in tables:
        bufs = [ctx.to_device(c_) for c_ in cols]
        keep.append(bufs)
        d_tables.append((t, int(cols.shape[1]).bit_length() - 1, [b.ptr for b in bufs]))
    secs, off = ob.logup_sections(d_mem.ptr, d_acc.ptr, log_mem, d_bc.ptr, d_bca.ptr, log_bc, d_tables)
    assert off == total
    d_nums, d_dens = ctx.logup_build(secs, c, alphas, n_vars)
    assert np.array_equal(d_nums.download(), nums)
    assert np.array_equal(d_dens.download().reshape(5, -1).T, dens)
