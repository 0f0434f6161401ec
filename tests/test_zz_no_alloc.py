"""GPU: the allocation-free ratio test (lpamd_ratio_test_noalloc) against lpamd_ratio_test.

Two calls in a row on the same GPU, the second with a smaller input: the library's per-GPU
partial-result array keeps the first call's contents, which the second call must not see.
Bit-exact bar: the pivot row equals the engine's, the ratio equals rhs[r] / (-col[r])."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(lpamd, eng_mod, m, n, seed, warm_pivots):
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    if warm_pivots:
        assert eng.run_pivots(lpamd.MAX_COST, warm_pivots) == (lpamd.LIMIT, warm_pivots)
    e = eng.price_max_cost()
    assert e >= 0
    r_ref, _ = eng.ratio_test(e)
    col, rhs = eng.read_column(e), eng.read_column(n + m)
    eng.close()
    return col, rhs, r_ref


def _noalloc(eng_mod, col, rhs, stream=None):
    dev = torch.device("cuda", 0)
    c = torch.from_numpy(col).to(dev)
    b = torch.from_numpy(rhs).to(dev)
    row = torch.full((1,), -7, dtype=torch.int64, device=dev)
    ratio = torch.full((1,), -1.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    eng_mod.ratio_test_noalloc(c.data_ptr(), b.data_ptr(), len(col), row.data_ptr(), ratio.data_ptr(), stream)
    torch.cuda.synchronize()
    return int(row.item()), float(ratio.item())


def _numpy_case(m, seed):
    """Random vectors and the rule of reference lp_simplex_tableau.cc:207-227 in numpy: rows whose entry
    narrows to a float <= -1e-6f compete with b / (-a); the first strict minimum wins."""
    rng = np.random.default_rng(seed)
    col = rng.uniform(-1.0, 0.5, m)
    col[rng.integers(0, m, m // 50)] = 0.0
    rhs = rng.uniform(0.5, 2.0, m)
    elig = col.astype(np.float32) <= np.float32(-1e-6)
    ratios = np.where(elig, rhs / np.where(elig, -col, 1.0), np.inf)
    return col, rhs, int(np.argmin(ratios))


def test_noalloc_ratio_test_two_calls_second_smaller(lpamd):
    eng_mod = importlib.import_module("linear-programming_amd.engine")
    torch.cuda.set_device(0)
    big = _numpy_case(300000, 5)                        # 1024 partials, every workgroup strides over the rows
    mid = _case(lpamd, eng_mod, 5000, 700, 3, 9)
    small = _case(lpamd, eng_mod, 300, 500, 4, 5)       # 2 partials: the rest of the array is stale
    for col, rhs, r_ref in (big, mid, small, mid):
        r, ratio = _noalloc(eng_mod, col, rhs)
        assert r == r_ref >= 0
        assert ratio == rhs[r] / (-col[r])


def test_noalloc_ratio_test_edge_cases(lpamd):
    eng_mod = importlib.import_module("linear-programming_amd.engine")
    torch.cuda.set_device(0)
    rhs = np.array([4.0, 2.0, 2.0, 9.0, 1.0])
    # ties go to the lowest row; entries the float-narrowed predicate does not call negative are skipped
    col = np.array([-2.0, -1.0, -1.0, 1.0, -1e-7])
    assert _noalloc(eng_mod, col, rhs) == (0, 2.0)
    col = np.array([-1.0, -1.0, -1.0, 1.0, 0.0])
    assert _noalloc(eng_mod, col, rhs) == (1, 2.0)
    # no eligible row: unbounded
    r, ratio = _noalloc(eng_mod, np.array([1.0, 0.0, 5e-7, 2.0, 0.0]), rhs)
    assert (r, ratio) == (-1, 0.0)
    # the reductions run on an integer image of (ratio, row): negative ratios, both zeros (equal: the lower row wins),
    # denormals and ties far apart (different wavefronts and workgroups) must order as the doubles do
    assert _noalloc(eng_mod, np.array([-1.0, -1.0, -1.0, -1.0]), np.array([1.0, -2.0, -3.0, -3.0])) == (2, -3.0)
    assert _noalloc(eng_mod, np.array([-1.0, -1.0, -1.0]), np.array([0.0, -0.0, 3.0])) == (0, 0.0)
    assert _noalloc(eng_mod, np.array([-1.0, -1.0, -1.0]), np.array([-0.0, 0.0, 3.0])) == (0, 0.0)
    assert _noalloc(eng_mod, np.array([-1.0, -1.0]), np.array([1e-310, 5e-324])) == (1, 5e-324)
    m = 70000
    col, rhs = np.full(m, -1.0), np.full(m, 7.0)
    rhs[[65, 300, 66000]] = 2.5                      # equal minima in three wavefronts of two workgroups
    assert _noalloc(eng_mod, col, rhs) == (65, 2.5)
    rhs[65] = 7.0
    assert _noalloc(eng_mod, col, rhs) == (300, 2.5)
    rhs[300] = np.nextafter(2.5, 3.0)
    assert _noalloc(eng_mod, col, rhs) == (66000, 2.5)
    # one row, on an explicit stream
    s = torch.cuda.Stream()
    assert _noalloc(eng_mod, np.array([-4.0]), np.array([2.0]), s.cuda_stream) == (0, 0.5)
    with pytest.raises(lpamd.LpamdError, match="outside"):
        eng_mod.ratio_test_noalloc(1, 1, 0, 1)
    with pytest.raises(lpamd.LpamdError, match="outside"):
        eng_mod.ratio_test_noalloc(1, 1, (1 << 24) + 1, 1)
