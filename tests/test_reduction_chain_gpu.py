"""GPU parity of MAX_REDUCTION pricing from bounds (lpamd_prbound_kernels.hip.h), which lets those pivots
share a pass over the tableau like the MAX_COST ones.  The reference scores every candidate column on the
whole tableau (reference lp_simplex_tableau.cc:235-374); the engine reaches the same (column, row) from a
few rows per column plus exact evaluations of the columns the bounds cannot rule out.  Forced on inputs the
oracle solves fully (LPAMD_CHAIN makes short columns chain too): trace, tableau, objective row and basis
flags must equal the oracle's bit for bit, goldens included."""
import numpy as np
import pytest

from cases import all_cases, case_golden, case_text
from oracle import lp_frontend as fe
from oracle import lp_oracle as lo
from test_engine_parity import _random_tied_lp, assert_same_tableau, bits

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[2, 5, 16, 32])
def depth(request, monkeypatch):
    monkeypatch.setenv("LPAMD_CHAIN", str(request.param))
    return request.param


@pytest.mark.parametrize("name", sorted({n for n, s in all_cases() if s == "max_reduction"}))
def test_goldens_max_reduction_chained(lpamd, name):
    tab, dt = lo.from_text(case_text(name))
    want = dt.solve(lo.MAX_REDUCTION)
    eng = lpamd.TableauEngine(tab.m, tab.N)
    assert eng.loop_depth(lpamd.MAX_REDUCTION) > 1
    eng.upload(tab.T, tab.obj, tab.is_basic)
    res, n = eng.solve(lpamd.MAX_REDUCTION)
    assert res == want == lpamd.SOLVED and n == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    x = eng.column_values()
    assert np.array_equal(bits(x), bits(dt.column_values()))
    out = fe.format_solution(tab, x, eng.objective_constant())
    assert out.rstrip("\n") == case_golden(name, "max_reduction").rstrip("\n")
    eng.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_lps_with_ties_max_reduction_chained(lpamd, seed):
    """Integer data: equal scores, equal ratios, degenerate rows, unbounded and infeasible outcomes."""
    rng = np.random.default_rng(100 + seed)
    text = _random_tied_lp(rng, int(rng.integers(3, 14)), int(rng.integers(2, 12)))
    tab, dt = lo.from_text(text)
    want = dt.solve(lo.MAX_REDUCTION)
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    res, n = eng.solve(lpamd.MAX_REDUCTION)
    assert res == want, text
    assert eng.trace() == dt.trace(), text
    assert_same_tableau(eng, dt)
    eng.close()


@pytest.mark.parametrize("m,n,seed", [(600, 1100, 7), (2000, 900, 11), (40, 2500, 3), (2500, 40, 2), (1, 5, 1), (3, 1, 1)])
def test_synthetic_max_reduction_chained_to_the_optimum(lpamd, m, n, seed):
    dt = lo.DenseTableau.synthetic(m, n, seed)
    want = dt.solve(lo.MAX_REDUCTION)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    res, npiv = eng.solve(lpamd.MAX_REDUCTION)
    assert res == want == lpamd.SOLVED and npiv == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


@pytest.mark.parametrize("cache,rows", [("0", 32), ("1", 3), ("0", 3), ("1", 9)])
def test_bound_rows_from_the_cache_or_from_the_tableau(lpamd, monkeypatch, depth, cache, rows):
    """The rows the bounds look at come from a row-major cache that keeps them across the pricings of a pass
    (slots reused, evicted and refilled when few rows are allowed) or straight from the tableau: same pivots."""
    if depth not in (5, 32):
        pytest.skip("two depths are enough")
    monkeypatch.setenv("LPAMD_PRB_CACHE", cache)
    monkeypatch.setenv("LPAMD_PRB_ROWS", str(rows))
    m, n, seed = 700, 1300, 5
    dt = lo.DenseTableau.synthetic(m, n, seed)
    want = dt.solve(lo.MAX_REDUCTION)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    res, npiv = eng.solve(lpamd.MAX_REDUCTION)
    assert res == want == lpamd.SOLVED and npiv == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


def test_max_reduction_budgets_end_inside_a_chain(lpamd, depth):
    m, n = 600, 1100
    dt = lo.DenseTableau.synthetic(m, n, 7)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, 7)
    total = 0
    for k in (1, 2, 37, 0, 1, 4, 3, 16, 17):
        assert eng.run_pivots(lpamd.MAX_REDUCTION, k) == (lpamd.LIMIT, k)
        total += k
        assert dt.solve(lo.MAX_REDUCTION, max_pivots=total) == lo.LIMIT
        assert eng.pivot_count() == total and eng.trace() == dt.trace()
        if k:
            assert eng.pass_stats()[0] == (k - 1) % depth + 1
    assert_same_tableau(eng, dt)
    assert eng.run_pivots(lpamd.MAX_COST, 7) == (lpamd.LIMIT, 7)          # the other strategy in between
    assert dt.solve(lo.MAX_COST, max_pivots=total + 7) == lo.LIMIT
    assert eng.solve(lpamd.MAX_REDUCTION)[0] == dt.solve(lo.MAX_REDUCTION) == lpamd.SOLVED
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


def test_max_reduction_bounds_equal_the_full_pricing_pass(lpamd, monkeypatch, depth):
    """LPAMD_PR_BOUNDS=0 keeps one pivot per pass with the pricing fused into the update; same bits."""
    if depth != 16:
        pytest.skip("one run is enough")
    m, n, k = 2304, 1500, 45
    monkeypatch.delenv("LPAMD_CHAIN")
    a = lpamd.TableauEngine(m, n + m)          # default knobs: long columns chain
    assert a.loop_depth(lpamd.MAX_REDUCTION) >= 8
    monkeypatch.setenv("LPAMD_PR_BOUNDS", "0")
    b = lpamd.TableauEngine(m, n + m)
    assert b.loop_depth(lpamd.MAX_REDUCTION) == 1
    for eng in (a, b):
        eng.fill_synthetic(n, 5)
        assert eng.run_pivots(lpamd.MAX_REDUCTION, k) == (lpamd.LIMIT, k)
    assert a.trace() == b.trace()
    ta, tb = a.download(), b.download()
    for x, y in zip(ta[:2], tb[:2]):
        assert np.array_equal(bits(x), bits(y))
    assert np.array_equal(ta[2], tb[2])
    dt = lo.DenseTableau.synthetic(m, n, 5)
    dt.s.skip_zero_cols = 1
    dt.s.fast_basis = 1
    assert dt.solve(lo.MAX_REDUCTION, max_pivots=k) == lo.LIMIT
    assert a.trace() == dt.trace()
    assert_same_tableau(a, dt)
    a.close()
    b.close()


def test_max_reduction_unbounded_and_infeasible(lpamd):
    for text, want in (("max x1\nst\nx1 <= 1\nx1 >= 2\nx1 >= 0\n", lo.NOSOLUTION),
                       ("max x1 + x2\nst\nx1 - x2 <= 1\nx1 >= 0\nx2 >= 0\n", lo.UNBOUNDED),
                       ("max x1 + 2 * x2 + x3\nst\nx1 - x2 + x3 <= 4\nx1 + x3 <= 3\nx1 >= 0\nx2 >= 0\nx3 >= 0\n", lo.UNBOUNDED)):
        tab, dt = lo.from_text(text)
        assert dt.solve(lo.MAX_REDUCTION) == want
        eng = lpamd.TableauEngine(tab.m, tab.N)
        eng.upload(tab.T, tab.obj, tab.is_basic)
        res, _ = eng.solve(lpamd.MAX_REDUCTION)
        assert res == want and eng.trace() == dt.trace()
        assert_same_tableau(eng, dt)
        eng.close()


def test_cache_slots_outlive_a_pass_only_while_nothing_else_writes_the_tableau(lpamd, monkeypatch, depth):
    """The rows the bounds look at stay cached from one pass to the next.  Whatever else writes the tableau in between
    (a column filled by hand, single pivots, passes of the other strategy, a new upload into the same engine) must make
    the next pricing gather them again: same pivots and bits as an engine that never keeps them (LPAMD_PRB_CARRY=0)."""
    if depth not in (5, 16):
        pytest.skip("two depths are enough")
    m, n, seed = 700, 1300, 9
    keep = lpamd.TableauEngine(m, n + m)
    monkeypatch.setenv("LPAMD_PRB_CARRY", "0")
    fresh = lpamd.TableauEngine(m, n + m)
    dt = lo.DenseTableau.synthetic(m, n, seed)
    total = 0

    def both(f):
        out = [f(e) for e in (keep, fresh)]
        assert out[0] == out[1]
        return out[0]

    both(lambda e: e.fill_synthetic(n, seed))
    for k in (2 * depth + 3, depth, 1):
        both(lambda e: e.run_pivots(lpamd.MAX_REDUCTION, k))
        total += k
    assert dt.solve(lo.MAX_REDUCTION, max_pivots=total) == lo.LIMIT
    assert keep.trace() == fresh.trace() == dt.trace()
    assert_same_tableau(keep, dt)
    # a non-basic column changed by hand: the pivot count stays, the cached rows are stale
    basic = keep.download_basic()
    col = next(j for j in range(n) if not basic[j])
    both(lambda e: e.fill_column(col, -0.75))
    both(lambda e: e.run_pivots(lpamd.MAX_REDUCTION, depth + 2))
    # a single pivot through the step-by-step entry points, then passes of the other strategy
    e_col, row, leaving = both(lambda e: e.price_max_reduction())
    if e_col >= 0 and row >= 0:
        both(lambda e: e.pivot(leaving, e_col, row))
    both(lambda e: e.run_pivots(lpamd.MAX_REDUCTION, depth + 1))
    both(lambda e: e.run_pivots(lpamd.MAX_COST, 2 * depth + 1))
    both(lambda e: e.run_pivots(lpamd.MAX_REDUCTION, 3 * depth))
    assert keep.trace() == fresh.trace()
    a, b = keep.download(), fresh.download()
    for x, y in zip(a[:2], b[:2]):
        assert np.array_equal(bits(x), bits(y))
    # a new tableau in the same engines
    both(lambda e: e.fill_synthetic(n, seed + 1))
    dt2 = lo.DenseTableau.synthetic(m, n, seed + 1)
    both(lambda e: e.run_pivots(lpamd.MAX_REDUCTION, 2 * depth + 1))
    assert dt2.solve(lo.MAX_REDUCTION, max_pivots=2 * depth + 1) == lo.LIMIT
    assert keep.trace() == fresh.trace() == dt2.trace()
    assert_same_tableau(keep, dt2)
    keep.close()
    fresh.close()
