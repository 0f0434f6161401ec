"""GPU parity of the revised simplex on the tableau (lpamd_rev_* through the C ABI)
against the CPU oracle and the reference's goldens.  Bit-exact bar: the pivot
trace (entering column, leaving basis position), the basis inverse, the basic
solution, the basis and the optimum must be identical."""
import numpy as np
import pytest

from cases import REVISED_CASES, case_text, revised_golden
from oracle import lp_oracle as lo

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_same_state(eng, rt):
    Binv, BinvT, xB, cB, basis = eng.read_state()
    assert np.array_equal(basis, rt.basis), "basis differs"
    assert np.array_equal(bits(xB), bits(rt.xB)), "basic solution differs"
    assert np.array_equal(bits(cB), bits(rt.cB)), "basis cost differs"
    assert np.array_equal(bits(Binv), bits(rt.Binv)), "basis inverse differs"
    assert np.array_equal(bits(BinvT), bits(np.ascontiguousarray(rt.Binv.T))), "transposed storage differs"


def upload(lpamd, tab):
    eng = lpamd.RevisedTableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic, tab.opt_reverted)
    return eng


@pytest.mark.parametrize("name,golden", REVISED_CASES)
def test_revised_matches_oracle_and_golden(lpamd, name, golden):
    tab, rt = lo.revised_from_text(case_text(name))
    want = rt.solve()
    eng = upload(lpamd, tab)
    res, n = eng.solve()
    assert res == want == lpamd.SOLVED
    assert eng.trace() == rt.trace()
    assert n == rt.n_pivots
    assert_same_state(eng, rt)
    assert bits(np.array([eng.optimum()]))[0] == bits(np.array([rt.optimum]))[0]
    x = eng.column_values()
    assert np.array_equal(bits(x), bits(rt.column_values()))

    class Got:   # the formatter only needs these two
        optimum = eng.optimum()

        @staticmethod
        def column_values():
            return x
    assert lo.format_revised(tab, Got).rstrip("\n") == revised_golden(golden).rstrip("\n")
    eng.close()


@pytest.mark.parametrize("m,n,seed", [(96, 200, 1), (130, 70, 2), (257, 513, 3)])
def test_revised_synthetic_trace_and_inverse(lpamd, m, n, seed):
    dt = lo.DenseTableau.synthetic(m, n, seed)
    rt = lo.RevisedTableau(dt.T, dt.obj, dt.is_basic, n + m, False)
    want = rt.solve()
    eng = lpamd.RevisedTableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    res, npiv = eng.solve()
    assert res == want
    assert npiv == rt.n_pivots and npiv > 0
    assert eng.trace() == rt.trace()
    assert_same_state(eng, rt)
    if res == lpamd.SOLVED:
        assert eng.optimum() == rt.optimum
    eng.close()


@pytest.mark.parametrize("knobs", [{"LPAMD_REV_ROWDOT_ROWS": "64"}, {"LPAMD_REV_ROWDOT_ROWS": "16"}, {"LPAMD_REV_ROWDOT_ROWS": "1"},
                                   {"LPAMD_REV_ROWDOT_ROWS": "1", "LPAMD_REV_UNIT": "0"}, {"LPAMD_REV_UPDATE_UNROLL": "16"},
                                   {"LPAMD_REV_TILE": "64"}, {"LPAMD_REV_TILE": "16"}, {"LPAMD_REV_TILE": "0"},
                                   {"LPAMD_REV_TILE": "0", "LPAMD_REV_COLSUM_VEC": "2"}, {"LPAMD_REV_SPLIT": "1"},
                                   {"LPAMD_REV_LIST": "0"}, {"LPAMD_REV_LIST": "0", "LPAMD_REV_TILE": "64"},
                                   {"LPAMD_REV_UNIT": "0"}, {"LPAMD_REV_UNIT": "0", "LPAMD_REV_ROWDOT_ROWS": "64"},
                                   {"LPAMD_REV_WIDE": "0"}, {"LPAMD_REV_WIDE": "0", "LPAMD_REV_SPLIT": "1"},
                                   {"LPAMD_REV_SPLIT": "1", "LPAMD_REV_TILE": "16"}, {"LPAMD_REV_SPLIT": "1", "LPAMD_REV_UNIT": "0"},
                                   {"LPAMD_REV_ROWDOT_ROWS": "64", "LPAMD_REV_UPDATE_UNROLL": "16", "LPAMD_REV_TILE": "64",
                                    "LPAMD_REV_SPLIT": "1"}])
@pytest.mark.parametrize("m,n,seed", [(200, 333, 4), (65, 129, 5)])
def test_revised_large_problem_kernel_shapes_on_small_lps(lpamd, monkeypatch, knobs, m, n, seed):
    """The kernel variants picked for large m / wide tableaus, forced on small LPs the oracle solves fully."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    dt = lo.DenseTableau.synthetic(m, n, seed)
    rt = lo.RevisedTableau(dt.T, dt.obj, dt.is_basic, n + m, False)
    want = rt.solve()
    eng = lpamd.RevisedTableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    res, npiv = eng.solve()
    assert res == want and npiv == rt.n_pivots
    assert eng.trace() == rt.trace()
    assert_same_state(eng, rt)
    eng.close()


def test_revised_phase1_with_large_problem_kernel_shapes(lpamd, monkeypatch):
    for k, v in {"LPAMD_REV_ROWDOT_ROWS": "64", "LPAMD_REV_UPDATE_UNROLL": "16", "LPAMD_REV_TILE": "64",
                 "LPAMD_REV_SPLIT": "1"}.items():
        monkeypatch.setenv(k, v)
    tab, rt = lo.revised_from_text(case_text("israel"))
    want = rt.solve()
    eng = upload(lpamd, tab)
    res, n = eng.solve()
    assert res == want == lpamd.SOLVED and eng.trace() == rt.trace()
    assert_same_state(eng, rt)
    eng.close()


def test_revised_wide_tableau_and_many_rows(lpamd):
    """Default kernel selection at sizes that switch variants: wide tableau (64-wide tiles in the
    pricing product) and m >= 16384 (64-row tiles, 16-deep update runs); first pivots only."""
    for m, n, seed, piv in [(64, 70000, 11, 12), (16384, 300, 12, 4)]:
        dt = lo.DenseTableau.synthetic(m, n, seed)
        rt = lo.RevisedTableau(dt.T, dt.obj, dt.is_basic, n + m, False)
        del dt
        want = rt.solve(max_pivots=piv)
        eng = lpamd.RevisedTableauEngine(m, n + m)
        eng.fill_synthetic(n, seed)
        res, npiv = eng.solve(max_pivots=piv)
        assert res == want and npiv == rt.n_pivots
        assert eng.trace() == rt.trace()
        _, BinvT, xB, cB, basis = eng.read_state(inverse=False)
        assert np.array_equal(basis, rt.basis) and np.array_equal(bits(xB), bits(rt.xB)) and np.array_equal(bits(cB), bits(rt.cB))
        assert np.array_equal(BinvT.view(np.uint64), rt.Binv.T.view(np.uint64)), "basis inverse differs"
        eng.close()
        del rt, BinvT


@pytest.mark.parametrize("m,n,seed", [(1, 1, 1), (1, 100, 2), (100, 1, 3), (513, 3, 4), (2, 1100, 5), (63, 64, 6), (1025, 17, 7)])
def test_revised_extreme_shapes(lpamd, m, n, seed):
    dt = lo.DenseTableau.synthetic(m, n, seed)
    rt = lo.RevisedTableau(dt.T, dt.obj, dt.is_basic, n + m, False)
    want = rt.solve()
    eng = lpamd.RevisedTableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    res, npiv = eng.solve()
    assert res == want and npiv == rt.n_pivots
    assert eng.trace() == rt.trace()
    assert_same_state(eng, rt)
    eng.close()


def test_revised_full_size_properties(lpamd):
    """m=32768 n=65536 (the benchmark's size): no oracle run at this size, so size-independent
    properties: the run is deterministic, the basic solution stays feasible, every entering
    column was non-basic and every basis position holds a distinct column."""
    m, n, seed, piv = 32768, 65536, 1, 6
    runs = []
    for _ in range(2):
        eng = lpamd.RevisedTableauEngine(m, n + m)
        eng.fill_synthetic(n, seed)
        res, npiv = eng.solve(max_pivots=piv)
        assert res == lpamd.LIMIT and npiv == piv
        _, _, xB, cB, basis = eng.read_state(inverse=False, transposed=False)
        runs.append((eng.trace(), xB.copy(), basis.copy()))
        eng.close()
    assert runs[0][0] == runs[1][0]
    assert np.array_equal(bits(runs[0][1]), bits(runs[1][1])) and np.array_equal(runs[0][2], runs[1][2])
    trace, xB, basis = runs[0]
    assert (xB >= 0).all()
    assert len(set(basis.tolist())) == m
    entered = [e for e, _ in trace]
    assert all(0 <= e < n for e in entered)          # only structural columns carry a cost
    for e, l in trace:
        assert 0 <= l < m
    assert set(basis.tolist()) - set(range(n, n + m)) <= set(entered)   # structural columns got in through a pivot


def test_revised_pivot_limit_and_continue(lpamd):
    m, n, seed = 120, 300, 7
    dt = lo.DenseTableau.synthetic(m, n, seed)
    full = lo.RevisedTableau(dt.T, dt.obj, dt.is_basic, n + m, False)
    assert full.solve() == lo.SOLVED
    part = lo.RevisedTableau(dt.T, dt.obj, dt.is_basic, n + m, False)
    assert part.solve(max_pivots=5) == lo.LIMIT
    eng = lpamd.RevisedTableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    res, npiv = eng.solve(max_pivots=5)
    assert res == lpamd.LIMIT and npiv == 5
    assert eng.trace() == part.trace()
    assert_same_state(eng, part)
    res, npiv = eng.solve()
    assert res == lpamd.SOLVED and npiv == full.n_pivots
    assert eng.trace() == full.trace()
    assert_same_state(eng, full)
    with pytest.raises(lpamd.LpamdError):
        eng.solve()
    eng.close()


def test_revised_phase1_limit(lpamd):
    """A budget that ends inside the feasibility phase reports LIMIT like the oracle."""
    tab, rt = lo.revised_from_text(case_text("afiro"))
    assert rt.solve(max_pivots=3) == lo.LIMIT
    eng = upload(lpamd, tab)
    res, n = eng.solve(max_pivots=3)
    assert res == lpamd.LIMIT and n == 3
    assert eng.trace() == rt.trace()
    eng.close()


def test_revised_infeasible_and_unbounded(lpamd):
    infeasible = "max x1\nst\nx1 <= 1\nx1 >= 2\nx1 >= 0\n"
    unbounded = "max x1 + x2\nst\nx1 - x2 <= 1\nx1 >= 0\nx2 >= 0\n"
    for text in (infeasible, unbounded):
        tab, rt = lo.revised_from_text(text)
        want = rt.solve()
        eng = upload(lpamd, tab)
        res, n = eng.solve()
        assert res == want
        assert eng.trace() == rt.trace()
        eng.close()
    assert want == lo.UNBOUNDED
