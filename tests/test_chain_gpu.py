"""GPU parity of the loop that applies a chain of MAX_COST pivots per pass over the tableau
(k_chain_ratio / k_chain_finalize / k_update_chain), forced on inputs the oracle solves fully:
trace, tableau, objective row and basis flags must be bit-identical to one pivot per pass."""
import numpy as np
import pytest

from cases import all_cases, case_golden, case_text
from oracle import lp_frontend as fe
from oracle import lp_oracle as lo
from test_engine_parity import _random_tied_lp, assert_same_tableau, bits

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[2, 3, 8, 16, 32])
def depth(request, monkeypatch):
    monkeypatch.setenv("LPAMD_CHAIN", str(request.param))
    return request.param


@pytest.mark.parametrize("name", sorted({n for n, s in all_cases() if s == "max_cost"}))
def test_goldens_chained_pivots(lpamd, name):
    tab, dt = lo.from_text(case_text(name))
    want = dt.solve(lo.MAX_COST)
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    res, n = eng.solve(lpamd.MAX_COST)
    assert res == want == lpamd.SOLVED and n == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    x = eng.column_values()
    assert np.array_equal(bits(x), bits(dt.column_values()))
    out = fe.format_solution(tab, x, eng.objective_constant())
    assert out.rstrip("\n") == case_golden(name, "max_cost").rstrip("\n")
    eng.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_lps_with_ties_chained_pivots(lpamd, seed):
    rng = np.random.default_rng(seed)
    text = _random_tied_lp(rng, int(rng.integers(3, 12)), int(rng.integers(2, 10)))
    tab, dt = lo.from_text(text)
    want = dt.solve(lo.MAX_COST)
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    res, n = eng.solve(lpamd.MAX_COST)
    assert res == want, text
    assert eng.trace() == dt.trace(), text
    assert_same_tableau(eng, dt)
    eng.close()


def test_budgets_and_continuation(lpamd, depth):
    """Budgets that end at every position of a chain, and runs that go on from there; then to the optimum."""
    m, n = 600, 1100
    dt = lo.DenseTableau.synthetic(m, n, 7)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, 7)
    total = 0
    assert eng.run_pivots(lpamd.MAX_COST, 0) == (lpamd.LIMIT, 0)      # an empty budget is spent, not an error
    for k in (1, 2, 37, 0, 1, 1, 4, 3, 8, 9, 16, 17):
        res, done = eng.run_pivots(lpamd.MAX_COST, k)
        total += k
        assert (res, done) == (lpamd.LIMIT, k)
        if k == 0:
            continue
        assert dt.solve(lo.MAX_COST, max_pivots=total) == lo.LIMIT
        assert eng.pivot_count() == total
        assert eng.trace() == dt.trace()
        assert_same_tableau(eng, dt)
        pivots, streamed, touched = eng.pass_stats()
        assert pivots == (k - 1) % depth + 1 and streamed <= touched <= pivots * streamed
    res, done = eng.run_pivots(lpamd.MAX_REDUCTION, 5)     # the other strategy in between
    assert (res, done) == (lpamd.LIMIT, 5)
    assert dt.solve(lo.MAX_REDUCTION, max_pivots=total + 5) == lo.LIMIT
    assert_same_tableau(eng, dt)
    res, _ = eng.solve(lpamd.MAX_COST)
    assert res == dt.solve(lo.MAX_COST) == lpamd.SOLVED
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


def test_max_cost_run_after_a_max_reduction_run_that_ended(lpamd):
    """A MAX_REDUCTION run that ends on a terminal status leaves its own scores in the partial buffers;
    a MAX_COST run started afterwards must price afresh instead of trusting them (it would report
    SOLVED on a tableau that still has positive reduced costs)."""
    tab, dt = lo.from_text("max x1 + x2\nst\nx1 - x2 <= 1\nx1 >= 0\nx2 >= 0\n")
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    assert eng.run_pivots(lpamd.MAX_REDUCTION, -1)[0] == dt.solve(lo.MAX_REDUCTION) == lpamd.UNBOUNDED
    assert eng.run_pivots(lpamd.MAX_COST, -1)[0] == dt.solve(lo.MAX_COST) == lpamd.UNBOUNDED
    assert eng.run_pivots(lpamd.MAX_COST, 3)[0] == lpamd.UNBOUNDED
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


@pytest.mark.parametrize("m,n", [(1, 1), (1, 5), (3, 1), (511, 2), (512, 3), (513, 3), (2, 700), (2500, 40)])
def test_extreme_shapes_chained_pivots(lpamd, m, n):
    dt = lo.DenseTableau.synthetic(m, n, 2)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, 2)
    res, _ = eng.solve(lpamd.MAX_COST)
    assert res == dt.solve(lo.MAX_COST)
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


def test_unbounded_found_inside_a_chain(lpamd):
    # after x1 enters (row 0), x2 has no eligible row: the pass applies the first pivot alone
    tab, dt = lo.from_text("max x1 + x2\nst\nx1 - x2 <= 1\nx1 >= 0\nx2 >= 0\n")
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    assert eng.solve(lpamd.MAX_COST)[0] == dt.solve(lo.MAX_COST) == lpamd.UNBOUNDED
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


def test_default_loop_chains_long_columns(lpamd, monkeypatch, depth):
    """m > 2048 on one GPU chains pivots by default; LPAMD_CHAIN=1 gives the same bits."""
    if depth != 2:
        pytest.skip("one run is enough")
    m, n, k = 2304, 1500, 41
    monkeypatch.delenv("LPAMD_CHAIN")
    a = lpamd.TableauEngine(m, n + m)
    a.fill_synthetic(n, 5)
    d = a.loop_depth(lpamd.MAX_COST)
    assert d >= 8, "the default loop is expected to chain pivots for long columns"
    assert a.run_pivots(lpamd.MAX_COST, k) == (lpamd.LIMIT, k)
    assert a.pass_stats()[0] == (k - 1) % d + 1          # whole passes, then what is left of the budget
    assert a.run_pivots(lpamd.MAX_COST, k + 3) == (lpamd.LIMIT, k + 3)
    assert a.pass_stats()[0] == (k + 2) % d + 1
    monkeypatch.setenv("LPAMD_CHAIN", "1")
    b = lpamd.TableauEngine(m, n + m)
    b.fill_synthetic(n, 5)
    assert b.run_pivots(lpamd.MAX_COST, 2 * k + 3) == (lpamd.LIMIT, 2 * k + 3)
    assert b.pass_stats()[0] == 1
    assert a.trace() == b.trace()
    ta, tb = a.download(), b.download()
    for x, y in zip(ta[:2], tb[:2]):
        assert np.array_equal(bits(x), bits(y))
    assert np.array_equal(ta[2], tb[2])
    dt = lo.DenseTableau.synthetic(m, n, 5)
    assert dt.solve(lo.MAX_COST, max_pivots=2 * k + 3) == lo.LIMIT
    assert a.trace() == dt.trace()
    assert_same_tableau(a, dt)
    a.close()
    b.close()


@pytest.mark.parametrize("slab", [1, 2, 3, 4])
@pytest.mark.parametrize("chain,nt", [(4, "0"), (8, "1"), (16, "0"), (24, "1"), (32, "0")])
def test_slab_pass_with_col_in_registers(lpamd, monkeypatch, depth, slab, chain, nt):
    """k_update_slab (col' slices in registers, a workgroup per row slab) against the oracle: full
    solve, so short chains at the end, columns a step leaves alone and list tails all occur."""
    if depth != 2:
        pytest.skip("one run is enough")
    monkeypatch.setenv("LPAMD_CHAIN", str(chain))
    monkeypatch.setenv("LPAMD_SLAB", str(slab))
    monkeypatch.setenv("LPAMD_DEEP", "0")
    monkeypatch.setenv("LPAMD_UPDATE_NT", nt)
    monkeypatch.setenv("LPAMD_SLAB_GRID", "24")
    m, n, seed = 2000, 900, 11
    dt = lo.DenseTableau.synthetic(m, n, seed)
    want = dt.solve(lo.MAX_COST)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    res, npiv = eng.solve(lpamd.MAX_COST)
    assert res == want == lpamd.SOLVED and npiv == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


def test_update_stats_average_over_passes_of_different_length(lpamd, monkeypatch, depth):
    """13 pivots with chains of 8 are a pass of 8 and a pass of 5: active_columns is the mean per launch."""
    if depth != 8:
        pytest.skip("one run is enough")
    m, n = 600, 1100
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, 7)
    eng.set_profiling(True)
    assert eng.run_pivots(lpamd.MAX_COST, 13) == (lpamd.LIMIT, 13)
    ms, launches, active = eng.update_stats()
    assert launches == 2 and ms > 0
    assert 6.5 * n <= active <= 6.5 * (n + 14)
    assert eng.pass_stats()[0] == 5
    eng.set_profiling(False)
    eng.close()


@pytest.mark.parametrize("batch", [4, 8])
@pytest.mark.parametrize("chain,nt,grid", [(3, "0", 16), (16, "1", 64), (24, "0", 40), (32, "1", 2048), (32, "0", 1)])
def test_deep_pass_with_col_in_lds(lpamd, monkeypatch, depth, batch, chain, nt, grid):
    """k_update_deep (col' slices in LDS, steps in the outer loop, batches of columns dealt over the wavefronts)
    against the oracle: full solve, so short chains at the end, columns a step leaves alone, list tails and
    fewer batches than wavefronts all occur."""
    if depth != 2:
        pytest.skip("one run is enough")
    monkeypatch.setenv("LPAMD_CHAIN", str(chain))
    monkeypatch.setenv("LPAMD_DEEP", str(batch))
    monkeypatch.setenv("LPAMD_UPDATE_NT", nt)
    monkeypatch.setenv("LPAMD_DEEP_GRID", str(grid))
    m, n, seed = 2000, 900, 11
    dt = lo.DenseTableau.synthetic(m, n, seed)
    want = dt.solve(lo.MAX_COST)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    res, npiv = eng.solve(lpamd.MAX_COST)
    assert res == want == lpamd.SOLVED and npiv == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


@pytest.mark.parametrize("chain", [3, 24, 32])
@pytest.mark.parametrize("knobs", [
    {"LPAMD_FUSED": "0"},                                                           # step-by-step set-up kernels
    {"LPAMD_FUSED": "1"},                                                           # default shape: one row / column per thread
    {"LPAMD_FUSED": "1", "LPAMD_FUSED_THREADS": "512"},
    {"LPAMD_FUSED": "1", "LPAMD_FUSED_THREADS": "1024"},                            # history from memory
    {"LPAMD_FUSED": "1", "LPAMD_FUSED_REG": "0"},
    {"LPAMD_FUSED": "1", "LPAMD_FUSED_ROWS": "2", "LPAMD_FUSED_COLS": "3"},         # several rows / columns per thread
    {"LPAMD_FUSED": "1", "LPAMD_FUSED_ROWS": "1", "LPAMD_FUSED_COLS": "1"},
], ids=lambda k: "-".join("%s%s" % (a.replace("LPAMD_FUSED", "F").replace("_", ""), b) for a, b in k.items()))
def test_fused_setup_of_a_chain(lpamd, monkeypatch, depth, chain, knobs):
    """k_chain_fused (steps 1.. of a chain set up by one launch of resident workgroups that hand results to each other
    through tagged records) in every shape against the oracle: full solve, so chains that end at the optimum, and budgets
    that end inside a chain."""
    if depth != 2:
        pytest.skip("one run is enough")
    monkeypatch.setenv("LPAMD_CHAIN", str(chain))
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    m, n, seed = 2000, 900, 11
    dt = lo.DenseTableau.synthetic(m, n, seed)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    total = 0
    for k in (1, 2, chain - 1, chain, chain + 1, 5):
        assert eng.run_pivots(lpamd.MAX_COST, k) == (lpamd.LIMIT, k)
        total += k
        assert dt.solve(lo.MAX_COST, max_pivots=total) == lo.LIMIT
        assert eng.pivot_count() == total
        assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    want = dt.solve(lo.MAX_COST)
    res, npiv = eng.solve(lpamd.MAX_COST)
    assert res == want == lpamd.SOLVED and npiv == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()
