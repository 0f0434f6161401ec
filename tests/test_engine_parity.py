"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the
reference's golden outputs.  Bit-exact bar: pivot trace (e, r), every tableau
entry, the objective row and the basis flags must be identical."""
import numpy as np
import pytest

from cases import all_cases, case_golden, case_text
from oracle import lp_frontend as fe
from oracle import lp_oracle as lo

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_same_tableau(eng, dt):
    T, obj, basic = eng.download()
    assert np.array_equal(bits(T), bits(dt.T)), "tableau differs"
    assert np.array_equal(bits(obj), bits(dt.obj)), "objective row differs"
    assert np.array_equal(basic[: dt.N], dt.is_basic[: dt.N]), "basis flags differ"


@pytest.mark.parametrize("name,strategy", all_cases())
def test_solve_matches_oracle_and_golden(lpamd, name, strategy):
    text = case_text(name)
    tab, dt = lo.from_text(text)
    want_res = dt.solve(lo.STRATEGY[strategy])
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    res, n = eng.solve(lpamd.STRATEGY[strategy])
    assert res == want_res == lpamd.SOLVED
    assert eng.trace() == dt.trace()
    assert n == dt.n_pivots
    assert_same_tableau(eng, dt)
    x = eng.column_values()
    assert np.array_equal(bits(x), bits(dt.column_values()))
    out = fe.format_solution(tab, x, eng.objective_constant())
    assert out.rstrip("\n") == case_golden(name, strategy).rstrip("\n")
    eng.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("strategy", ["max_cost", "max_reduction"])
def test_synthetic_trace_and_tableau(lpamd, seed, strategy):
    m, n = 96, 200
    dt = lo.DenseTableau.synthetic(m, n, seed)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    assert_same_tableau(eng, dt)
    # step-by-step: every primitive agrees before the pivot is applied
    for _ in range(10):
        if strategy == "max_cost":
            e = eng.price_max_cost()
            assert e == dt.price_max_cost()
            r, d = eng.ratio_test(e)
            assert (r, d) == dt.ratio_test(e)
        else:
            e, r, d = eng.price_max_reduction()
            assert (e, r, d) == dt.price_max_reduction()
        eng.pivot(d, e, r)
        dt.pivot(d, e, r)
        assert_same_tableau(eng, dt)
    # then the device-resident loop to optimality
    res, _ = eng.solve(lpamd.STRATEGY[strategy])
    assert res == dt.solve(lo.STRATEGY[strategy]) == lpamd.SOLVED
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


def test_run_pivots_budget(lpamd):
    m, n = 600, 1100   # two 512-row slabs, exercises the partial row chunk
    dt = lo.DenseTableau.synthetic(m, n, 7)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, 7)
    res, done = eng.run_pivots(lpamd.MAX_COST, 37)
    assert (res, done) == (lpamd.LIMIT, 37)
    assert dt.solve(lo.MAX_COST, max_pivots=37) == lo.LIMIT
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    res, done = eng.run_pivots(lpamd.MAX_REDUCTION, 5)
    assert (res, done) == (lpamd.LIMIT, 5)
    assert dt.solve(lo.MAX_REDUCTION, max_pivots=42) == lo.LIMIT
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


def test_unbounded_and_infeasible(lpamd):
    # max x1 s.t. x1 - x2 <= 1 : unbounded along x2
    tab, dt = lo.from_text("max x1 + x2\nst\nx1 - x2 <= 1\nx1 >= 0\nx2 >= 0\n")
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    for s in (lpamd.MAX_COST, lpamd.MAX_REDUCTION):
        eng.upload(tab.T, tab.obj, tab.is_basic)
        assert eng.solve(s)[0] == lpamd.UNBOUNDED
    assert dt.solve(lo.MAX_COST) == lo.UNBOUNDED
    eng.close()
    tab, dt = lo.from_text("max x1\nst\nx1 <= 1\nx1 >= 2\nx1 >= 0\n")
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    assert eng.solve(lpamd.MAX_COST)[0] == lpamd.NOSOLUTION
    assert dt.solve(lo.MAX_COST) == lo.NOSOLUTION
    eng.close()


def _random_tied_lp(rng, n_vars, n_cons):
    """Small-integer coefficients: many equal reduced costs and equal ratios, so the
    lowest-index tie-breaks of pricing and of the ratio test decide the path."""
    names = ["v%d" % k for k in range(n_vars)]
    obj = " + ".join("%d * %s" % (rng.integers(1, 4), v) for v in names)
    lines = ["max " + obj, "st"]
    for _ in range(n_cons):
        picks = rng.choice(n_vars, size=rng.integers(2, min(6, n_vars) + 1), replace=False)
        lhs = " + ".join("%d * %s" % (rng.integers(1, 4), names[k]) for k in sorted(picks))
        lines.append("%s <= %d" % (lhs, rng.integers(2, 7)))
    if rng.integers(0, 2):                       # an equality and a >= row force phase 1
        lines.append("%s + %s = %d" % (names[0], names[1], rng.integers(1, 3)))
        lines.append("%s >= 1" % names[2 % n_vars])
    lines += ["%s >= 0" % v for v in names]
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed", range(12))
def test_random_lps_with_ties(lpamd, seed):
    rng = np.random.default_rng(seed)
    text = _random_tied_lp(rng, int(rng.integers(3, 12)), int(rng.integers(2, 10)))
    for strategy in ("max_cost", "max_reduction"):
        tab, dt = lo.from_text(text)
        want = dt.solve(lo.STRATEGY[strategy])
        eng = lpamd.TableauEngine(tab.m, tab.N)
        eng.upload(tab.T, tab.obj, tab.is_basic)
        res, n = eng.solve(lpamd.STRATEGY[strategy])
        assert res == want, text
        assert eng.trace() == dt.trace(), text
        assert_same_tableau(eng, dt)
        eng.close()


@pytest.mark.parametrize("m,n", [(1, 1), (1, 5), (3, 1), (511, 2), (512, 3), (513, 3), (2, 700)])
def test_extreme_shapes(lpamd, m, n):
    dt = lo.DenseTableau.synthetic(m, n, 2)
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, 2)
    for strategy in ("max_reduction", "max_cost"):
        res, _ = eng.run_pivots(lpamd.STRATEGY[strategy], 3)
        want = dt.solve(lo.STRATEGY[strategy], max_pivots=dt.n_pivots + 3)
        assert res == want
        assert eng.trace() == dt.trace()
        assert_same_tableau(eng, dt)
    eng.close()


@pytest.mark.parametrize("knobs", [
    {"LPAMD_UPDATE_NT": "1", "LPAMD_UPDATE_PREC": "0", "LPAMD_UPDATE_PREC_FUSED": "0"},
    {"LPAMD_UPDATE_NT": "1", "LPAMD_UPDATE_PREC": "1", "LPAMD_UPDATE_PREC_FUSED": "1"},
    {"LPAMD_UPDATE_NT": "0", "LPAMD_UPDATE_PREC": "0", "LPAMD_UPDATE_PREC_FUSED": "0"},
    {"LPAMD_FUSE_PR": "0"},
    {"LPAMD_UPDATE_UNROLL": "1"},
    {"LPAMD_UPDATE_UNROLL": "4", "LPAMD_UPDATE_GRID": "7"},
    {"LPAMD_PLACEMENT": "0"},
    {"LPAMD_TWO_LAUNCH": "0"},
    {"LPAMD_TWO_LAUNCH": "0", "LPAMD_CHAIN": "1"},
    {"LPAMD_TWO_LAUNCH": "1", "LPAMD_UPDATE_UNROLL": "2"},
    {"LPAMD_CHAIN": "2"},
    {"LPAMD_CHAIN": "8", "LPAMD_UPDATE_NT": "1", "LPAMD_UPDATE_UNROLL": "2"},
    {"LPAMD_CHAIN": "3", "LPAMD_UPDATE_NT": "0", "LPAMD_UPDATE_GRID": "7"},
])
@pytest.mark.parametrize("strategy", ["max_cost", "max_reduction"])
def test_update_kernel_variants_are_bit_identical(lpamd, monkeypatch, knobs, strategy):
    """Every variant of the update kernel the engine may pick by shard size (cache policy, place of
    the col' load, fused pricing, unroll depth, grid) forced on one LP the oracle solves fully."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    m, n, seed = 1100, 700, 21
    dt = lo.DenseTableau.synthetic(m, n, seed)
    want = dt.solve(lo.STRATEGY[strategy])
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    res, npiv = eng.solve(lpamd.STRATEGY[strategy])
    assert res == want == lpamd.SOLVED and npiv == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()
