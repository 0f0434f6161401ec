"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the
reference's golden outputs.  Bit-exact bar: pivot trace (e, r), every tableau
entry, the objective row and the basis flags must be identical."""
import os

import numpy as np
import pytest

from cases import all_cases, case_golden, case_text
from oracle import lp_frontend as fe
from oracle import lp_oracle as lo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


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


_25FV47 = {}


def _oracle_25fv47(strategy):
    """The largest real instance the reference ships (822 rows before bounds; 1337 x 2910 as a tableau).  The reference
    excludes it from its own tests for its run time (reference CMakeLists.txt:130), so there is no golden output: the
    oracle's pivot sequence is what pins the GPU here (parity with the reference itself is unpinned for this input)."""
    if strategy not in _25FV47:
        tab, dt = lo.from_text(case_text("25fv47"), trace_cap=1 << 17)
        res = dt.solve(lo.STRATEGY[strategy])
        _25FV47[strategy] = (tab, dt, res)
    return _25FV47[strategy]


@pytest.mark.parametrize("knobs", [{}, {"LPAMD_CHAIN": "8"}, {"LPAMD_CHAIN": "22", "LPAMD_FUSED_THREADS": "512"}], ids=["default", "chain8", "chain22"])
@pytest.mark.parametrize("strategy", ["max_cost", "max_reduction"])
def test_25fv47_matches_oracle(lpamd, monkeypatch, strategy, knobs):
    """Degenerate, 26 k pivots under MAX_COST (feasibility phase included): trace, every tableau entry, objective row,
    basis flags and the extracted solution bit-equal to the oracle; with the default loop for short columns and with
    chains of pivots forced."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    tab, dt, want = _oracle_25fv47(strategy)
    assert want == lo.SOLVED and dt.n_pivots > 5000
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    res, n = eng.solve(lpamd.STRATEGY[strategy])
    assert res == lpamd.SOLVED and n == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    x = eng.column_values()
    assert np.array_equal(bits(x), bits(dt.column_values()))
    assert bits(np.array([eng.objective_constant()]))[0] == bits(np.array([dt.obj[tab.N]]))[0]
    eng.close()


def test_qap15_first_pivots_match_the_oracle(lpamd):
    """datasets/plato/qap15.mps (6330 equality rows: 12660 x 34937 as a tableau, 3.5 GB).  The reference ships the input
    without an expected output, and the whole solve takes several hundred thousand pivots, which the CPU oracle (a few
    pivots per second at this size) cannot follow; what is pinned is the start: the pivots of the oracle's first
    `pivot_budget` pivots (feasibility phase, reference lp_simplex_tableau.cc:64-111, then its inner solve), computed once
    by tests/golden/make_qap15_outcome.py, against the engine's, pivot by pivot through a digest, and the objective
    constant at that point bit for bit.  Parity with the reference itself is unpinned for this input."""
    import hashlib
    import json
    from oracle import mps_to_text as mt
    want = json.load(open(os.path.join(HERE, "golden", "plato", "qap15_oracle_outcome_max_cost.json")))
    tab = fe.build_tableau(mt.mps_to_text(os.path.join(HERE, "golden", "plato", "qap15.mps")))
    assert (tab.m, tab.N) == (want["m"], want["N"])
    eng = lpamd.TableauEngine(tab.m, tab.N)
    eng.upload(tab.T, tab.obj, tab.is_basic)
    res, n = eng.solve(lpamd.MAX_COST, max_pivots=want["pivot_budget"])
    assert res == lpamd.LIMIT == want["result"] and n == want["n_pivots"]
    trace = eng.trace()
    assert [list(p) for p in trace[:16]] == want["first_pivots"] and [list(p) for p in trace[-16:]] == want["last_pivots"]
    h = hashlib.sha256()
    for e, r in trace:
        h.update(b"%d %d\n" % (e, r))
    assert h.hexdigest() == want["trace_sha256"]
    assert "%016x" % int(bits(np.array([eng.objective_constant()]))[0]) == want["objective_constant_bits"]
    eng.close()


def _lp_with_tiny_right_hand_sides(m, n, seed):
    """Dense feasible LP in tableau form whose right-hand side holds exact zeros (degenerate first pivots) and stored
    entries BELOW the drop threshold (the reference keeps a right-hand side however small, reference lp.cc:180)."""
    rng = np.random.default_rng(seed)
    N = n + m
    T = np.zeros((m, N + 2), order="F")
    T[:, :n] = -(0.1 + rng.random((m, n)))
    T[np.arange(m), n + np.arange(m)] = -1.0
    b = (1.0 + rng.random(m)) * n
    b[rng.choice(m, max(2, m // 8), replace=False)] = 0.0
    tiny = rng.choice(m, max(2, m // 8), replace=False)
    b[tiny] = np.where(rng.random(len(tiny)) < 0.5, 5e-7, 9.9e-7)
    T[:, N] = b
    obj = np.zeros(N + 2)
    obj[:n] = 0.5 + rng.random(n)
    basic = np.zeros(N + 2, dtype=np.uint8)
    basic[n:N] = 1
    return T, obj, basic, N


@pytest.mark.parametrize("strategy", ["max_cost", "max_reduction"])
@pytest.mark.parametrize("m,n", [(5, 4), (40, 30), (2304, 60)])
def test_stored_entries_below_the_drop_threshold(lpamd, strategy, m, n):
    """The reference's update adds to every entry and drops what ends up below 1e-6, so its FIRST pivot clears an uploaded
    right-hand side of 5e-7 even when that pivot is degenerate (pivot-row right-hand side 0, row' of the right-hand-side
    column 0) -- a column the engine would otherwise leave alone.  Plain dense oracle (no shortcuts) against the GPU, short
    columns (two-launch loop) and long ones (chains of pivots per pass)."""
    T, obj, basic, N = _lp_with_tiny_right_hand_sides(m, n, 3)
    dt = lo.DenseTableau(T, obj, basic, N)
    eng = lpamd.TableauEngine(m, N)
    eng.upload(T, obj, basic)
    # first pivot alone: degenerate, and the right-hand side loses its tiny entries
    res, done = eng.run_pivots(lpamd.STRATEGY[strategy], 1)
    assert dt.solve(lo.STRATEGY[strategy], max_pivots=1) == lo.LIMIT and (res, done) == (lpamd.LIMIT, 1)
    assert eng.trace() == dt.trace()
    assert dt.T[dt.trace()[0][1], N] == 0.0, "the first pivot is meant to be degenerate"
    assert not np.any((dt.T[:, N] != 0) & (np.abs(dt.T[:, N]) < 1e-6))
    assert_same_tableau(eng, dt)
    # (the plain oracle discovers the basis with an O(m * m) scan per MAX_REDUCTION pivot: a budget keeps the long columns affordable)
    budget = 70 if (m > 1000 and strategy == "max_reduction") else -1
    want = dt.solve(lo.STRATEGY[strategy], max_pivots=dt.n_pivots + budget if budget > 0 else -1)
    res, npiv = eng.run_pivots(lpamd.STRATEGY[strategy], budget)
    assert res == want and eng.pivot_count() == dt.n_pivots
    assert eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()
    # and in one go, from a fresh upload
    dt = lo.DenseTableau(T, obj, basic, N)
    want = dt.solve(lo.STRATEGY[strategy], max_pivots=budget)
    eng = lpamd.TableauEngine(m, N)
    eng.upload(T, obj, basic)
    res, npiv = eng.run_pivots(lpamd.STRATEGY[strategy], budget)
    assert res == want and npiv == dt.n_pivots and eng.trace() == dt.trace()
    assert_same_tableau(eng, dt)
    eng.close()


@pytest.mark.parametrize("strategy", ["max_cost", "max_reduction"])
def test_stored_entries_below_the_drop_threshold_over_shards(lpamd, strategy):
    m, n = 300, 50
    T, obj, basic, N = _lp_with_tiny_right_hand_sides(m, n, 5)
    dt = lo.DenseTableau(T, obj, basic, N)
    want = dt.solve(lo.STRATEGY[strategy])
    g = lpamd.ShardGroup(m, N, 3, devices=[0, 0, 0], local_exchange=True)
    g.upload(T, obj, basic)
    res, npiv = g.solve(lpamd.STRATEGY[strategy])
    assert res == want and npiv == dt.n_pivots and g.trace() == dt.trace()
    Tg, og, bg = g.download()
    assert np.array_equal(bits(Tg), bits(dt.T)) and np.array_equal(bits(og), bits(dt.obj))
    g.close()
