"""GPU: one LP over column shards driven by the library itself (lpamd_group_*): the pivot loop, the
feasibility phase and the exchange live behind the C ABI.  On the one-GPU test box the shards share
the GPU and trade through device copies (LPAMD_GROUP_LOCAL_EXCHANGE); RCCL itself runs with a world
of one.  Bar: every shard walks the oracle's pivots, the assembled tableau, the objective row and
the right-hand-side replicas equal the oracle bit for bit (reference lp_simplex_tableau.cc:10-52,
64-171, 173-379)."""
import numpy as np
import pytest

from cases import case_golden, case_text
from oracle import lp_frontend as fe
from oracle import lp_oracle as lo

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def check_against(group, dt, want_trace):
    N = group.N
    T, obj, basic = group.download()
    assert np.array_equal(bits(T), bits(dt.T)), "tableau differs from the oracle"
    assert np.array_equal(bits(obj), bits(dt.obj))
    assert np.array_equal(basic[:N], dt.is_basic[:N])
    rhs0 = None
    for k in range(group.n_local):
        eng = group.engine(k)
        assert eng.trace() == want_trace, "shard %d walked other pivots" % k
        rhs = eng.read_column(N)
        rhs0 = rhs if rhs0 is None else rhs0
        assert np.array_equal(bits(rhs), bits(rhs0)), "RHS replicas diverged"


@pytest.mark.parametrize("strategy", ["max_cost", "max_reduction"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_group_synthetic_to_the_optimum(lpamd, strategy, world):
    m, n, seed = 700, 1300, 11
    g = lpamd.ShardGroup(m, n + m, world, devices=[0] * world, local_exchange=True)
    g.fill_synthetic(n, seed)
    dt = lo.DenseTableau.synthetic(m, n, seed)
    res, piv = g.solve(lpamd.STRATEGY[strategy])
    assert res == dt.solve(lo.STRATEGY[strategy]) == lpamd.SOLVED
    assert piv == dt.n_pivots == g.pivot_count()
    assert g.trace() == dt.trace()
    check_against(g, dt, dt.trace())
    x = g.column_values()
    assert np.array_equal(bits(x), bits(dt.column_values()))
    assert g.objective_constant() == dt.obj[dt.N]
    g.close()


@pytest.mark.parametrize("fused", [3, 2, 1, 0])
@pytest.mark.parametrize("chain", [1, 3, 8])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_group_budgets_end_inside_a_chain(lpamd, monkeypatch, world, chain, fused):
    """MAX_COST legs whose budgets end at different positions of a chain of pivots per pass, on
    shards long enough (m > 2048 rows) that the exchange stream really runs beside the pass.
    fused: a chained step is set up straight from the gathered slots (the ratio-test partials travel with
    the column), 2: by the one launch that also writes the rank's slot for the step after it (k_shard_step, several
    shards only), 3: the launches also store the slots into every shard's receive buffer, no collective inside a chain; 1: by a set-up launch and an export launch; 0: with the fold kernel and a
    ratio-test launch of its own."""
    m, n, seed, legs = 2560, 3072, 7, (23, 1, 40, 0, 9)
    if chain == 1 and fused != 1:
        pytest.skip("a chain of one has no chained step")
    if world == 1 and fused >= 2:
        pytest.skip("one shard never takes the one-launch step")
    if fused == 3 and chain < 2:
        pytest.skip("a chain of one has no chained step")
    monkeypatch.setenv("LPAMD_SHARD_CHAIN", str(chain))
    monkeypatch.setenv("LPAMD_SHARD_FUSED", str(min(fused, 1)))
    monkeypatch.setenv("LPAMD_SHARD_STEP", "1" if fused >= 2 else "0")
    monkeypatch.setenv("LPAMD_SHARD_LINK", "1" if fused == 3 else "0")
    g = lpamd.ShardGroup(m, n + m, world, devices=[0] * world, local_exchange=True)
    assert g.chain() == chain
    assert g.link_active() == (fused == 3)
    g.fill_synthetic(n, seed)
    dt = lo.DenseTableau.synthetic(m, n, seed)
    dt.s.skip_zero_cols = 1
    dt.s.fast_basis = 1
    total = 0
    for k in legs:
        assert g.run_pivots(lpamd.MAX_COST, k) == (lpamd.LIMIT, k)
        total += k
        assert dt.solve(lo.MAX_COST, max_pivots=total) == lo.LIMIT
        assert g.trace() == dt.trace()
    assert g.run_pivots(lpamd.MAX_REDUCTION, 5) == (lpamd.LIMIT, 5)     # the in-order loop continues the same LP
    assert dt.solve(lo.MAX_REDUCTION, max_pivots=total + 5) == lo.LIMIT
    check_against(g, dt, dt.trace())
    g.close()


@pytest.mark.parametrize("name,strategy,world", [
    ("test15", "max_cost", 2), ("test16", "max_reduction", 3), ("afiro", "max_cost", 3), ("kb2", "max_reduction", 2),
    ("adlittle", "max_cost", 4), ("israel", "max_cost", 2), ("adlittle", "max_reduction", 8), ("test17", "max_cost", 1),
])
def test_group_solve_with_feasibility_phase(lpamd, name, strategy, world):
    """TableauSimplexSolve over the shards, feasibility phase, artificial pivot-out and objective
    re-expression included; the solution prints as the reference's golden."""
    tab, dt = lo.from_text(case_text(name))
    want = dt.solve(lo.STRATEGY[strategy])
    g = lpamd.ShardGroup(tab.m, tab.N, world, devices=[0] * world, local_exchange=True)
    g.upload(tab.T, tab.obj, tab.is_basic)
    res, piv = g.solve(lpamd.STRATEGY[strategy])
    assert res == want == lpamd.SOLVED and piv == dt.n_pivots
    check_against(g, dt, dt.trace())
    x = g.column_values()
    assert np.array_equal(bits(x), bits(dt.column_values()))
    assert fe.format_solution(tab, x, g.objective_constant()).rstrip("\n") == case_golden(name, strategy).rstrip("\n")
    g.close()


def test_group_reports_infeasible_and_unbounded(lpamd):
    for text, want in (("max x1\nst\nx1 <= 1\nx1 >= 2\nx1 >= 0\n", lo.NOSOLUTION),
                       ("max x1 + x2\nst\nx1 - x2 <= 1\nx1 >= 0\nx2 >= 0\n", lo.UNBOUNDED)):
        for strategy in ("max_cost", "max_reduction"):
            tab, dt = lo.from_text(text)
            assert dt.solve(lo.STRATEGY[strategy]) == want
            g = lpamd.ShardGroup(tab.m, tab.N, 2, devices=[0, 0], local_exchange=True)
            g.upload(tab.T, tab.obj, tab.is_basic)
            res, piv = g.solve(lpamd.STRATEGY[strategy])
            assert res == want and g.trace() == dt.trace()
            g.close()


def test_group_explicit_pivot(lpamd):
    """LPModel::TableauPivot over shards (reference lp_test.cc:492-513 walks two explicit pivots)."""
    tab, dt = lo.from_text(case_text("test3"))
    g = lpamd.ShardGroup(tab.m, tab.N, 2, devices=[0, 0], local_exchange=True)
    g.upload(tab.T, tab.obj, tab.is_basic)
    for _ in range(2):
        e = dt.price_max_cost()
        r, d = dt.ratio_test(e)
        assert dt.pivot(d, e, r) == 0
        g.pivot(d, e, r)
    check_against(g, dt, dt.trace())
    g.close()


@pytest.mark.parametrize("strategy", ["max_cost", "max_reduction"])
def test_group_over_rccl_world_of_one(lpamd, strategy):
    """The RCCL path (communicator, all-gather on both streams) with the one rank a one-GPU box allows."""
    m, n, seed = 2560, 3072, 9
    g = lpamd.ShardGroup(m, n + m, 1)
    g.fill_synthetic(n, seed)
    dt = lo.DenseTableau.synthetic(m, n, seed)
    dt.s.skip_zero_cols = 1
    dt.s.fast_basis = 1
    assert g.run_pivots(lpamd.STRATEGY[strategy], 21) == (lpamd.LIMIT, 21)
    assert dt.solve(lo.STRATEGY[strategy], max_pivots=21) == lo.LIMIT
    check_against(g, dt, dt.trace())
    tab, dt = lo.from_text(case_text("israel"))
    g2 = lpamd.ShardGroup(tab.m, tab.N, 1)
    g2.upload(tab.T, tab.obj, tab.is_basic)
    res, _ = g2.solve(lpamd.STRATEGY[strategy])
    assert res == dt.solve(lo.STRATEGY[strategy]) == lpamd.SOLVED
    check_against(g2, dt, dt.trace())
    g.close()
    g2.close()


def test_group_join_with_unique_id_world_of_one(lpamd):
    """The one-process-per-GPU form: id from lpamd_comm_unique_id, ncclCommInitRank."""
    m, n, seed = 300, 500, 4
    g = lpamd.ShardGroup.join(m, n + m, 0, 1, lpamd.comm_unique_id())
    g.fill_synthetic(n, seed)
    dt = lo.DenseTableau.synthetic(m, n, seed)
    assert g.solve(lpamd.MAX_COST)[0] == dt.solve(lo.MAX_COST) == lpamd.SOLVED
    check_against(g, dt, dt.trace())
    g.close()


def test_group_rejects_bad_arguments(lpamd):
    with pytest.raises(lpamd.LpamdError, match="used twice"):
        lpamd.ShardGroup(64, 100, 2, devices=[0, 0])
    with pytest.raises(lpamd.LpamdError, match="bad shape"):
        lpamd.ShardGroup(64, 100, 0)
    with pytest.raises(lpamd.LpamdError, match="visible"):
        lpamd.ShardGroup(64, 100, 2, devices=[0, 99], local_exchange=True)
