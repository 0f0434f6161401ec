"""GPU, the sizes of the headline benchmark.  m=4096 n=8192 is still cheap enough for the oracle, so the
first pivots are compared bit for bit; m=32768 n=65536 (25.8 GB on the device)
is checked through size-independent properties of the simplex tableau."""
import numpy as np
import pytest

from oracle import lp_oracle as lo

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("strategy,k", [("max_cost", 24), ("max_reduction", 6)])
def test_m4096_n8192_first_pivots_bit_exact(lpamd, strategy, k):
    m, n, seed = 4096, 8192, 1
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    dt = lo.DenseTableau.synthetic(m, n, seed)
    dt.s.skip_zero_cols = 1      # equivalence with the plain restatement is covered by the CPU tests
    dt.s.fast_basis = 1
    assert eng.run_pivots(lpamd.STRATEGY[strategy], k) == (lpamd.LIMIT, k)
    assert dt.solve(lo.STRATEGY[strategy], max_pivots=k) == lo.LIMIT
    assert eng.trace() == dt.trace()
    T, obj, basic = eng.download()
    assert np.array_equal(bits(T), bits(dt.T))
    assert np.array_equal(bits(obj), bits(dt.obj))
    assert np.array_equal(basic[: dt.N], dt.is_basic[: dt.N])
    eng.close()


def test_m32768_n65536_properties(lpamd):
    m, n, seed, k = 32768, 65536, 1, 48
    N = n + m
    eng = lpamd.TableauEngine(m, N)
    eng.fill_synthetic(n, seed)
    # the device generator equals the oracle's entry function on sampled entries
    rng = np.random.default_rng(0)
    for j in list(rng.integers(0, n, 3)) + [n + 5, N]:
        col = eng.read_column(int(j))
        for i in rng.integers(0, m, 8):
            assert col[i] == lo.lib().lpo_synth_entry(seed, m, n, int(i), int(j))
    objs = [eng.objective_constant()]
    for _ in range(4):
        res, done = eng.run_pivots(lpamd.MAX_COST, k // 4)
        assert (res, done) == (lpamd.LIMIT, k // 4)
        objs.append(eng.objective_constant())
    assert all(b > a for a, b in zip(objs, objs[1:])), "objective must increase on this non-degenerate LP"
    trace = eng.trace()
    assert len(trace) == k
    assert all(0 <= e < N and 0 <= r < m for e, r in trace)
    rhs = eng.read_column(N)
    assert (rhs.astype(np.float32) > np.float32(-1e-6)).all(), "primal feasibility lost"
    basis = eng.basis_of_row()
    for e, r in trace[-8:]:
        if basis[r] != e:
            continue                      # left the basis again later
        col = eng.read_column(e)          # a basic column is an exact unit column: one entry near -1, zeros elsewhere
        nz = np.nonzero(col)[0]
        assert list(nz) == [r] and abs(col[r] + 1.0) < 1e-9
    # determinism: a second engine replays the same sequence
    eng2 = lpamd.TableauEngine(m, N)
    eng2.fill_synthetic(n, seed)
    eng2.run_pivots(lpamd.MAX_COST, k)
    assert eng2.trace() == trace
    assert eng2.objective_constant() == objs[-1]
    eng.close()
    eng2.close()
