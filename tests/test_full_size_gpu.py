"""GPU, the sizes of the headline benchmark, each at its own size.

m=4096 n=8192 and m=32768 n=65536 (25.8 GB on the device) are compared with the CPU oracle bit for bit
(pivot trace, every tableau entry, objective row, basis flags) over two full chained passes, and the chained pass
against one pass per pivot; m=32768 n=65536 additionally through size-independent properties of the simplex
tableau; m=65536 n=262144 with max-reduction pricing (172 GB on the device, too large for the host oracle) through
the same properties plus the fused pricing against the separate pricing pass."""
import gc

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


def _sample_columns(rng, n, N, trace, k):
    cols = set(int(j) for j in rng.integers(0, n, k))
    cols |= {e for e, _ in trace} | {N, n, n + 1, N - 1}
    return sorted(cols)


def test_m32768_n65536_two_chained_passes_bit_exact(lpamd, monkeypatch):
    """The configuration of the headline benchmark at its own size: two full passes of the default
    chained loop (44 pivots at 22 per pass) against the oracle (reference lp_simplex_tableau.cc:10-52, 188-227), every
    entry of the 25.8 GB tableau included; then the same pivots with one pass per pivot."""
    m, n, seed = 32768, 65536, 1
    N = n + m
    eng = lpamd.TableauEngine(m, N)          # default knobs
    k = 2 * eng.loop_depth(lpamd.MAX_COST)
    eng.fill_synthetic(n, seed)
    assert eng.run_pivots(lpamd.MAX_COST, k) == (lpamd.LIMIT, k)
    pivots, streamed, touched = eng.pass_stats()
    assert pivots == k // 2 >= 16, "the default loop is expected to chain pivots at this size"
    dt = lo.DenseTableau.synthetic(m, n, seed)
    dt.s.skip_zero_cols = 1
    dt.s.fast_basis = 1
    assert dt.solve(lo.MAX_COST, max_pivots=k) == lo.LIMIT
    trace = eng.trace()
    assert trace == dt.trace()
    # the whole tableau, half of the columns at a time (bounded host memory)
    obj = eng.download_objective()
    assert np.array_equal(bits(obj), bits(dt.obj))
    basic = eng.download_basic()
    assert np.array_equal(basic[: dt.N], dt.is_basic[: dt.N])
    T = eng.download()[0]
    assert np.array_equal(bits(T), bits(dt.T)), "tableau differs from the oracle"
    del T
    gc.collect()
    # one pass per pivot: same trace, same entries on sampled columns (+ every entering column, RHS)
    monkeypatch.setenv("LPAMD_CHAIN", "1")
    one = lpamd.TableauEngine(m, N)
    monkeypatch.delenv("LPAMD_CHAIN")
    one.fill_synthetic(n, seed)
    assert one.run_pivots(lpamd.MAX_COST, k) == (lpamd.LIMIT, k)
    assert one.pass_stats()[0] == 1
    assert one.trace() == trace
    assert np.array_equal(bits(one.download_objective()), bits(dt.obj))
    rng = np.random.default_rng(5)
    for j in _sample_columns(rng, n, N, trace, 64):
        a, b = one.read_column(j), eng.read_column(j)
        assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(a), bits(dt.T[:, j])), "column %d" % j
    eng.close()
    one.close()


def test_m65536_n262144_max_reduction_properties_and_fused_pricing(lpamd, monkeypatch):
    """BASELINE's largest configuration on one GPU (172 GB stored): max-reduction pricing (reference
    lp_simplex_tableau.cc:235-374) from bounds (the default) against the full pricing fused into the update, the
    size-independent properties of the tableau after the pivots, and the storage-free oracle (pivot
    sequence, objective row, right-hand side, sampled columns bit for bit)."""
    m, n, seed, k = 65536, 262144, 1, 4
    N = n + m
    rng = np.random.default_rng(1)

    def run(bounds):
        # bounds: the default loop (pricing from bounds, a chain of pivots per pass); otherwise one pivot per pass with
        # the pricing of every column fused into the update
        monkeypatch.setenv("LPAMD_PR_BOUNDS", "1" if bounds else "0")
        eng = lpamd.TableauEngine(m, N)
        monkeypatch.delenv("LPAMD_PR_BOUNDS")
        assert (eng.loop_depth(lpamd.MAX_REDUCTION) > 1) == bounds
        eng.fill_synthetic(n, seed)
        for j in list(rng.integers(0, n, 2)) + [n + 7, N]:       # generator spot checks against the oracle's entry function
            col = eng.read_column(int(j))
            for i in rng.integers(0, m, 6):
                assert col[i] == lo.lib().lpo_synth_entry(seed, m, n, int(i), int(j))
        objs = [eng.objective_constant()]
        for leg in ((3, 1) if bounds else (1, 1, 1, 1)):       # the chained loop applies the first three pivots in one pass
            assert eng.run_pivots(lpamd.MAX_REDUCTION, leg) == (lpamd.LIMIT, leg)
            objs.append(eng.objective_constant())
        assert all(b > a for a, b in zip(objs, objs[1:])), "objective must increase on this non-degenerate LP"
        trace = eng.trace()
        rhs = eng.read_column(N)
        assert (rhs.astype(np.float32) > np.float32(-1e-6)).all(), "primal feasibility lost"
        basis = eng.basis_of_row()
        for e, r in trace:
            assert 0 <= e < N and 0 <= r < m
            if basis[r] != e:
                continue
            col = eng.read_column(e)          # a basic column is an exact unit column
            assert list(np.nonzero(col)[0]) == [r] and abs(col[r] + 1.0) < 1e-9
        cols = {j: eng.read_column(j) for j in _sample_columns(np.random.default_rng(9), n, N, trace, 24)}
        obj = eng.download_objective()
        eng.close()
        del eng
        gc.collect()
        return trace, objs, rhs, obj, cols

    fused = run(True)
    plain = run(False)
    assert fused[0] == plain[0], "pricing from bounds changed the pivot sequence"
    assert fused[1][-1] == plain[1][-1]
    assert np.array_equal(bits(fused[2]), bits(plain[2])) and np.array_equal(bits(fused[3]), bits(plain[3]))
    for j in fused[4]:
        assert np.array_equal(bits(fused[4][j]), bits(plain[4][j])), "column %d" % j
    # the oracle without a tableau in memory (entries regenerated from the seed and the pivots so far):
    # pivot sequence, objective row, right-hand side and the sampled columns bit for bit
    st = lo.StreamedSynthetic(m, n, seed, kmax=k)
    assert st.run(lo.MAX_REDUCTION, k) == lo.LIMIT
    trace, obj, rhs, basic = st.state()
    assert fused[0] == trace
    assert np.array_equal(bits(fused[2]), bits(rhs)) and np.array_equal(bits(fused[3]), bits(obj))
    for j in fused[4]:
        assert np.array_equal(bits(fused[4][j]), bits(st.column(j))), "column %d" % j
    st.close()


def test_m32768_n65536_as_eight_shards(lpamd):
    """BASELINE's sharded configuration at its own size: the 25.8 GB LP as 8 column shards (8 x 3.2 GB) behind
    the library's own loop (lpamd_group_*), here on one GPU with the in-process exchange: two full
    chained passes; every shard walks the single-engine pivots, the RHS replicas stay bit-identical, and
    sampled columns, objective row and basis flags equal the single-engine loop's."""
    m, n, seed = 32768, 65536, 1
    N = n + m
    g = lpamd.ShardGroup(m, N, 8, devices=[0] * 8, local_exchange=True)
    assert g.chain() >= 16
    k = 2 * g.chain()
    g.fill_synthetic(n, seed)
    assert g.run_pivots(lpamd.MAX_COST, k) == (lpamd.LIMIT, k)
    one = lpamd.TableauEngine(m, N)
    one.fill_synthetic(n, seed)
    assert one.run_pivots(lpamd.MAX_COST, k) == (lpamd.LIMIT, k)
    trace = one.trace()
    rhs = one.read_column(N)
    engines = [g.engine(q) for q in range(8)]
    for q, eng in enumerate(engines):
        assert eng.trace() == trace, "shard %d walked other pivots" % q
        assert np.array_equal(bits(eng.read_column(N)), bits(rhs)), "RHS replica of shard %d diverged" % q
    _, obj, basic = g.download(tableau=False)
    assert np.array_equal(bits(obj[: N + 1]), bits(one.download_objective()[: N + 1]))
    assert np.array_equal(basic[:N], one.download_basic()[:N])
    rng = np.random.default_rng(3)
    for j in _sample_columns(rng, n, N, trace, 64):
        if j == N:
            continue
        assert np.array_equal(bits(engines[j % 8].read_column(j)), bits(one.read_column(j))), "column %d" % j
    g.close()
    one.close()


def test_m65536_n262144_max_reduction_as_eight_shards(lpamd):
    """BASELINE's largest configuration in its sharded form at its own size: 8 column shards of 21.5 GB each
    (here on one GPU, in-process exchange) behind lpamd_group_*, max-reduction pricing from bounds with a chain
    of pivots per pass, against the storage-free oracle: pivot sequence on every shard, right-hand-side
    replicas, objective row and sampled columns bit for bit (reference lp_simplex_tableau.cc:235-374, 10-52)."""
    m, n, seed, k = 65536, 262144, 1, 4
    N = n + m
    g = lpamd.ShardGroup(m, N, 8, devices=[0] * 8, local_exchange=True)
    assert g.chain() > 1
    g.fill_synthetic(n, seed)
    for leg in (3, 1):                        # a pass of three chained pivots, then one more
        assert g.run_pivots(lpamd.MAX_REDUCTION, leg) == (lpamd.LIMIT, leg)
    st = lo.StreamedSynthetic(m, n, seed, kmax=k)
    assert st.run(lo.MAX_REDUCTION, k) == lo.LIMIT
    trace, obj, rhs, basic = st.state()
    engines = [g.engine(q) for q in range(8)]
    for q, eng in enumerate(engines):
        assert eng.trace() == trace, "shard %d walked other pivots" % q
        assert np.array_equal(bits(eng.read_column(N)), bits(rhs)), "RHS replica of shard %d differs from the oracle" % q
    _, gobj, gbasic = g.download(tableau=False)
    assert np.array_equal(bits(gobj[: N + 1]), bits(obj[: N + 1]))
    assert np.array_equal(gbasic[:N], basic[:N])
    for j in _sample_columns(np.random.default_rng(9), n, N, trace, 24):
        if j == N:
            continue
        assert np.array_equal(bits(engines[j % 8].read_column(j)), bits(st.column(j))), "column %d" % j
    st.close()
    g.close()
