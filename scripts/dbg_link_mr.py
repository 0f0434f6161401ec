import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
lp = importlib.import_module("linear-programming_amd")
from oracle import lp_oracle as lo
m, n, seed = 700, 1300, 11
cases = [(s, w) for s in ("max_cost", "max_reduction") for w in (1, 2, 3, 8)]
if len(sys.argv) > 1:
    cases = [tuple(a.split(":")) for a in sys.argv[1:]]
for case in cases:
    strategy, world = case[0], int(case[1])
    if len(case) > 2: os.environ["LPAMD_SHARD_LINK"] = case[2]
    g = lp.ShardGroup(m, n + m, world, devices=[0] * world, local_exchange=True)
    g.fill_synthetic(n, seed)
    dt = lo.DenseTableau.synthetic(m, n, seed)
    want = dt.solve(lo.STRATEGY[strategy])
    try:
        res, piv = g.solve(lp.STRATEGY[strategy])
        t, w = g.trace(), dt.trace()
        i = next((q for q in range(min(len(t), len(w))) if t[q] != w[q]), None)
        print(strategy, world, "link", g.link_active(), "result", res, want, "pivots", piv, dt.n_pivots, "first difference", i)
    except Exception as ex:
        t, w = g.engine(0).trace(), dt.trace()
        i = next((q for q in range(min(len(t), len(w))) if t[q] != w[q]), None)
        print(strategy, world, "link", g.link_active(), "error", ex, "| trace", len(t), "oracle", len(w), "first difference", i)
    g.close()
