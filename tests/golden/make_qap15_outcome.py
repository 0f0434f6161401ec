#!/usr/bin/env python3
"""Writes tests/golden/plato/qap15_oracle_outcome_<strategy>.json: what the CPU oracle (plain dense restatement, no
shortcuts) does on datasets/plato/qap15.mps with `simplex_tableau <strategy>` within a pivot budget -- result, pivot
count, a digest of the pivot sequence, its first and last pivots, the objective constant.  The reference ships the
input but no golden output for it, and the oracle manages a few pivots per second on this instance (12660 x 34937
tableau, 3.5 GB; the whole solve takes the engine several hundred thousand pivots), so a prefix is computed once here
and the GPU test compares its own first pivots with the file.

  python tests/golden/make_qap15_outcome.py [max_cost|max_reduction] [pivots]
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import lp_oracle as lo   # noqa: E402
from oracle import mps_to_text as mt  # noqa: E402

strategy = sys.argv[1] if len(sys.argv) > 1 else "max_cost"
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
t0 = time.time()
text = mt.mps_to_text(os.path.join(HERE, "plato", "qap15.mps"))
tab, dt = lo.from_text(text, trace_cap=1 << 22)
print("tableau %d x %d built in %.0f s" % (tab.m, tab.N + 2, time.time() - t0), flush=True)
t0 = time.time()
res = dt.solve(lo.STRATEGY[strategy], max_pivots=budget)
trace = dt.trace()
h = hashlib.sha256()
for e, r in trace:
    h.update(b"%d %d\n" % (e, r))
out = {"input": "datasets/plato/qap15.mps", "strategy": strategy, "pivot_budget": budget, "m": tab.m, "N": tab.N, "result": int(res), "n_pivots": dt.n_pivots,
       "trace_sha256": h.hexdigest(), "first_pivots": trace[:16], "last_pivots": trace[-16:],
       "objective_constant_bits": "%016x" % int(dt.obj[tab.N:tab.N + 1].view("uint64")[0]),
       "oracle_seconds": round(time.time() - t0)}
path = os.path.join(HERE, "plato", "qap15_oracle_outcome_%s.json" % strategy)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out)[:600], flush=True)
