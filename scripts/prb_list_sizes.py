"""How many candidate columns the MAX_REDUCTION pricing from bounds evaluates exactly, per pricing."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lp = importlib.import_module("linear-programming_amd")
m, n = (int(x) for x in sys.argv[1:3])
legs = [int(x) for x in sys.argv[3:]] or [1] * 24
eng = lp.TableauEngine(m, n + m)
eng.fill_synthetic(n, 1)
prev = (0, 0)
out = []
for k in legs:
    eng.run_pivots(lp.MAX_REDUCTION, k)
    cur = eng.pricing_stats()
    out.append("%d:%d" % (cur[0] - prev[0], cur[1] - prev[1]))
    prev = cur
print("m=%d n=%d (pricings:columns evaluated) per leg:" % (m, n), " ".join(out))
