"""TEST INFRASTRUCTURE.  Pins the CPU oracle: replays every tableau-path golden
the reference ships (tests/sol*.txt per CMakeLists.txt:85-103, netlib goldens
per CMakeLists.txt:122-136) plus the lp_test.cc tableau string snapshots, and
compares the oracle's output text byte-for-byte.

Run: python -m oracle.pin_oracle [--fast]   (from the repo root)
"""
import os
import sys
import time

from . import lp_frontend as fe
from . import lp_oracle as lo
from . import mps_to_text as mt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def lp_cases():
    for case in (12, 13, 14, 15, 16, 17):
        for strat, suffix in (("max_cost", "_simplex_tableau"), ("max_reduction", "_simplex_tableau_max_reduction")):
            sol = os.path.join(G, "lp", "sol%d%s.txt" % (case, suffix))
            if not os.path.exists(sol):
                sol = os.path.join(G, "lp", "sol%d.txt" % case)
            yield "test%d" % case, strat, fe.read_text(os.path.join(G, "lp", "test%d.txt" % case)), sol


def netlib_cases(names=None):
    for strat, d in (("max_cost", "netlib_sol_max_cost"), ("max_reduction", "netlib_sol_max_reduction")):
        for fn in sorted(os.listdir(os.path.join(G, d))):
            name = fn[:-4]
            if names is not None and name not in names:
                continue
            text = mt.mps_to_text(os.path.join(G, "netlib_mps", name + ".mps"))
            yield name, strat, text, os.path.join(G, d, fn)


def main():
    fast = "--fast" in sys.argv
    bad = 0
    cases = list(lp_cases()) + list(netlib_cases({"afiro", "adlittle", "kb2"} if fast else None))
    for name, strat, text, sol in cases:
        t0 = time.time()
        out, dt = lo.solve_text(text, strat)
        want = open(sol).read()
        ok = out == want or out.rstrip("\n") == want.rstrip("\n")
        bad += not ok
        print("%-9s %-13s m=%-5d W=%-5d pivots=%-5d %s  (%.1fs)" % (
            name, strat, dt.m, dt.ncols - 1, dt.n_pivots, "EXACT" if ok else "MISMATCH", time.time() - t0))
        if not ok:
            a, b = out.splitlines(), want.splitlines()
            for k in range(min(len(a), len(b))):
                if a[k] != b[k]:
                    print("   first diff line %d: got %r want %r" % (k, a[k], b[k]))
                    break
    print("mismatches:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
