"""Times the drop-in `solver` binary on the reference's own LP files (tests/ and netlib, the
decoded inputs under tests/golden): wall time of the solve call, pivots, and whether stdout
equals the reference's golden output byte for byte.  One JSON line per case.

    python scripts/netlib_bench.py [--repeat 3]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "linear-programming_amd", "bin", "solver")

TABLEAU = [("test%d" % c, s) for c in (12, 13, 14, 15, 16, 17) for s in ("max_cost", "max_reduction")]
TABLEAU += [(n, "max_cost") for n in ("afiro", "adlittle", "kb2", "israel", "agg", "agg2", "agg3", "fit1d")]
TABLEAU += [(n, "max_reduction") for n in ("afiro", "adlittle", "kb2", "israel", "agg", "agg2", "agg3")]
REVISED = [("test12", "lp/sol12.txt"), ("test13", "lp/sol13.txt"), ("test14", "lp/sol14.txt"),
           ("test15", "lp/sol15_revised_simplex_tableau.txt"), ("test16", "lp/sol16_revised_simplex_tableau.txt"),
           ("test17", "lp/sol17_revised_simplex_tableau.txt"), ("afiro", "netlib_sol_revised/afiro.txt"),
           ("kb2", "netlib_sol_revised/kb2.txt"), ("adlittle", "netlib_sol_revised/adlittle.txt"),
           ("israel", "netlib_sol_revised/israel.txt")]


def path_of(name):
    if name.startswith("test"):
        return os.path.join(GOLDEN, "lp", name + ".txt")
    return os.path.join(GOLDEN, "netlib_mps", name + ".mps")


def tableau_golden(name, strategy):
    if name.startswith("test"):
        suffix = "_simplex_tableau" if strategy == "max_cost" else "_simplex_tableau_max_reduction"
        p = os.path.join(GOLDEN, "lp", "sol%s%s.txt" % (name[4:], suffix))
        if not os.path.exists(p):
            p = os.path.join(GOLDEN, "lp", "sol%s.txt" % name[4:])
        return p
    d = "netlib_sol_max_cost" if strategy == "max_cost" else "netlib_sol_max_reduction"
    return os.path.join(GOLDEN, d, name + ".txt")


def run(name, algo, strategy, golden, repeat):
    best, pivots, same = None, None, True
    for _ in range(repeat):
        out = subprocess.run([BIN, path_of(name), algo, strategy, "--time"], capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            return {"case": name, "algo": algo, "strategy": strategy, "error": out.stderr[-200:]}
        same &= out.stdout.rstrip("\n") == open(golden).read().rstrip("\n")
        for line in out.stderr.splitlines():
            if line.startswith("solve_ms"):
                parts = line.split()
                ms, pivots = float(parts[1]), int(parts[3])
                best = ms if best is None else min(best, ms)
    return {"case": name, "algo": algo, "strategy": strategy, "pivots": pivots, "solve_ms": round(best, 3),
            "pivots_per_s": round(pivots / (best * 1e-3), 1) if best else None, "stdout_equals_reference_golden": same}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    for name, strategy in TABLEAU:
        print(json.dumps(run(name, "simplex_tableau", strategy, tableau_golden(name, strategy), a.repeat)), flush=True)
    for name, rel in REVISED:
        print(json.dumps(run(name, "revised_simplex_tableau", "max_cost", os.path.join(GOLDEN, rel), a.repeat)), flush=True)


if __name__ == "__main__":
    sys.exit(main())
