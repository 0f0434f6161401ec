"""Shared case lists for the parity tests (inputs + the reference golden each maps to)."""
import os

from oracle import lp_frontend as fe
from oracle import mps_to_text as mt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

LP_CASES = [12, 13, 14, 15, 16, 17]
NETLIB_MAX_COST = ["afiro", "adlittle", "kb2", "israel", "agg", "agg2", "agg3", "fit1d"]
NETLIB_MAX_REDUCTION = ["afiro", "adlittle", "kb2", "israel", "agg", "agg2", "agg3"]


def lp_text(case):
    return fe.read_text(os.path.join(GOLDEN, "lp", "test%d.txt" % case))


def lp_golden(case, strategy):
    suffix = "_simplex_tableau" if strategy == "max_cost" else "_simplex_tableau_max_reduction"
    p = os.path.join(GOLDEN, "lp", "sol%d%s.txt" % (case, suffix))
    if not os.path.exists(p):
        p = os.path.join(GOLDEN, "lp", "sol%d.txt" % case)  # reference CMakeLists.txt:85-103
    return open(p).read()


def netlib_text(name):
    return mt.mps_to_text(os.path.join(GOLDEN, "netlib_mps", name + ".mps"))


def netlib_golden(name, strategy):
    d = "netlib_sol_max_cost" if strategy == "max_cost" else "netlib_sol_max_reduction"
    return open(os.path.join(GOLDEN, d, name + ".txt")).read()


def all_cases(netlib=True):
    out = []
    for c in LP_CASES:
        for s in ("max_cost", "max_reduction"):
            out.append(("test%d" % c, s))
    if netlib:
        out += [(n, "max_cost") for n in NETLIB_MAX_COST]
        out += [(n, "max_reduction") for n in NETLIB_MAX_REDUCTION]
    return out


def case_text(name):
    return lp_text(int(name[4:])) if name.startswith("test") else netlib_text(name)


def case_golden(name, strategy):
    return lp_golden(int(name[4:]), strategy) if name.startswith("test") else netlib_golden(name, strategy)


# revised simplex on the tableau: reference CMakeLists.txt:105-114 and 138-144
REVISED_CASES = [("test12", "lp/sol12.txt"), ("test13", "lp/sol13.txt"), ("test14", "lp/sol14.txt"),
                 ("test15", "lp/sol15_revised_simplex_tableau.txt"), ("test16", "lp/sol16_revised_simplex_tableau.txt"),
                 ("test17", "lp/sol17_revised_simplex_tableau.txt"), ("afiro", "netlib_sol_revised/afiro.txt"),
                 ("kb2", "netlib_sol_revised/kb2.txt"), ("adlittle", "netlib_sol_revised/adlittle.txt"),
                 ("israel", "netlib_sol_revised/israel.txt")]


def revised_golden(rel):
    return open(os.path.join(GOLDEN, rel)).read()
