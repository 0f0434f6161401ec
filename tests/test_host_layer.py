"""CPU: the C++ host layer (parser, normal-form transforms, dense tableau build,
MPS conversion) against the reference's unit-test strings and against the
oracle's independently built tableau."""
import os
import subprocess

import numpy as np
import pytest

from cases import GOLDEN, ROOT, all_cases, case_text
from oracle import lp_frontend as fe
from oracle import mps_to_text as mt

BIN = os.path.join(ROOT, "linear-programming_amd", "bin")
NAMES = sorted({n for n, _ in all_cases()})


def _path(name):
    if name.startswith("test"):
        return os.path.join(GOLDEN, "lp", name + ".txt")
    return os.path.join(GOLDEN, "netlib_mps", name + ".mps")


def test_selftest_binary_host_cases():
    out = subprocess.run([os.path.join(BIN, "lp_host_selftest"), os.path.join(GOLDEN, "lp")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failed" in out.stdout


@pytest.mark.parametrize("name", NAMES)
def test_host_tableau_is_bit_identical_to_oracle_frontend(name):
    out = subprocess.check_output([os.path.join(BIN, "lp_host_probe"), "tableau", _path(name)]).decode().split("\n")
    m, N = map(int, out[0].split())
    tab = fe.build_tableau(case_text(name))
    assert (m, N) == (tab.m, tab.N)
    T = np.array([[int(x, 16) for x in out[1 + j].split()] for j in range(N + 2)], dtype=np.uint64).T
    obj = np.array([int(x, 16) for x in out[1 + N + 2].split()], dtype=np.uint64)
    basic = np.array([int(x) for x in out[2 + N + 2].split()], dtype=np.uint8)
    assert np.array_equal(T, tab.T.view(np.uint64))
    assert np.array_equal(obj, tab.obj.view(np.uint64))
    assert np.array_equal(basic[:N], tab.is_basic[:N])


@pytest.mark.parametrize("name", [n for n in NAMES if not n.startswith("test")])
def test_mps_conversion_matches(name):
    got = subprocess.check_output([os.path.join(BIN, "lp_host_probe"), "mps2txt", _path(name)]).decode()
    assert got == mt.mps_to_text(_path(name))


def test_kb2_bounds_become_rows():
    # kb2 has 9 UP bounds; they are written as `x<=ub` lines in name-sorted order of
    # the original MPS column names and each becomes a tableau row
    text = mt.mps_to_text(_path("kb2"))
    assert sum(1 for l in text.split("\n") if "<=" in l and "*" not in l) == 9


def test_solver_cli_rejects_other_algorithms():
    out = subprocess.run([os.path.join(BIN, "solver"), _path("test12"), "simplex"], capture_output=True, text=True)
    assert out.returncode == 2 and out.stdout == "Error\n"
