"""TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of oracle/liblporacle.so (the C restatement in
lp_oracle_core.c) plus the text-in / text-out convenience used by the golden
checks.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.
"""
import ctypes
import os

import numpy as np

from . import lp_frontend as fe

ERROR, UNBOUNDED, NOSOLUTION, SOLVED, LIMIT = 0, 1, 2, 3, 4
MAX_COST, MAX_REDUCTION = 2, 3
STRATEGY = {"max_cost": MAX_COST, "max_reduction": MAX_REDUCTION}
RESULT_TEXT = {NOSOLUTION: "No Solution\n", UNBOUNDED: "Unbounded\n", ERROR: "Error\n"}

_HERE = os.path.dirname(os.path.abspath(__file__))


class _LpoT(ctypes.Structure):
    _fields_ = [
        ("m", ctypes.c_int64), ("ncols", ctypes.c_int64), ("lda", ctypes.c_int64), ("N", ctypes.c_int64),
        ("T", ctypes.c_void_p), ("obj", ctypes.c_void_p), ("is_basic", ctypes.c_void_p),
        ("skip_zero_cols", ctypes.c_int32), ("fast_basis", ctypes.c_int32),
        ("basis_of_row", ctypes.c_void_p),
        ("trace_e", ctypes.c_void_p), ("trace_r", ctypes.c_void_p), ("trace_d", ctypes.c_void_p),
        ("trace_cap", ctypes.c_int64), ("n_pivots", ctypes.c_int64), ("max_pivots", ctypes.c_int64),
        ("update_seconds", ctypes.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liblporacle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liblporacle.so missing: run `make -C oracle`")
        L = ctypes.CDLL(path)
        P = ctypes.POINTER(_LpoT)
        i64 = ctypes.c_int64
        L.lpo_pivot.argtypes = [P, i64, i64, i64]
        L.lpo_pivot.restype = ctypes.c_int
        L.lpo_need_init.argtypes = [P]
        L.lpo_price_max_cost.argtypes = [P]
        L.lpo_price_max_cost.restype = i64
        L.lpo_ratio_test.argtypes = [P, i64, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.lpo_ratio_test.restype = None
        L.lpo_price_max_reduction.argtypes = [P, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.lpo_price_max_reduction.restype = None
        L.lpo_phase1.argtypes = [P, ctypes.c_int]
        L.lpo_solve.argtypes = [P, ctypes.c_int]
        L.lpo_column_values.argtypes = [P, ctypes.c_void_p]
        L.lpo_column_values.restype = None
        L.lpo_synth_fill.argtypes = [P, ctypes.c_uint64, i64]
        L.lpo_synth_fill.restype = None
        L.lpo_synth_entry.argtypes = [ctypes.c_uint64, i64, i64, i64, i64]
        L.lpo_synth_entry.restype = ctypes.c_double
        L.lpo_num_threads.restype = ctypes.c_int
        vp = ctypes.c_void_p
        L.lpo_stream_create.argtypes = [ctypes.c_uint64, i64, i64, i64]
        L.lpo_stream_create.restype = vp
        L.lpo_stream_free.argtypes = [vp]
        L.lpo_stream_free.restype = None
        L.lpo_stream_run.argtypes = [vp, ctypes.c_int, i64]
        L.lpo_stream_column.argtypes = [vp, i64, vp]
        L.lpo_stream_column.restype = None
        L.lpo_stream_state.argtypes = [vp, vp, vp, vp, vp, vp]
        L.lpo_stream_state.restype = i64
        _lib = L
    return _lib


class DenseTableau:
    """Column-major dense tableau driven by the C core."""

    def __init__(self, T, obj, is_basic, N, trace_cap=1 << 16, basis_of_row=None):
        T = np.array(T, dtype=np.float64, order="F", copy=True)   # never alias the caller's matrix
        self.T = T
        self.m, self.ncols = T.shape
        assert self.ncols == N + 2
        self.N = N
        self.obj = np.ascontiguousarray(obj, dtype=np.float64).copy()
        self.is_basic = np.ascontiguousarray(is_basic, dtype=np.uint8).copy()
        self.trace_e = np.full(trace_cap, -1, dtype=np.int64)
        self.trace_r = np.full(trace_cap, -1, dtype=np.int64)
        self.trace_d = np.full(trace_cap, -1, dtype=np.int64)
        self.basis_of_row = None if basis_of_row is None else np.ascontiguousarray(basis_of_row, dtype=np.int64).copy()
        self.s = _LpoT()
        s = self.s
        s.m, s.ncols, s.lda, s.N = self.m, self.ncols, max(self.m, 1), N
        s.T = T.ctypes.data
        s.obj = self.obj.ctypes.data
        s.is_basic = self.is_basic.ctypes.data
        s.skip_zero_cols = 0
        s.fast_basis = 0
        s.basis_of_row = None if self.basis_of_row is None else self.basis_of_row.ctypes.data
        s.trace_e, s.trace_r, s.trace_d = self.trace_e.ctypes.data, self.trace_r.ctypes.data, self.trace_d.ctypes.data
        s.trace_cap = trace_cap
        s.n_pivots = 0
        s.max_pivots = -1
        s.update_seconds = 0.0

    @classmethod
    def synthetic(cls, m, n, seed, trace_cap=1 << 16):
        N = n + m
        T = np.zeros((m, N + 2), dtype=np.float64, order="F")
        t = cls(T, np.zeros(N + 2), np.zeros(N + 2, dtype=np.uint8), N, trace_cap, basis_of_row=np.zeros(m, dtype=np.int64))
        lib().lpo_synth_fill(ctypes.byref(t.s), seed, n)
        return t

    @property
    def n_pivots(self):
        return int(self.s.n_pivots)

    def trace(self):
        k = min(self.n_pivots, len(self.trace_e))
        return list(zip(self.trace_e[:k].tolist(), self.trace_r[:k].tolist()))

    def pivot(self, d, e, r):
        return lib().lpo_pivot(ctypes.byref(self.s), d, e, r)

    def need_init(self):
        return bool(lib().lpo_need_init(ctypes.byref(self.s)))

    def price_max_cost(self):
        return int(lib().lpo_price_max_cost(ctypes.byref(self.s)))

    def ratio_test(self, e):
        r, d = ctypes.c_int64(), ctypes.c_int64()
        lib().lpo_ratio_test(ctypes.byref(self.s), e, ctypes.byref(r), ctypes.byref(d))
        return r.value, d.value

    def price_max_reduction(self):
        e, r, d = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        lib().lpo_price_max_reduction(ctypes.byref(self.s), ctypes.byref(e), ctypes.byref(r), ctypes.byref(d))
        return e.value, r.value, d.value

    def phase1(self, strategy):
        return lib().lpo_phase1(ctypes.byref(self.s), strategy)

    def solve(self, strategy, max_pivots=-1):
        self.s.max_pivots = max_pivots
        return lib().lpo_solve(ctypes.byref(self.s), strategy)

    def column_values(self):
        x = np.zeros(self.ncols, dtype=np.float64)
        lib().lpo_column_values(ctypes.byref(self.s), x.ctypes.data)
        return x


class StreamedSynthetic:
    """The synthetic LP without a tableau in memory: every entry is regenerated from (seed, i, j) and
    the pivots so far (see lp_oracle_core.c); for the benchmark sizes the host cannot store."""

    def __init__(self, m, n, seed, kmax=64):
        self.m, self.n, self.N, self.ncols, self.kmax = m, n, n + m, n + m + 2, kmax
        self.h = lib().lpo_stream_create(seed, m, n, kmax)
        if not self.h:
            raise MemoryError("lpo_stream_create")

    def run(self, strategy, budget):
        return lib().lpo_stream_run(self.h, strategy, budget)

    def column(self, j):
        out = np.zeros(self.m, dtype=np.float64)
        lib().lpo_stream_column(self.h, int(j), out.ctypes.data)
        return out

    def state(self):
        """(trace, objective row, right-hand side, basis flags)"""
        te, tr = np.zeros(self.kmax, dtype=np.int64), np.zeros(self.kmax, dtype=np.int64)
        obj, rhs, basic = np.zeros(self.ncols), np.zeros(self.m), np.zeros(self.ncols, dtype=np.uint8)
        k = lib().lpo_stream_state(self.h, te.ctypes.data, tr.ctypes.data, obj.ctypes.data, rhs.ctypes.data, basic.ctypes.data)
        return list(zip(te[:k].tolist(), tr[:k].tolist())), obj, rhs, basic

    def close(self):
        if self.h:
            lib().lpo_stream_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def from_text(text, **kw):
    tab = fe.build_tableau(text)
    return tab, DenseTableau(tab.T, tab.obj, tab.is_basic, tab.N, **kw)


def solve_text(text, strategy="max_cost"):
    """What `solver <file> simplex_tableau <strategy>` prints (solver.cc:123-171)."""
    tab, dt = from_text(text)
    res = dt.solve(STRATEGY[strategy])
    if res == SOLVED:
        return fe.format_solution(tab, dt.column_values(), dt.obj[tab.N]), dt
    return RESULT_TEXT.get(res, "Error\n"), dt


# ---------------------------------------------------------------- revised simplex
class _LpoRevT(ctypes.Structure):
    _fields_ = [
        ("m", ctypes.c_int64), ("N", ctypes.c_int64), ("ncols", ctypes.c_int64), ("lda", ctypes.c_int64),
        ("T", ctypes.c_void_p), ("obj", ctypes.c_void_p), ("is_basic", ctypes.c_void_p), ("basis", ctypes.c_void_p),
        ("Binv", ctypes.c_void_p), ("xB", ctypes.c_void_p), ("cB", ctypes.c_void_p),
        ("opt_reverted", ctypes.c_int32), ("optimum", ctypes.c_double),
        ("trace_e", ctypes.c_void_p), ("trace_l", ctypes.c_void_p),
        ("trace_cap", ctypes.c_int64), ("n_pivots", ctypes.c_int64), ("max_pivots", ctypes.c_int64),
    ]


class RevisedTableau:
    """Dense state of the revised simplex on the tableau (reference
    lp_revised_simplex_tableau.cc), driven by lpo_revised_solve."""

    def __init__(self, T, obj, is_basic, N, opt_reverted, trace_cap=1 << 16):
        m = T.shape[0]
        self.m, self.N = m, N
        W = N + 1 + m
        self.T = np.zeros((m, W), dtype=np.float64, order="F")
        self.T[:, : N + 1] = T[:, : N + 1]
        self.obj = np.zeros(W, dtype=np.float64)
        self.obj[: N + 1] = obj[: N + 1]
        self.is_basic = np.zeros(W, dtype=np.uint8)
        self.is_basic[:N] = is_basic[:N]
        self.basis = np.array([j for j in range(N) if is_basic[j]], dtype=np.int64)   # Variable order == column order
        assert len(self.basis) == m
        self.Binv = np.zeros((m, m), dtype=np.float64)
        self.xB = np.zeros(m, dtype=np.float64)
        self.cB = np.zeros(m, dtype=np.float64)
        self.trace_e = np.full(trace_cap, -1, dtype=np.int64)
        self.trace_l = np.full(trace_cap, -1, dtype=np.int64)
        s = self.s = _LpoRevT()
        s.m, s.N, s.ncols, s.lda = m, N, W, max(m, 1)
        s.T, s.obj, s.is_basic, s.basis = self.T.ctypes.data, self.obj.ctypes.data, self.is_basic.ctypes.data, self.basis.ctypes.data
        s.Binv, s.xB, s.cB = self.Binv.ctypes.data, self.xB.ctypes.data, self.cB.ctypes.data
        s.opt_reverted = 1 if opt_reverted else 0
        s.optimum = 0.0
        s.trace_e, s.trace_l = self.trace_e.ctypes.data, self.trace_l.ctypes.data
        s.trace_cap, s.n_pivots, s.max_pivots = trace_cap, 0, -1
        L = lib()
        L.lpo_revised_solve.argtypes = [ctypes.POINTER(_LpoRevT)]
        L.lpo_revised_solve.restype = ctypes.c_int

    def solve(self, max_pivots=-1):
        self.s.max_pivots = max_pivots
        return lib().lpo_revised_solve(ctypes.byref(self.s))

    @property
    def n_pivots(self):
        return int(self.s.n_pivots)

    @property
    def optimum(self):
        return float(self.s.optimum)

    def trace(self):
        k = min(self.n_pivots, len(self.trace_e))
        return list(zip(self.trace_e[:k].tolist(), self.trace_l[:k].tolist()))

    def column_values(self):
        x = np.zeros(self.N + 2, dtype=np.float64)
        for i in range(self.m):
            if self.basis[i] < self.N:
                x[self.basis[i]] = self.xB[i]
        return x


def revised_from_text(text, **kw):
    tab = fe.build_tableau(text)
    return tab, RevisedTableau(tab.T, tab.obj, tab.is_basic, tab.N, tab.opt_reverted, **kw)


def format_revised(tab, rt):
    """stdout of `solver <file> revised_simplex_tableau` (solver.cc:135-171): the optimum
    already carries its sign (lp_revised_simplex_tableau.cc:253-257)."""
    import copy
    t2 = copy.copy(tab)
    t2.opt_reverted = False
    return fe.format_solution(t2, rt.column_values(), rt.optimum)


def solve_text_revised(text):
    tab, rt = revised_from_text(text)
    res = rt.solve()
    if res == SOLVED:
        return format_revised(tab, rt), rt
    return RESULT_TEXT.get(res, "Error\n"), rt
