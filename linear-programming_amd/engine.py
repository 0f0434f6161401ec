"""ctypes binding of the lpamd C ABI (include/lpamd.h), one method per entry point.

Mirrors the tableau part of the reference's LPModel surface (reference
lp.h:64-106): ToTableau -> TableauEngine.upload, TableauPivot -> pivot,
TableauSimplexInitialize -> phase1, TableauSimplexSolve -> solve,
GetTableauSimplexOptimum/Solution -> objective_constant / column_values;
TableauRevisedSimplexSolve and its getters (reference lp.h:128-134) ->
RevisedTableauEngine.
"""
import ctypes
import os

import numpy as np

# reference lp.h:32-37 (enum Result) + LIMIT, lp.h:74-79 (PivotingStrategy)
ERROR, UNBOUNDED, NOSOLUTION, SOLVED, LIMIT = 0, 1, 2, 3, 4
MAX_COST, MAX_REDUCTION = 2, 3
STRATEGY = {"max_cost": MAX_COST, "max_reduction": MAX_REDUCTION}
CANDIDATE_BYTES = 32

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LpamdError(RuntimeError):
    pass


def library_path():
    # LPAMD_LIBRARY: another build of the same library (experiments: instrumented kernels)
    return os.environ.get("LPAMD_LIBRARY") or os.path.join(_HERE, "lib", "liblpamd.so")


def library():
    """Loads liblpamd.so; raises if it has not been built (no fallback path)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise LpamdError("%s is missing: build it with `make` (or __graft_entry__.build())" % path)
    L = ctypes.CDLL(path)
    i32, i64, u64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p
    P = ctypes.POINTER
    sig = {
        "lpamd_exchange_bytes": (i64, [i64]),
        "lpamd_last_error": (ctypes.c_char_p, []),
        "lpamd_version": (ctypes.c_char_p, []),
        "lpamd_create": (ctypes.c_int, [P(vp), i64, i64, i32, i32, vp]),
        "lpamd_destroy": (ctypes.c_int, [vp]),
        "lpamd_upload": (ctypes.c_int, [vp, vp, i64, vp, vp]),
        "lpamd_fill_synthetic": (ctypes.c_int, [vp, i64, u64]),
        "lpamd_download": (ctypes.c_int, [vp, vp, i64, vp, vp]),
        "lpamd_read_column": (ctypes.c_int, [vp, i64, vp]),
        "lpamd_read_row": (ctypes.c_int, [vp, i64, vp]),
        "lpamd_write_objective": (ctypes.c_int, [vp, vp]),
        "lpamd_basis_of_row": (ctypes.c_int, [vp, vp]),
        "lpamd_need_init": (ctypes.c_int, [vp, P(i32)]),
        "lpamd_price_max_cost": (ctypes.c_int, [vp, P(i64)]),
        "lpamd_ratio_test": (ctypes.c_int, [vp, i64, P(i64), P(i64)]),
        "lpamd_price_max_reduction": (ctypes.c_int, [vp, P(i64), P(i64), P(i64)]),
        "lpamd_pivot": (ctypes.c_int, [vp, i64, i64, i64]),
        "lpamd_phase1": (ctypes.c_int, [vp, i32, P(i32)]),
        "lpamd_solve": (ctypes.c_int, [vp, i32, i64, P(i32), P(i64)]),
        "lpamd_run_pivots": (ctypes.c_int, [vp, i32, i64, P(i32), P(i64)]),
        "lpamd_column_values": (ctypes.c_int, [vp, vp]),
        "lpamd_objective_constant": (ctypes.c_int, [vp, P(ctypes.c_double)]),
        "lpamd_pivot_count": (ctypes.c_int, [vp, P(i64)]),
        "lpamd_trace": (ctypes.c_int, [vp, vp, vp, i64, P(i64)]),
        "lpamd_set_profiling": (ctypes.c_int, [vp, i32]),
        "lpamd_update_stats": (ctypes.c_int, [vp, P(ctypes.c_double), P(i64), P(i64)]),
        "lpamd_pass_stats": (ctypes.c_int, [vp, P(ctypes.c_int32), P(i64), P(i64)]),
        "lpamd_profile_totals": (ctypes.c_int, [vp, P(ctypes.c_double), P(i64), P(i64), P(i64), P(i64)]),
        "lpamd_loop_depth": (ctypes.c_int, [vp, i32, P(i32)]),
        "lpamd_pricing_stats": (ctypes.c_int, [vp, P(i64), P(i64)]),
        "lpamd_local_candidate": (ctypes.c_int, [vp, i32, vp]),
        "lpamd_select_candidate": (ctypes.c_int, [vp, i32, vp, i32, P(i64), P(i64), P(i32), P(i32)]),
        "lpamd_fold_candidates": (ctypes.c_int, [i32, vp, i32, i64, P(i64), P(i64), P(i32), P(i32), vp]),
        "lpamd_export_column": (ctypes.c_int, [vp, i64, vp]),
        "lpamd_select_candidate_device": (ctypes.c_int, [vp, i32, vp, i32]),
        "lpamd_export_column_or_zero": (ctypes.c_int, [vp, vp]),
        "lpamd_set_pivot_budget": (ctypes.c_int, [vp, i64]),
        "lpamd_apply_pivot": (ctypes.c_int, [vp, i32, i64, i64, vp]),
        "lpamd_poll": (ctypes.c_int, [vp, P(i32)]),
        "lpamd_lookahead_setup": (ctypes.c_int, [vp, vp]),
        "lpamd_lookahead_export": (ctypes.c_int, [vp, vp, vp]),
        "lpamd_lookahead_update": (ctypes.c_int, [vp]),
        "lpamd_chain_depth": (ctypes.c_int, [vp, P(i32)]),
        "lpamd_chain_step": (ctypes.c_int, [vp, vp]),
        "lpamd_lookahead_select": (ctypes.c_int, [vp, vp, i32, vp, vp, vp]),
        "lpamd_fill_column": (ctypes.c_int, [vp, i64, ctypes.c_double]),
        "lpamd_set_basic_flag": (ctypes.c_int, [vp, i64, i32]),
        "lpamd_pivot_shard": (ctypes.c_int, [vp, i64, i64, i64, ctypes.c_double, vp]),
        "lpamd_comm_unique_id": (ctypes.c_int, [vp]),
        "lpamd_group_create": (ctypes.c_int, [P(vp), i64, i64, i32, vp, i32]),
        "lpamd_group_create_rank": (ctypes.c_int, [P(vp), i64, i64, i32, i32, vp]),
        "lpamd_group_create_rank_hosted": (ctypes.c_int, [P(vp), i64, i64, i32, i32, vp, vp]),
        "lpamd_group_destroy": (ctypes.c_int, [vp]),
        "lpamd_group_shards": (ctypes.c_int, [vp, P(i32), P(i32)]),
        "lpamd_group_comm_ranks": (ctypes.c_int, [vp, P(i32)]),
        "lpamd_group_link_active": (ctypes.c_int, [vp, P(i32)]),
        "lpamd_group_engine": (ctypes.c_int, [vp, i32, P(vp)]),
        "lpamd_group_chain": (ctypes.c_int, [vp, P(i32)]),
        "lpamd_group_upload": (ctypes.c_int, [vp, vp, i64, vp, vp]),
        "lpamd_group_fill_synthetic": (ctypes.c_int, [vp, i64, u64]),
        "lpamd_group_download": (ctypes.c_int, [vp, vp, i64, vp, vp]),
        "lpamd_group_need_init": (ctypes.c_int, [vp, P(i32)]),
        "lpamd_group_phase1": (ctypes.c_int, [vp, i32, P(i32)]),
        "lpamd_group_solve": (ctypes.c_int, [vp, i32, i64, P(i32), P(i64)]),
        "lpamd_group_run_pivots": (ctypes.c_int, [vp, i32, i64, P(i32), P(i64)]),
        "lpamd_group_pivot": (ctypes.c_int, [vp, i64, i64, i64]),
        "lpamd_group_column_values": (ctypes.c_int, [vp, vp]),
        "lpamd_group_objective_constant": (ctypes.c_int, [vp, P(ctypes.c_double)]),
        "lpamd_group_pivot_count": (ctypes.c_int, [vp, P(i64)]),
        "lpamd_group_trace": (ctypes.c_int, [vp, vp, vp, i64, P(i64)]),
        "lpamd_ratio_test_noalloc": (ctypes.c_int, [vp, vp, i64, vp, vp, vp]),
        "lpamd_rev_create": (ctypes.c_int, [P(vp), i64, i64, vp]),
        "lpamd_rev_destroy": (ctypes.c_int, [vp]),
        "lpamd_rev_upload": (ctypes.c_int, [vp, vp, i64, vp, vp, i32]),
        "lpamd_rev_fill_synthetic": (ctypes.c_int, [vp, i64, u64]),
        "lpamd_rev_solve": (ctypes.c_int, [vp, i64, P(i32), P(i64)]),
        "lpamd_rev_optimum": (ctypes.c_int, [vp, P(ctypes.c_double)]),
        "lpamd_rev_column_values": (ctypes.c_int, [vp, vp]),
        "lpamd_rev_pivot_count": (ctypes.c_int, [vp, P(i64)]),
        "lpamd_rev_trace": (ctypes.c_int, [vp, vp, vp, i64, P(i64)]),
        "lpamd_rev_list_sizes": (ctypes.c_int, [vp, P(i64), P(i64), P(i64)]),
        "lpamd_rev_read_state": (ctypes.c_int, [vp, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError here = header and library disagree
        fn.restype, fn.argtypes = res, args
    L._lpamd_symbols = tuple(sig)
    _LIB = L
    return L


def _ck(rc):
    if rc != 0:
        raise LpamdError("lpamd error %d: %s" % (rc, library().lpamd_last_error().decode()))


def fold_candidates(strategy, records, world, N):
    """Global winner among per-rank candidate records (a float64 array of 4*world
    words laid out as {key, gcol:int64, row:int64, ce}); pure host code."""
    rec = np.ascontiguousarray(records, dtype=np.float64)
    assert rec.size == 4 * world
    e, r, owner, res = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
    win = np.zeros(4, dtype=np.float64)
    _ck(library().lpamd_fold_candidates(int(strategy), rec.ctypes.data, int(world), int(N), ctypes.byref(e), ctypes.byref(r),
                                        ctypes.byref(owner), ctypes.byref(res), win.ctypes.data))
    return e.value, r.value, owner.value, res.value, win


def ratio_test_noalloc(col_ptr, rhs_ptr, m, row_out_ptr, ratio_out_ptr=None, stream=None):
    """lpamd_ratio_test_noalloc on raw device pointers (ints); asynchronous on `stream`."""
    _ck(library().lpamd_ratio_test_noalloc(ctypes.c_void_p(col_ptr), ctypes.c_void_p(rhs_ptr), int(m), ctypes.c_void_p(row_out_ptr),
                                           ctypes.c_void_p(ratio_out_ptr) if ratio_out_ptr else None,
                                           ctypes.c_void_p(stream) if stream else None))


class TableauEngine:
    """Device-resident m x (N+2) tableau shard on the current HIP device."""

    def __init__(self, m, N, rank=0, world=1, stream=None, borrowed=None, group=None):
        self.L = library()
        self.m, self.N, self.rank, self.world = int(m), int(N), int(rank), int(world)
        self.owned = borrowed is None
        self.group = group                # a view keeps its group alive and goes dead with it (ShardGroup.close)
        if borrowed is not None:          # a shard of a ShardGroup: the group owns the handle
            self._h = ctypes.c_void_p(borrowed)
            return
        self._h = ctypes.c_void_p()
        _ck(self.L.lpamd_create(ctypes.byref(self._h), self.m, self.N, self.rank, self.world,
                                ctypes.c_void_p(stream) if stream else None))

    @property
    def h(self):
        """The engine handle; a view of a shard refuses once its group has been closed (the handle is freed memory then)."""
        if not self.owned and (self.group is None or not self.group.h):
            raise LpamdError("this engine is a view of a ShardGroup that has been closed")
        return self._h

    def close(self):
        if self._h and self.owned:
            self.L.lpamd_destroy(self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data ----
    def upload(self, T, obj, is_basic):
        T = np.asfortranarray(T, dtype=np.float64)
        assert T.shape == (self.m, self.N + 2), (T.shape, self.m, self.N)
        obj = np.ascontiguousarray(obj, dtype=np.float64)
        is_basic = np.ascontiguousarray(is_basic, dtype=np.uint8)
        assert obj.shape == (self.N + 2,) and is_basic.shape == (self.N + 2,)
        _ck(self.L.lpamd_upload(self.h, T.ctypes.data, max(T.strides[1] // 8, self.m), obj.ctypes.data, is_basic.ctypes.data))

    def fill_synthetic(self, n_struct, seed):
        _ck(self.L.lpamd_fill_synthetic(self.h, int(n_struct), int(seed)))

    def download(self):
        T = np.zeros((self.m, self.N + 2), dtype=np.float64, order="F")
        obj = np.zeros(self.N + 2, dtype=np.float64)
        is_basic = np.zeros(self.N + 2, dtype=np.uint8)
        _ck(self.L.lpamd_download(self.h, T.ctypes.data, self.m, obj.ctypes.data, is_basic.ctypes.data))
        return T, obj, is_basic

    def download_objective(self):
        """Objective row only (entries of columns other ranks own stay 0); the tableau is not copied."""
        obj = np.zeros(self.N + 2, dtype=np.float64)
        _ck(self.L.lpamd_download(self.h, None, 0, obj.ctypes.data, None))
        return obj

    def download_basic(self):
        """Basis flags only (tableau_is_base_variable_, reference lp.cc:156-166)."""
        is_basic = np.zeros(self.N + 2, dtype=np.uint8)
        _ck(self.L.lpamd_download(self.h, None, 0, None, is_basic.ctypes.data))
        return is_basic

    def read_column(self, j):
        out = np.zeros(self.m, dtype=np.float64)
        _ck(self.L.lpamd_read_column(self.h, int(j), out.ctypes.data))
        return out

    def read_row(self, i):
        out = np.zeros(self.N + 2, dtype=np.float64)
        _ck(self.L.lpamd_read_row(self.h, int(i), out.ctypes.data))
        return out

    def write_objective(self, obj):
        obj = np.ascontiguousarray(obj, dtype=np.float64)
        assert obj.shape == (self.N + 2,)
        _ck(self.L.lpamd_write_objective(self.h, obj.ctypes.data))

    def basis_of_row(self):
        out = np.zeros(self.m, dtype=np.int64)
        _ck(self.L.lpamd_basis_of_row(self.h, out.ctypes.data))
        return out

    # ---- single steps ----
    def need_init(self):
        v = ctypes.c_int32()
        _ck(self.L.lpamd_need_init(self.h, ctypes.byref(v)))
        return bool(v.value)

    def price_max_cost(self):
        e = ctypes.c_int64()
        _ck(self.L.lpamd_price_max_cost(self.h, ctypes.byref(e)))
        return e.value

    def ratio_test(self, e):
        r, d = ctypes.c_int64(), ctypes.c_int64()
        _ck(self.L.lpamd_ratio_test(self.h, int(e), ctypes.byref(r), ctypes.byref(d)))
        return r.value, d.value

    def price_max_reduction(self):
        e, r, d = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _ck(self.L.lpamd_price_max_reduction(self.h, ctypes.byref(e), ctypes.byref(r), ctypes.byref(d)))
        return e.value, r.value, d.value

    def pivot(self, d, e, r):
        _ck(self.L.lpamd_pivot(self.h, int(d), int(e), int(r)))

    def phase1(self, strategy):
        res = ctypes.c_int32()
        _ck(self.L.lpamd_phase1(self.h, int(strategy), ctypes.byref(res)))
        return res.value

    def solve(self, strategy, max_pivots=-1):
        res, n = ctypes.c_int32(), ctypes.c_int64()
        _ck(self.L.lpamd_solve(self.h, int(strategy), int(max_pivots), ctypes.byref(res), ctypes.byref(n)))
        return res.value, n.value

    def run_pivots(self, strategy, k):
        res, n = ctypes.c_int32(), ctypes.c_int64()
        _ck(self.L.lpamd_run_pivots(self.h, int(strategy), int(k), ctypes.byref(res), ctypes.byref(n)))
        return res.value, n.value

    def column_values(self):
        x = np.zeros(self.N + 2, dtype=np.float64)
        _ck(self.L.lpamd_column_values(self.h, x.ctypes.data))
        return x

    def objective_constant(self):
        v = ctypes.c_double()
        _ck(self.L.lpamd_objective_constant(self.h, ctypes.byref(v)))
        return v.value

    def pivot_count(self):
        n = ctypes.c_int64()
        _ck(self.L.lpamd_pivot_count(self.h, ctypes.byref(n)))
        return n.value

    def trace(self):
        n = self.pivot_count()
        e = np.zeros(max(n, 1), dtype=np.int64)
        r = np.zeros(max(n, 1), dtype=np.int64)
        got = ctypes.c_int64()
        _ck(self.L.lpamd_trace(self.h, e.ctypes.data, r.ctypes.data, n, ctypes.byref(got)))
        return list(zip(e[:n].tolist(), r[:n].tolist()))

    def set_profiling(self, on):
        _ck(self.L.lpamd_set_profiling(self.h, 1 if on else 0))

    def update_stats(self):
        ms, n, act = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
        _ck(self.L.lpamd_update_stats(self.h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(act)))
        return ms.value, n.value, act.value

    def pass_stats(self):
        """(pivots applied by the last pass over the tableau, columns it streamed, columns the pivots touch one by one)"""
        k, s, t = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64()
        _ck(self.L.lpamd_pass_stats(self.h, ctypes.byref(k), ctypes.byref(s), ctypes.byref(t)))
        return k.value, s.value, t.value

    def profile_totals(self):
        """Since set_profiling(True): (ms inside the update kernel, passes, pivots, columns streamed, columns touched)."""
        ms, a, b, c, d = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _ck(self.L.lpamd_profile_totals(self.h, ctypes.byref(ms), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)))
        return ms.value, a.value, b.value, c.value, d.value

    def pricing_stats(self):
        """MAX_REDUCTION pricing from bounds: (pricings done, candidate columns evaluated exactly over all rows)."""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        _ck(self.L.lpamd_pricing_stats(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def loop_depth(self, strategy):
        d = ctypes.c_int32()
        _ck(self.L.lpamd_loop_depth(self.h, int(strategy), ctypes.byref(d)))
        return d.value

    # ---- multi-GPU steps (device pointers are plain ints, e.g. tensor.data_ptr()) ----
    def local_candidate(self, strategy, cand_ptr):
        _ck(self.L.lpamd_local_candidate(self.h, int(strategy), ctypes.c_void_p(cand_ptr)))

    def select_candidate(self, strategy, cands_ptr):
        e, r, owner, res = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
        _ck(self.L.lpamd_select_candidate(self.h, int(strategy), ctypes.c_void_p(cands_ptr), self.world,
                                          ctypes.byref(e), ctypes.byref(r), ctypes.byref(owner), ctypes.byref(res)))
        return e.value, r.value, owner.value, res.value

    def select_candidate_device(self, strategy, cands_ptr):
        _ck(self.L.lpamd_select_candidate_device(self.h, int(strategy), ctypes.c_void_p(cands_ptr), self.world))

    def export_column_or_zero(self, xbuf_ptr):
        _ck(self.L.lpamd_export_column_or_zero(self.h, ctypes.c_void_p(xbuf_ptr)))

    def set_pivot_budget(self, max_pivots):
        _ck(self.L.lpamd_set_pivot_budget(self.h, int(max_pivots)))

    def export_column(self, e, xbuf_ptr):
        _ck(self.L.lpamd_export_column(self.h, int(e), ctypes.c_void_p(xbuf_ptr)))

    def apply_pivot(self, strategy, e, r, xbuf_ptr):
        _ck(self.L.lpamd_apply_pivot(self.h, int(strategy), int(e), int(r), ctypes.c_void_p(xbuf_ptr)))

    def poll(self):
        res = ctypes.c_int32()
        _ck(self.L.lpamd_poll(self.h, ctypes.byref(res)))
        return res.value

    # look-ahead exchange (MAX_COST): set-up and update of a pivot as two calls, export of the
    # next candidate column in between, fold + fill of the exchange buffer on a second stream
    def lookahead_setup(self, xbuf_ptr):
        _ck(self.L.lpamd_lookahead_setup(self.h, ctypes.c_void_p(xbuf_ptr)))

    def lookahead_export(self, cand_ptr, xlocal_ptr):
        _ck(self.L.lpamd_lookahead_export(self.h, ctypes.c_void_p(cand_ptr), ctypes.c_void_p(xlocal_ptr)))

    def lookahead_update(self):
        _ck(self.L.lpamd_lookahead_update(self.h))

    def chain_depth(self):
        d = ctypes.c_int32()
        _ck(self.L.lpamd_chain_depth(self.h, ctypes.byref(d)))
        return d.value

    def chain_step(self, xbuf_ptr):
        _ck(self.L.lpamd_chain_step(self.h, ctypes.c_void_p(xbuf_ptr)))

    def lookahead_select(self, cands_ptr, xlocal_ptr, xbuf_ptr, stream=None):
        _ck(self.L.lpamd_lookahead_select(self.h, ctypes.c_void_p(cands_ptr), self.world, ctypes.c_void_p(xlocal_ptr),
                                          ctypes.c_void_p(xbuf_ptr), ctypes.c_void_p(stream) if stream else None))

    def exchange_doubles(self):
        return int(self.L.lpamd_exchange_bytes(self.m)) // 8

    # ---- pieces of the feasibility phase on shards ----
    def owns(self, col):
        """This rank holds global column `col` (the right-hand side is held everywhere)."""
        if col == self.N:
            return True
        return (col if col < self.N else self.N) % self.world == self.rank

    def fill_column(self, col, value):
        _ck(self.L.lpamd_fill_column(self.h, int(col), float(value)))

    def set_basic_flag(self, col, flag):
        _ck(self.L.lpamd_set_basic_flag(self.h, int(col), 1 if flag else 0))

    def pivot_shard(self, d, e, r, ce, xbuf_ptr):
        _ck(self.L.lpamd_pivot_shard(self.h, int(d), int(e), int(r), float(ce), ctypes.c_void_p(xbuf_ptr) if xbuf_ptr else None))


HOST_ALLGATHER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)


def comm_unique_id():
    """128-byte RCCL id for ShardGroup.join; made on one rank, delivered to the others by the caller."""
    buf = ctypes.create_string_buffer(128)
    _ck(library().lpamd_comm_unique_id(buf))
    return buf.raw


class ShardGroup:
    """One LP over column shards, driven by the library (lpamd_group_*): RCCL collectives and the
    pivot loop live behind the C ABI.  ShardGroup(m, N, n_shards, devices) runs all shards from this
    process (local_exchange=True: device copies instead of RCCL, shards may share a GPU);
    ShardGroup.join(m, N, rank, world, comm_id) is the one-process-per-GPU form."""

    def __init__(self, m, N, n_shards=1, devices=None, local_exchange=False, rehearse_rank0=False, _joined=None, _hosted=None):
        self.L = library()
        self.m, self.N = int(m), int(N)
        self.h = ctypes.c_void_p()
        if _hosted is not None:
            rank, world, gather = _hosted

            def trampoline(_ctx, send, recv, nbytes):
                try:
                    out = gather(ctypes.string_at(send, nbytes))
                    assert len(out) == nbytes * int(world)
                    ctypes.memmove(recv, out, len(out))
                    return 0
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return 1

            self._gather_cb = HOST_ALLGATHER(trampoline)   # kept alive as long as the group
            _ck(self.L.lpamd_group_create_rank_hosted(ctypes.byref(self.h), self.m, self.N, int(rank), int(world),
                                                      ctypes.cast(self._gather_cb, ctypes.c_void_p), None))
        elif _joined is not None:
            rank, world, comm_id = _joined
            _ck(self.L.lpamd_group_create_rank(ctypes.byref(self.h), self.m, self.N, int(rank), int(world), comm_id))
        else:
            dev = None
            if devices is not None:
                dev = (ctypes.c_int32 * int(n_shards))(*[int(d) for d in devices])
            flags = (1 if local_exchange else 0) | (2 if rehearse_rank0 else 0)
            _ck(self.L.lpamd_group_create(ctypes.byref(self.h), self.m, self.N, int(n_shards), dev, flags))
        a, b = ctypes.c_int32(), ctypes.c_int32()
        _ck(self.L.lpamd_group_shards(self.h, ctypes.byref(a), ctypes.byref(b)))
        self.n_local, self.world = a.value, b.value

    @classmethod
    def join(cls, m, N, rank, world, comm_id):
        return cls(m, N, _joined=(rank, world, comm_id))

    @classmethod
    def join_hosted(cls, m, N, rank, world, gather):
        """One process per shard with the collective supplied by the caller: gather(bytes) -> the ranks' bytes
        joined in rank order (lpamd_group_create_rank_hosted)."""
        return cls(m, N, _hosted=(rank, world, gather))

    def close(self):
        if self.h:
            self.L.lpamd_group_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def engine(self, k=0):
        """Local shard k as a TableauEngine view (trace, statistics, read-back; the group keeps ownership)."""
        h = ctypes.c_void_p()
        _ck(self.L.lpamd_group_engine(self.h, int(k), ctypes.byref(h)))
        e = TableauEngine(self.m, self.N, borrowed=h.value, group=self)
        e.world = self.world
        return e

    def comm_ranks(self):
        """Ranks of the RCCL communicator (0: in-process exchange)."""
        n = ctypes.c_int32()
        _ck(self.L.lpamd_group_comm_ranks(self.h, ctypes.byref(n)))
        return n.value

    def link_active(self):
        """True when chained MAX_COST steps trade their slots inside the step's launch instead of by a collective."""
        n = ctypes.c_int32()
        _ck(self.L.lpamd_group_link_active(self.h, ctypes.byref(n)))
        return n.value != 0

    def chain(self):
        d = ctypes.c_int32()
        _ck(self.L.lpamd_group_chain(self.h, ctypes.byref(d)))
        return d.value

    def upload(self, T, obj, is_basic):
        T = np.asfortranarray(T, dtype=np.float64)
        assert T.shape == (self.m, self.N + 2), (T.shape, self.m, self.N)
        obj = np.ascontiguousarray(obj, dtype=np.float64)
        is_basic = np.ascontiguousarray(is_basic, dtype=np.uint8)
        _ck(self.L.lpamd_group_upload(self.h, T.ctypes.data, max(T.strides[1] // 8, self.m), obj.ctypes.data, is_basic.ctypes.data))

    def fill_synthetic(self, n_struct, seed):
        _ck(self.L.lpamd_group_fill_synthetic(self.h, int(n_struct), int(seed)))

    def download(self, tableau=True):
        T = np.zeros((self.m, self.N + 2), dtype=np.float64, order="F") if tableau else None
        obj = np.zeros(self.N + 2, dtype=np.float64)
        is_basic = np.zeros(self.N + 2, dtype=np.uint8)
        _ck(self.L.lpamd_group_download(self.h, T.ctypes.data if tableau else None, self.m, obj.ctypes.data, is_basic.ctypes.data))
        return T, obj, is_basic

    def need_init(self):
        v = ctypes.c_int32()
        _ck(self.L.lpamd_group_need_init(self.h, ctypes.byref(v)))
        return bool(v.value)

    def phase1(self, strategy):
        res = ctypes.c_int32()
        _ck(self.L.lpamd_group_phase1(self.h, int(strategy), ctypes.byref(res)))
        return res.value

    def solve(self, strategy, max_pivots=-1):
        res, n = ctypes.c_int32(), ctypes.c_int64()
        _ck(self.L.lpamd_group_solve(self.h, int(strategy), int(max_pivots), ctypes.byref(res), ctypes.byref(n)))
        return res.value, n.value

    def run_pivots(self, strategy, k):
        res, n = ctypes.c_int32(), ctypes.c_int64()
        _ck(self.L.lpamd_group_run_pivots(self.h, int(strategy), int(k), ctypes.byref(res), ctypes.byref(n)))
        return res.value, n.value

    def pivot(self, d, e, r):
        _ck(self.L.lpamd_group_pivot(self.h, int(d), int(e), int(r)))

    def column_values(self):
        x = np.zeros(self.N + 2, dtype=np.float64)
        _ck(self.L.lpamd_group_column_values(self.h, x.ctypes.data))
        return x

    def objective_constant(self):
        v = ctypes.c_double()
        _ck(self.L.lpamd_group_objective_constant(self.h, ctypes.byref(v)))
        return v.value

    def pivot_count(self):
        n = ctypes.c_int64()
        _ck(self.L.lpamd_group_pivot_count(self.h, ctypes.byref(n)))
        return n.value

    def trace(self):
        n = self.pivot_count()
        e = np.zeros(max(n, 1), dtype=np.int64)
        r = np.zeros(max(n, 1), dtype=np.int64)
        got = ctypes.c_int64()
        _ck(self.L.lpamd_group_trace(self.h, e.ctypes.data, r.ctypes.data, n, ctypes.byref(got)))
        return list(zip(e[:n].tolist(), r[:n].tolist()))


class RevisedTableauEngine:
    """Revised simplex on the tableau, resident on the current HIP device
    (reference lp_revised_simplex_tableau.cc; one method per lpamd_rev_* entry point)."""

    def __init__(self, m, N, stream=None):
        self.L = library()
        self.m, self.N = int(m), int(N)
        self.h = ctypes.c_void_p()
        _ck(self.L.lpamd_rev_create(ctypes.byref(self.h), self.m, self.N, ctypes.c_void_p(stream) if stream else None))

    def close(self):
        if self.h:
            self.L.lpamd_rev_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, T, obj, is_basic, opt_reverted):
        T = np.asfortranarray(T, dtype=np.float64)
        assert T.shape[0] == self.m and T.shape[1] >= self.N + 1, (T.shape, self.m, self.N)
        obj = np.ascontiguousarray(obj, dtype=np.float64)
        is_basic = np.ascontiguousarray(is_basic, dtype=np.uint8)
        assert obj.shape[0] >= self.N + 1 and is_basic.shape[0] >= self.N
        _ck(self.L.lpamd_rev_upload(self.h, T.ctypes.data, max(T.strides[1] // 8, self.m), obj.ctypes.data, is_basic.ctypes.data,
                                    1 if opt_reverted else 0))

    def fill_synthetic(self, n_struct, seed):
        _ck(self.L.lpamd_rev_fill_synthetic(self.h, int(n_struct), int(seed)))

    def solve(self, max_pivots=-1):
        res, n = ctypes.c_int32(), ctypes.c_int64()
        _ck(self.L.lpamd_rev_solve(self.h, int(max_pivots), ctypes.byref(res), ctypes.byref(n)))
        return res.value, n.value

    def optimum(self):
        v = ctypes.c_double()
        _ck(self.L.lpamd_rev_optimum(self.h, ctypes.byref(v)))
        return v.value

    def column_values(self):
        x = np.zeros(self.N + 2, dtype=np.float64)
        _ck(self.L.lpamd_rev_column_values(self.h, x.ctypes.data))
        return x

    def pivot_count(self):
        n = ctypes.c_int64()
        _ck(self.L.lpamd_rev_pivot_count(self.h, ctypes.byref(n)))
        return n.value

    def trace(self):
        n = self.pivot_count()
        e = np.zeros(max(n, 1), dtype=np.int64)
        l = np.zeros(max(n, 1), dtype=np.int64)
        got = ctypes.c_int64()
        _ck(self.L.lpamd_rev_trace(self.h, e.ctypes.data, l.ctypes.data, n, ctypes.byref(got)))
        return list(zip(e[:got.value].tolist(), l[:got.value].tolist()))

    def list_sizes(self):
        """(rows the last update visited, rows the last pricing product read, dense rows of the inverse)."""
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _ck(self.L.lpamd_rev_list_sizes(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def read_state(self, inverse=True, transposed=True):
        """(Binv, BinvT, xB, cB, basis); a matrix that is not asked for comes back as None."""
        m = self.m
        Binv = np.zeros((m, m), dtype=np.float64) if inverse else None
        BinvT = np.zeros((m, m), dtype=np.float64) if transposed else None
        xB, cB, basis = np.zeros(m), np.zeros(m), np.zeros(m, dtype=np.int64)
        _ck(self.L.lpamd_rev_read_state(self.h, Binv.ctypes.data if inverse else None, BinvT.ctypes.data if transposed else None,
                                        xB.ctypes.data, cB.ctypes.data, basis.ctypes.data))
        return Binv, BinvT, xB, cB, basis
