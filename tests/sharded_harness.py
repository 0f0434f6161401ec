"""TEST HARNESS: the column-sharded pivot loop driven from Python over the step-wise C ABI, one
process per rank.

The product runs these loops itself behind lpamd_group_* (RCCL inside, no Python in the loop; that is
what bench.py and the `solver` binary use).  This module is the caller-driven form of the same
protocols: it shows how a host that owns the collectives drives the engines step by step
(lpamd_local_candidate, lpamd_lookahead_*, lpamd_chain_step, ...), and it is what the multi-process
CPU tests run over gloo with a host stand-in in place of the GPU shard.  Its feasibility phase
restates reference lp_simplex_tableau.cc:64-171 in numpy as a checker of the library's own
(lpamd_phase1, lpamd_group_solve).

Each rank owns the tableau columns whose ordinal is congruent to its rank
(round-robin deal, so the dense columns stay balanced while the basis drifts)
plus a replica of the right-hand-side column.  One iteration of the reference's
main loop (reference lp_simplex_tableau.cc:181-377) needs two exchanges:

  1. all-gather of one 32-byte candidate record per rank, folded identically on
     every rank with the reference's lowest-index tie-break (pricing);
  2. broadcast of the entering column (m doubles) from the rank that owns it.

With MAX_COST pricing both travel on a second stream while the previous pivot's
rank-1 update streams the shard (look-ahead exchange): the set-up of a pivot
already leaves the next objective row, so every rank knows its next candidate
and can send that column as the pending update will leave it.

Everything else (ratio test on the broadcast column against the local RHS
replica, row gather, rank-1 update of the local shard) is rank-local and
bit-identical across ranks, so no further traffic is needed.
`torch.distributed` carries both exchanges: backend "nccl" (RCCL over xGMI) on
GPUs, "gloo" in the CPU tests, which plug in a host stand-in for the shard.
"""
import numpy as np
import os

import torch
import torch.distributed as dist

import importlib

_eng = importlib.import_module("linear-programming_amd.engine")

_EPS32 = np.float32(1e-6)


# reference base.h:61-66: the predicates narrow to float before comparing
def _is_zero(x):
    return abs(np.float32(x)) < _EPS32


def _is_neg(x):
    return np.float32(x) <= -_EPS32


def _is_nonneg(x):
    return np.float32(x) > -_EPS32


def _flush(a):
    """Zero drop of the tableau library after every add (reference lp_simplex_tableau.cc:5-8)."""
    a = np.array(a, dtype=np.float64, copy=True)
    a[np.abs(a) < float(_EPS32)] = 0.0
    return a


class EngineShard:
    """Adapter: a TableauEngine shard addressed with torch tensors."""

    def __init__(self, m, N, rank, world, device, stream=None):
        self.device = device
        # The engine's kernels, the buffer copies and the collectives must share ONE
        # stream so that they are ordered without host synchronisation.  torch's
        # default stream has handle 0, which the C ABI reads as "create your own",
        # so the shard brings an explicit side stream; callers run the exchange
        # steps under `with torch.cuda.stream(shard.stream)`.
        self.stream = stream if stream is not None else torch.cuda.Stream(device)
        self.xstream = torch.cuda.Stream(device)     # exchange stream of the look-ahead loop
        self.eng = _eng.TableauEngine(m, N, rank, world, stream=self.stream.cuda_stream)
        self.m, self.N, self.rank, self.world = m, N, rank, world

    def exchange_doubles(self):
        return self.eng.exchange_doubles()

    def new_buffer(self, n_doubles):
        return torch.zeros(n_doubles, dtype=torch.float64, device=self.device)

    def local_candidate(self, strategy, cand):
        self.eng.local_candidate(strategy, cand.data_ptr())

    def select_candidate(self, strategy, cands):
        return self.eng.select_candidate(strategy, cands.data_ptr())

    def export_column(self, e, xbuf):
        self.eng.export_column(e, xbuf.data_ptr())

    def apply_pivot(self, strategy, e, r, xbuf):
        self.eng.apply_pivot(strategy, e, r, xbuf.data_ptr())

    def select_candidate_device(self, strategy, cands):
        self.eng.select_candidate_device(strategy, cands.data_ptr())

    def export_column_or_zero(self, xbuf):
        self.eng.export_column_or_zero(xbuf.data_ptr())

    def set_pivot_budget(self, k):
        self.eng.set_pivot_budget(k)

    def pivot_count(self):
        return self.eng.pivot_count()

    def poll(self):
        return self.eng.poll()

    # ---- look-ahead exchange (MAX_COST) ----
    def lookahead_setup(self, xbuf):
        self.eng.lookahead_setup(xbuf.data_ptr())

    def lookahead_export(self, cand, xlocal):
        self.eng.lookahead_export(cand.data_ptr(), xlocal.data_ptr())

    def lookahead_update(self):
        self.eng.lookahead_update()

    def chain_depth(self):
        return self.eng.chain_depth()

    def chain_step(self, xbuf):
        self.eng.chain_step(xbuf.data_ptr())

    def lookahead_select(self, cands, xlocal, xbuf):
        """Enqueued on torch's current stream (the exchange stream), not on the engine's."""
        self.eng.lookahead_select(cands.data_ptr(), xlocal.data_ptr(), xbuf.data_ptr(),
                                  torch.cuda.current_stream(self.device).cuda_stream)

    # ---- pieces of the feasibility phase (host vectors are numpy, indexed by global column) ----
    def owns(self, col):
        return self.eng.owns(col)

    def need_init(self):
        return self.eng.need_init()

    def read_row_local(self, i):
        return self.eng.read_row(i)

    def read_column(self, col):
        return self.eng.read_column(col)

    def download_objective(self):
        return self.eng.download_objective()

    def download_basic(self):
        return self.eng.download_basic()

    def write_objective(self, obj):
        self.eng.write_objective(obj)

    def fill_column(self, col, value):
        self.eng.fill_column(col, value)

    def set_basic_flag(self, col, flag):
        self.eng.set_basic_flag(col, flag)

    def pivot_shard(self, d, e, r, ce, xbuf):
        self.eng.pivot_shard(d, e, r, ce, xbuf.data_ptr())

    def objective_constant(self):
        return self.eng.objective_constant()

    def column_values(self):
        return self.eng.column_values()


class DistComm:
    """The three exchanges of the sharded loop over torch.distributed (RCCL on GPUs, gloo on CPU).
    Anything with the same five members can stand in for it (the tests use an in-process one)."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_gather(self, out, inp):
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_reduce_sum_words(self, words):
        dist.all_reduce(words, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast(self, t, owner):
        src = owner if self.group is None else dist.get_global_rank(self.group, owner)
        dist.broadcast(t, src=src, group=self.group)


class ShardedSolver:
    """Drives one shard per rank through the pivot loop.

    exchange="lookahead" (default): like "allreduce", and for MAX_COST the two collectives
    of pivot k+1 run on the shard's exchange stream while the rank-1 update of pivot k
    streams the tableau, so the critical path of a pivot holds no collective at all
    (MAX_REDUCTION prices on the updated columns and keeps the in-order exchange).
    exchange="allreduce": no host round trip inside the loop.  Every rank
    folds the gathered candidates on the device; the pivot column travels as a
    64-bit-integer sum-all-reduce in which only its owner contributes non-zero
    words, so the root never has to be known on the host.  The host only polls the
    status every `poll_every` pivots.
    exchange="broadcast": the host reads the gathered candidates each pivot to learn
    the owner and broadcasts the column from it.
    exchange="fused": ONE all-gather per pivot, the protocol of the library's own loop
    (lpamd_group_*): every rank sends its candidate record together with the candidate column
    itself in one slot; every rank folds the records and takes the winner's column from the
    winner's slot.  MAX_COST runs it as the look-ahead loop with chains of pivots per pass.
    """

    def __init__(self, shard, group=None, exchange="lookahead", poll_every=64, comm=None, chain=None):
        assert exchange in ("lookahead", "allreduce", "broadcast", "fused")
        self.shard = shard
        # look-ahead loop: pivots set up per pass over the shard (1: one pivot per pass); shards without
        # lpamd_chain_step (the host stand-in of the CPU tests) keep one pivot per pass
        depth = shard.chain_depth() if hasattr(shard, "chain_depth") else 1
        if chain is None:
            chain = int(os.environ.get("LPAMD_SHARD_CHAIN", depth))
        self.chain = max(1, min(int(chain), depth))
        if comm is None and dist.is_available() and dist.is_initialized():
            comm = DistComm(group)               # collectives also run for a world of one
        self.comm = comm
        self.exchange = exchange
        self.poll_every = poll_every
        self.use_dist = comm is not None
        self.rank = comm.rank if comm is not None else 0
        self.world = comm.world if comm is not None else 1
        assert self.world == shard.world and self.rank == shard.rank
        self.cand = shard.new_buffer(4)
        self.cands = shard.new_buffer(4 * self.world)
        self.xbuf = shard.new_buffer(shard.exchange_doubles())
        self.xbuf_words = self.xbuf.view(torch.int64)
        self.xlocal = shard.new_buffer(shard.exchange_doubles())    # look-ahead: this rank's next candidate column
        self.slot_words = 8 + shard.exchange_doubles()              # fused exchange: 64-byte header (record) + column
        self.slot = shard.new_buffer(self.slot_words) if exchange == "fused" else None
        self.slots = shard.new_buffer(self.slot_words * self.world) if exchange == "fused" else None
        self.ev_sent = self.ev_arrived = None
        if self.xbuf.is_cuda and getattr(shard, "xstream", None) is not None:
            self.ev_sent, self.ev_arrived = torch.cuda.Event(), torch.cuda.Event()
        if self.xbuf.is_cuda:
            # the zero-fills above ran on the allocating stream; the loop uses the shard's own stream
            torch.cuda.synchronize(self.xbuf.device)
        self.wbuf = None    # N + 2 doubles, created by the first solve()

    def run_pivots(self, strategy, k):
        """At most k pivots (k < 0: until termination).  Returns (Result, pivots done)."""
        if self.exchange == "broadcast":
            run = self._run_broadcast
        elif self.exchange == "fused" and strategy != _eng.MAX_COST:
            run = self._run_fused_in_order
        elif self.exchange in ("lookahead", "fused") and strategy == _eng.MAX_COST:
            run = self._run_lookahead
        else:
            run = self._run_allreduce
        stream = getattr(self.shard, "stream", None)
        if stream is None:
            return run(strategy, k)
        with torch.cuda.stream(stream):
            return run(strategy, k)

    def _gather_candidates(self):
        if self.use_dist:
            self.comm.all_gather(self.cands, self.cand)
        else:
            self.cands.copy_(self.cand)

    def _run_allreduce(self, strategy, k):
        sh = self.shard
        start = sh.pivot_count()
        sh.set_pivot_budget(k)
        enqueued = 0
        while True:
            n = self.poll_every if k < 0 else min(self.poll_every, k - enqueued)
            for _ in range(n):
                sh.local_candidate(strategy, self.cand)
                self._gather_candidates()
                sh.select_candidate_device(strategy, self.cands)
                sh.export_column_or_zero(self.xbuf)
                if self.use_dist:
                    self.comm.all_reduce_sum_words(self.xbuf_words)
                sh.apply_pivot(strategy, -1, -1, self.xbuf)
            enqueued += n
            res = sh.poll()                  # the only host synchronisation
            if res != _eng.LIMIT or (k >= 0 and enqueued >= k):
                break
        return res, sh.pivot_count() - start

    def _lookahead_exchange(self):
        """Exchange for the NEXT pivot: gather the records, keep the winner, deliver its column.
        With two streams it is ordered after what the loop stream holds so far (the export of
        the local candidate) and runs beside what the loop stream gets next (the update); the
        caller has made the exchange stream torch's current stream."""
        sh = self.shard
        two_streams = self.ev_sent is not None
        if two_streams:
            self.ev_sent.record(sh.stream)
            sh.xstream.wait_event(self.ev_sent)
        if self.exchange == "fused":
            owner = self._slot_exchange(_eng.MAX_COST)
            sh.lookahead_select(self.cands, self.xlocal, self.xbuf)      # keeps the winner's record in the engine
            if owner >= 0:
                self.xbuf.copy_(self.slots.view(self.world, self.slot_words)[owner, 8:])
        else:
            self._gather_candidates()
            sh.lookahead_select(self.cands, self.xlocal, self.xbuf)
            if self.use_dist:
                self.comm.all_reduce_sum_words(self.xbuf_words)
        if two_streams:
            self.ev_arrived.record(sh.xstream)

    def _slot_exchange(self, strategy):
        """The fused exchange: record (self.cand) and column (self.xlocal) travel in one all-gather.
        Leaves the records in self.cands; returns the winner's rank (-1: no winner).  The harness
        folds on the host; the library's loop does the same fold in a kernel."""
        self.slot[:4].copy_(self.cand)
        self.slot[8:].copy_(self.xlocal)
        if self.use_dist:
            self.comm.all_gather(self.slots, self.slot)
        else:
            self.slots.copy_(self.slot)
        self.cands.copy_(self.slots.view(self.world, self.slot_words)[:, :4].reshape(-1))
        _, _, owner, _, _ = _eng.fold_candidates(strategy, self.cands.cpu().numpy(), self.world, self.shard.N)
        return owner

    def _run_fused_in_order(self, strategy, k):
        """Pricing on the current tableau, ONE all-gather of record + column, fold, pivot."""
        sh = self.shard
        start = sh.pivot_count()
        sh.set_pivot_budget(k)
        enqueued = 0
        while True:
            n = self.poll_every if k < 0 else min(self.poll_every, k - enqueued)
            for _ in range(n):
                sh.local_candidate(strategy, self.cand)
                gcol = int(self.cand.cpu().view(torch.int64)[1])
                self.xlocal.zero_()
                if gcol >= 0:
                    sh.export_column(gcol, self.xlocal)
                owner = self._slot_exchange(strategy)
                sh.select_candidate_device(strategy, self.cands)      # winner + status (budget, optimum, unbounded)
                if owner >= 0:
                    self.xbuf.copy_(self.slots.view(self.world, self.slot_words)[owner, 8:])
                sh.apply_pivot(strategy, -1, -1, self.xbuf)
            enqueued += n
            res = sh.poll()
            if res != _eng.LIMIT or (k >= 0 and enqueued >= k):
                break
        return res, sh.pivot_count() - start

    def _run_lookahead(self, strategy, k):
        assert strategy == _eng.MAX_COST
        sh = self.shard
        if self.ev_sent is None:
            return self._lookahead_loop(k, False)
        # The engine enqueues on its own stream whatever torch's current stream is, so the exchange
        # stream stays current for the whole loop: the collectives and the fold land on it.
        with torch.cuda.stream(sh.xstream):
            return self._lookahead_loop(k, True)

    def _lookahead_loop(self, k, two_streams):
        sh = self.shard
        start = sh.pivot_count()
        sh.set_pivot_budget(k)
        sh.lookahead_export(self.cand, self.xlocal)      # nothing pending: the column as stored
        self._lookahead_exchange()
        enqueued = 0
        while True:
            n = self.poll_every if k < 0 else min(self.poll_every, k - enqueued)
            passes = (n + self.chain - 1) // self.chain
            for _ in range(passes):
                if two_streams:
                    sh.stream.wait_event(self.ev_arrived)
                sh.lookahead_setup(self.xbuf)                 # ratio test + set-up of the first pivot of the pass
                for _j in range(1, self.chain):
                    # the next pivot of the chain: its exchange is on the critical path (nothing to hide it behind),
                    # the pass over the shard that it saves is not
                    sh.lookahead_export(self.cand, self.xlocal)
                    self._lookahead_exchange()
                    if two_streams:
                        sh.stream.wait_event(self.ev_arrived)
                    sh.chain_step(self.xbuf)
                sh.lookahead_export(self.cand, self.xlocal)   # next candidate, as the pending pass will leave it
                self._lookahead_exchange()                    # ... travels while ...
                sh.lookahead_update()                         # ... the pass streams the shard
            n = passes * self.chain
            enqueued += n
            res = sh.poll()
            if res != _eng.LIMIT or (k >= 0 and enqueued >= k):
                break
        if two_streams:
            sh.stream.wait_event(self.ev_arrived)        # the last exchange is not consumed; keep the streams ordered
        return res, sh.pivot_count() - start

    def _run_broadcast(self, strategy, k):
        sh = self.shard
        sh.set_pivot_budget(-1)
        done = 0
        while k < 0 or done < k:
            sh.local_candidate(strategy, self.cand)
            self._gather_candidates()
            e, r, owner, res = sh.select_candidate(strategy, self.cands)
            if res != _eng.LIMIT:
                return res, done
            if owner == self.rank:
                sh.export_column(e, self.xbuf)
            if self.use_dist:
                self.comm.broadcast(self.xbuf, owner)
            sh.apply_pivot(strategy, e, r, self.xbuf)
            done += 1
        res = sh.poll()
        return res, done

    # ---- TableauSimplexSolve on shards: feasibility phase included ------------------------
    # The scans of reference lp_simplex_tableau.cc:64-171 look at single rows, columns and
    # the objective row.  Every rank assembles them from the shards with an exact sum
    # (only the owner of an entry contributes a non-zero bit pattern), takes the same
    # decisions on the same data, and applies the same explicit pivots to its shard.

    def _sum_exact(self, vec, buf):
        """All ranks get the element-wise 64-bit integer sum of `vec` (float64 numpy)."""
        n = len(vec)
        t = buf[:n]
        t.copy_(torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64)))
        if self.use_dist:
            self.comm.all_reduce_sum_words(t.view(torch.int64))
        return t.cpu().numpy().copy()

    def _assemble(self, local_vec):
        """Full N+2 vector from per-rank vectors that hold the local columns only; the
        replicated right-hand-side entry is taken from rank 0."""
        v = np.array(local_vec, dtype=np.float64, copy=True)
        if self.rank != 0:
            v[self.shard.N] = 0.0
        return self._sum_exact(v, self.wbuf)

    def _column(self, col):
        """Column `col` on every rank (read by its owner)."""
        m = self.shard.m
        v = np.zeros(m, dtype=np.float64)
        if col == self.shard.N:
            return self.shard.read_column(col)
        if self.shard.owns(col):
            v = self.shard.read_column(col)
        return self._sum_exact(v, self.xbuf)

    def _pivot(self, d, e, r, ce):
        """LPModel::TableauPivot(d -> out, e -> in, row r) on every shard."""
        sh = self.shard
        if self.use_dist or not sh.owns(e):
            self.xbuf.zero_()
            if sh.owns(e):
                sh.export_column(e, self.xbuf)
            if self.use_dist:
                self.comm.all_reduce_sum_words(self.xbuf_words)
        else:
            sh.export_column(e, self.xbuf)
        sh.pivot_shard(d, e, r, ce, self.xbuf)

    def _phase1(self, strategy):
        sh = self.shard
        N, A, W, m = sh.N, sh.N + 1, sh.N + 2, sh.m
        if not sh.need_init():
            return _eng.SOLVED
        saved = self._assemble(sh.download_objective())
        tmp = np.zeros(W, dtype=np.float64)
        tmp[A] = -1.0                                                    # :79
        sh.write_objective(tmp)
        sh.set_basic_flag(A, 0)                                          # :71
        sh.fill_column(A, 1.0)                                           # :82-86
        rhs = sh.read_column(N)
        r, mv = -1, 0.0
        for i in range(m):                                               # MinReduce, ties to the later row (:88-91)
            if not (mv < rhs[i]):
                mv, r = rhs[i], i
        if r < 0:
            return _eng.ERROR
        row = self._assemble(sh.read_row_local(r))
        basic = self._assemble(sh.download_basic().astype(np.float64)) != 0.0
        d = -1
        for j in range(W):                                               # :94-104
            if j == N or _is_zero(row[j]):
                continue
            if basic[j]:
                d = j
                break
        if d < 0:
            return _eng.ERROR
        self._pivot(d, A, r, tmp[A])
        rhs = sh.read_column(N)
        if any(not _is_nonneg(b) for b in rhs):                          # :106-108 (live assert upstream)
            return _eng.ERROR
        inner, _ = self.run_pivots(strategy, -1)                         # :110
        if inner != _eng.SOLVED:
            return _eng.ERROR                                            # :111 (live assert upstream)
        if _is_neg(sh.objective_constant()):
            return _eng.NOSOLUTION                                       # :113-115
        basic = self._assemble(sh.download_basic().astype(np.float64)) != 0.0
        if basic[A]:                                                     # :117-135
            col_a = self._column(A)
            pivoted = False
            for i in range(m):
                if pivoted:
                    break
                if _is_zero(col_a[i]):
                    continue
                row = self._assemble(sh.read_row_local(i))
                for j in range(W):
                    if j != N and basic[j]:
                        continue
                    if _is_zero(row[j]):
                        continue
                    if j == N:
                        return _eng.ERROR   # upstream would pivot on the constant column and abort on its assert
                    obj_now = self._assemble(sh.download_objective())
                    self._pivot(A, j, i, obj_now[j])
                    pivoted = True
                    break
            if not pivoted:
                return _eng.ERROR
            sh.set_basic_flag(A, 0)
            basic = self._assemble(sh.download_basic().astype(np.float64)) != 0.0
        sh.fill_column(A, 0.0)                                           # :143
        saved[A] = 0.0
        acc = np.zeros(W, dtype=np.float64)                              # objective re-expression (:149-166)
        for j in range(W):
            if j == N or not basic[j] or _is_zero(saved[j]):
                continue
            col = self._column(j)
            rho, bv = -1, float(_EPS32)                                  # MaxAbsReduce init {-1, kEpsilonF}
            for i in range(m):
                if not (abs(bv) > abs(col[i])):
                    bv, rho = col[i], i
            if rho < 0:
                return _eng.ERROR
            row = self._assemble(sh.read_row_local(rho))
            s = -saved[j] / row[j]                                       # :162
            acc = _flush(acc + row * s)
        saved = _flush(saved + acc)                                      # :166
        sh.write_objective(saved)
        return _eng.SOLVED

    def solve(self, strategy, max_pivots=-1):
        """LPModel::TableauSimplexSolve (reference lp_simplex_tableau.cc:173-379) over the
        shards: feasibility phase if the right-hand side needs it, then the pivot loop
        (max_pivots limits the loop after that phase).  Returns (Result, pivots applied)."""
        if strategy not in (_eng.MAX_COST, _eng.MAX_REDUCTION):
            raise _eng.LpamdError("pivoting strategy %r is not implemented for the tableau path" % (strategy,))
        sh = self.shard
        if self.wbuf is None:
            self.wbuf = sh.new_buffer(sh.N + 2)
            if self.wbuf.is_cuda:
                torch.cuda.synchronize(self.wbuf.device)
        start = sh.pivot_count()
        stream = getattr(sh, "stream", None)
        if sh.need_init():
            if stream is None:
                r1 = self._phase1(strategy)
            else:
                with torch.cuda.stream(stream):
                    r1 = self._phase1(strategy)
            if r1 != _eng.SOLVED:
                return (_eng.NOSOLUTION if r1 == _eng.NOSOLUTION else _eng.ERROR), sh.pivot_count() - start
            if sh.need_init():
                return _eng.ERROR, sh.pivot_count() - start              # :179
        res, _ = self.run_pivots(strategy, max_pivots)
        return res, sh.pivot_count() - start

    def column_values(self):
        """x[j] of every global column on every rank (GetTableauSimplexSolution's values)."""
        x = self.shard.column_values()
        return self._assemble(x)
