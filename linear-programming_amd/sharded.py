"""Column-sharded pivot loop over several GPUs, one process per GPU.

Each rank owns the tableau columns whose ordinal is congruent to its rank
(round-robin deal, so the dense columns stay balanced while the basis drifts)
plus a replica of the right-hand-side column.  One iteration of the reference's
main loop (reference lp_simplex_tableau.cc:181-377) needs two exchanges:

  1. all-gather of one 32-byte candidate record per rank, folded identically on
     every rank with the reference's lowest-index tie-break (pricing);
  2. broadcast of the entering column (m doubles) from the rank that owns it.

Everything else (ratio test on the broadcast column against the local RHS
replica, row gather, rank-1 update of the local shard) is rank-local and
bit-identical across ranks, so no further traffic is needed.
`torch.distributed` carries both exchanges: backend "nccl" (RCCL over xGMI) on
GPUs, "gloo" in the CPU tests, which plug in a host stand-in for the shard.
"""
import torch
import torch.distributed as dist

from . import engine as _eng


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


class ShardedSolver:
    """Drives one shard per rank through the pivot loop.

    exchange="allreduce" (default): no host round trip inside the loop.  Every rank
    folds the gathered candidates on the device; the pivot column travels as a
    64-bit-integer sum-all-reduce in which only its owner contributes non-zero
    words, so the root never has to be known on the host.  The host only polls the
    status every `poll_every` pivots.
    exchange="broadcast": the host reads the gathered candidates each pivot to learn
    the owner and broadcasts the column from it.
    """

    def __init__(self, shard, group=None, exchange="allreduce", poll_every=64):
        assert exchange in ("allreduce", "broadcast")
        self.shard = shard
        self.group = group
        self.exchange = exchange
        self.poll_every = poll_every
        self.use_dist = dist.is_available() and dist.is_initialized()   # collectives also run for a world of one
        self.rank = dist.get_rank(group) if self.use_dist else 0
        self.world = dist.get_world_size(group) if self.use_dist else 1
        assert self.world == shard.world and self.rank == shard.rank
        self.cand = shard.new_buffer(4)
        self.cands = shard.new_buffer(4 * self.world)
        self.xbuf = shard.new_buffer(shard.exchange_doubles())
        self.xbuf_words = self.xbuf.view(torch.int64)
        if self.xbuf.is_cuda:
            # the zero-fills above ran on the allocating stream; the loop uses the shard's own stream
            torch.cuda.synchronize(self.xbuf.device)

    def run_pivots(self, strategy, k):
        """At most k pivots (k < 0: until termination).  Returns (Result, pivots done)."""
        run = self._run_allreduce if self.exchange == "allreduce" else self._run_broadcast
        stream = getattr(self.shard, "stream", None)
        if stream is None:
            return run(strategy, k)
        with torch.cuda.stream(stream):
            return run(strategy, k)

    def _gather_candidates(self):
        if self.use_dist:
            dist.all_gather_into_tensor(self.cands, self.cand, group=self.group)
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
                    dist.all_reduce(self.xbuf_words, op=dist.ReduceOp.SUM, group=self.group)
                sh.apply_pivot(strategy, -1, -1, self.xbuf)
            enqueued += n
            res = sh.poll()                  # the only host synchronisation
            if res != _eng.LIMIT or (k >= 0 and enqueued >= k):
                break
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
                src = owner if self.group is None else dist.get_global_rank(self.group, owner)
                dist.broadcast(self.xbuf, src=src, group=self.group)
            sh.apply_pivot(strategy, e, r, self.xbuf)
            done += 1
        res = sh.poll()
        return res, done
