"""The exchange inside the launches between PROCESSES: two or three ranks, one process each, all on GPU 0
(RCCL refuses that, so the collective is a gloo all-gather handed to lpamd_group_create_rank_hosted).  The
ranks trade IPC handles of their receive buffers, run the self-test, and from step 2 of every pass on their
launches store the slots into each other's buffers and poll their own while the other ranks' launches are
running: what the ranks of a multi-GPU run do, minus the xGMI hop.  Pivots against the CPU oracle."""
import json
import os
import socket
import subprocess
import sys

import pytest

from oracle import lp_oracle as lo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,chain,link", [(2, 8, 1), (3, 5, 1), (2, 8, 0)])
def test_ranks_in_processes_of_their_own(lpamd, tmp_path, world, chain, link):
    m, n, seed, legs = 2560, 3072, 7, (23, 1, 40, 0, 9)
    env = dict(os.environ, LPAMD_SHARD_CHAIN=str(chain), LPAMD_SHARD_LINK=str(link), LPAMD_TEST_MR_TAIL="37")
    port = free_port()
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "link_two_process_worker.py"), str(r), str(world), str(port), outs[r],
                               str(m), str(n), str(seed), ",".join(str(k) for k in legs)], env=env) for r in range(world)]
    try:
        codes = [p.wait(timeout=240) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world
    seen = [json.load(open(f)) for f in outs]
    for s in seen:
        assert "error" not in s, s.get("error")
        assert s["chain"] == chain and s["comm_ranks"] == 0
        assert s["link"] == bool(link)
        assert s["legs"] == [[lpamd.LIMIT, k] for k in legs]
        assert s["tail"] == [lpamd.LIMIT, 37]      # MAX_REDUCTION steps: their all-gathers go over the receive buffers as well
    dt = lo.DenseTableau.synthetic(m, n, seed)
    dt.s.skip_zero_cols = 1
    dt.s.fast_basis = 1
    assert dt.solve(lo.MAX_COST, max_pivots=sum(legs)) == lo.LIMIT
    assert dt.solve(lo.MAX_REDUCTION, max_pivots=sum(legs) + 37) == lo.LIMIT
    want = [list(t) for t in dt.trace()]
    for s in seen:
        assert s["trace"] == want, "rank %d walked other pivots" % s["rank"]
        assert s["rhs"] == seen[0]["rhs"], "the right-hand-side replicas diverged"
    assert seen[0]["rhs"] == [float(v).hex() for v in dt.T[:, dt.N]], "the right-hand side differs from the oracle's"


@pytest.mark.parametrize("world", [2, 4])
def test_full_size_ranks_in_processes_of_their_own(lpamd, tmp_path, world):
    """m=32768 n=65536 dealt over two / four processes (12.9 / 6.4 GB each, default depth of 26 pivots per pass, 128 row and
    193 / 97 column workgroups per launch): the ranks' pivots and right-hand side against the single-engine loop on the same GPU."""
    m, n, seed, legs = 32768, 65536, 1, (61, 30)
    env = dict(os.environ, LPAMD_SHARD_LINK="1")
    env.pop("LPAMD_SHARD_CHAIN", None)
    port = free_port()
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "link_two_process_worker.py"), str(r), str(world), str(port), outs[r],
                               str(m), str(n), str(seed), ",".join(str(k) for k in legs)], env=env) for r in range(world)]
    try:
        codes = [p.wait(timeout=280) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world
    seen = [json.load(open(f)) for f in outs]
    for s in seen:
        assert "error" not in s, s.get("error")
        assert s["link"] is True and s["chain"] == 26
        assert s["legs"] == [[lpamd.LIMIT, k] for k in legs]
    eng = lpamd.TableauEngine(m, n + m)
    eng.fill_synthetic(n, seed)
    assert eng.run_pivots(lpamd.MAX_COST, sum(legs)) == (lpamd.LIMIT, sum(legs))
    want = [list(t) for t in eng.trace()]
    rhs = [float(v).hex() for v in eng.read_column(n + m)]
    for s in seen:
        assert s["trace"] == want, "rank %d walked other pivots" % s["rank"]
        assert s["rhs"] == rhs, "rank %d: right-hand side differs from the single engine's" % s["rank"]


def test_a_slot_that_never_arrives_ends_the_run_on_every_rank(lpamd, tmp_path):
    """Rank 0 mis-tags its fifth slot (LPAMD_LINK_FAULT): the other rank's launch gives up after the time limit, both ranks
    learn of it at the next read of the state and stop with LPAMD_ERROR, nobody is left waiting; new groups that keep the
    collective then walk the oracle's pivots (the fallback of bench.py)."""
    m, n, seed, legs, world, chain = 2560, 3072, 7, (23, 40), 2, 8
    env = dict(os.environ, LPAMD_SHARD_CHAIN=str(chain), LPAMD_SHARD_LINK="1", LPAMD_LINK_FAULT="5", LPAMD_LINK_TIMEOUT_MS="300")
    port = free_port()
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "link_two_process_worker.py"), str(r), str(world), str(port), outs[r],
                               str(m), str(n), str(seed), ",".join(str(k) for k in legs)], env=env) for r in range(world)]
    try:
        codes = [p.wait(timeout=240) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world
    seen = [json.load(open(f)) for f in outs]
    dt = lo.DenseTableau.synthetic(m, n, seed)
    dt.s.skip_zero_cols = 1
    dt.s.fast_basis = 1
    assert dt.solve(lo.MAX_COST, max_pivots=sum(legs)) == lo.LIMIT
    want = [list(t) for t in dt.trace()]
    for s in seen:
        assert s["link"] is True
        assert "error" in s or s["legs"][-1][0] == lpamd.ERROR, "the run went on although a slot never arrived: %r" % (s["legs"],)
        again = s["again"]
        assert again["link"] is False and "error" not in again
        assert again["legs"] == [[lpamd.LIMIT, k] for k in legs]
        assert again["trace"] == want
