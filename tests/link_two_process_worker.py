"""One rank of a column-sharded run whose ranks are PROCESSES that share GPU 0 (started by
tests/test_link_processes_gpu.py).  The collective is a gloo all-gather handed to the library
(lpamd_group_create_rank_hosted); steps 2.. of a MAX_COST pass trade inside their launches through
receive buffers the ranks map from each other with IPC handles.  Writes what it saw as JSON."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    m, n, seed = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    legs = [int(x) for x in sys.argv[8].split(",")]
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    lp = importlib.import_module("linear-programming_amd")

    def gather(mine):
        every = [None] * world
        dist.all_gather_object(every, mine)
        return b"".join(every)

    def walk(g, seen):
        g.fill_synthetic(n, seed)
        try:
            for k in legs:
                res, done = g.run_pivots(lp.MAX_COST, k)
                seen["legs"].append([res, done])
                if res != lp.LIMIT:
                    break
            tail = int(os.environ.get("LPAMD_TEST_MR_TAIL", "0"))
            if tail and all(r == lp.LIMIT for r, _ in seen["legs"]):     # the MAX_REDUCTION loop continues the same LP
                seen["tail"] = list(g.run_pivots(lp.MAX_REDUCTION, tail))
            seen["trace"] = g.trace()
            seen["rhs"] = [float(v).hex() for v in g.engine(0).read_column(n + m)]
        except Exception as ex:   # the test reads the message
            seen["error"] = str(ex)

    g = lp.ShardGroup.join_hosted(m, n + m, rank, world, gather)
    seen = {"rank": rank, "link": g.link_active(), "chain": g.chain(), "comm_ranks": g.comm_ranks(), "legs": []}
    walk(g, seen)
    if os.environ.get("LPAMD_LINK_FAULT"):
        # what bench.py does when the trial passes fail: every rank forms a new group that keeps the collective
        bad = [None] * world
        dist.all_gather_object(bad, "error" in seen or any(r != lp.LIMIT for r, _ in seen["legs"]))
        if any(bad):
            g.close()
            os.environ["LPAMD_SHARD_LINK"] = "0"
            g = lp.ShardGroup.join_hosted(m, n + m, rank, world, gather)
            again = {"rank": rank, "link": g.link_active(), "chain": g.chain(), "comm_ranks": g.comm_ranks(), "legs": []}
            walk(g, again)
            seen["again"] = again
    with open(out_path, "w") as f:
        json.dump(seen, f)
    g.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
