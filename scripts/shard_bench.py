"""Per-rank cost of one pivot at multi-GPU shard sizes, measured on ONE GPU.

Creates the shard that rank 0 of a `--world`-GPU run would hold, and drives the
host-free exchange loop with this rank's own candidate as the winner (the other
ranks' records are empty), so every step does the work it would do in the real
run; the two collectives run over a one-rank RCCL group (their multi-rank wire
time is not included).  Usage: python scripts/shard_bench.py --world 8
"""
import argparse, importlib, os, socket, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lp = importlib.import_module("linear-programming_amd")
sharded = importlib.import_module("linear-programming_amd.sharded")

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--m", type=int, default=32768)
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--steps", type=int, default=128)
ap.add_argument("--strategy", default="max_cost")
ap.add_argument("--exchange", default="lookahead", choices=["lookahead", "allreduce"])
ap.add_argument("--chain", type=int, default=8, help="pivots set up per pass over the shard (look-ahead loop)")
args = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
m, n, G = args.m, args.n, args.world
shard = sharded.EngineShard(m, n + m, 0, G, dev)
shard.eng.fill_synthetic(n, 1)
strat = lp.STRATEGY[args.strategy]
cand = shard.new_buffer(4)
cands = shard.new_buffer(4 * G)
cands.view(torch.int64)[1::4] = -1          # other ranks: no candidate
xbuf = shard.new_buffer(shard.exchange_doubles())
one = shard.new_buffer(4)
xlocal = shard.new_buffer(shard.exchange_doubles())
ev_sent, ev_arrived = torch.cuda.Event(), torch.cuda.Event()
torch.cuda.synchronize()

def exchange_ahead():
    ev_sent.record(shard.stream)
    shard.xstream.wait_event(ev_sent)
    dist.all_gather_into_tensor(one, cand)
    cands[:4].copy_(one)
    shard.lookahead_select(cands, xlocal, xbuf)
    dist.all_reduce(xbuf_words)
    ev_arrived.record(shard.xstream)

xbuf_words = xbuf.view(torch.int64)

def run_lookahead(k):
    with torch.cuda.stream(shard.xstream):   # the engine uses its own stream whatever torch's current one is
        shard.set_pivot_budget(k)
        shard.lookahead_export(cand, xlocal)
        exchange_ahead()
        for _ in range((k + args.chain - 1) // args.chain):
            shard.stream.wait_event(ev_arrived)
            shard.lookahead_setup(xbuf)
            for _j in range(1, args.chain):
                shard.lookahead_export(cand, xlocal)
                exchange_ahead()
                shard.stream.wait_event(ev_arrived)
                shard.chain_step(xbuf)
            shard.lookahead_export(cand, xlocal)
            exchange_ahead()
            shard.lookahead_update()
        shard.stream.wait_event(ev_arrived)
        return shard.poll()

def run(k):
    if args.exchange == "lookahead" and args.strategy == "max_cost":
        return run_lookahead(k)
    with torch.cuda.stream(shard.stream):
        shard.set_pivot_budget(k)
        for _ in range(k):
            shard.local_candidate(strat, cand)
            dist.all_gather_into_tensor(one, cand)          # 32 B/rank exchange (one-rank group)
            cands[:4].copy_(one)
            shard.select_candidate_device(strat, cands)
            shard.export_column_or_zero(xbuf)
            dist.all_reduce(xbuf.view(torch.int64))          # pivot column exchange (one-rank group)
            shard.apply_pivot(strat, -1, -1, xbuf)
        return shard.poll()

run(8)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(args.steps)
torch.cuda.synchronize()
el = time.perf_counter() - t0
shard.eng.set_profiling(True)
run(args.steps)
ms, launches, active = shard.eng.update_stats()
print("chain=%d " % args.chain, end="")
print("world=%d shard: %d local columns, %.2f GB; %.3f ms per pivot (%.1f pivots/s); k_update %.3f ms avg (%.0f GB/s, %d active cols) -> non-update overhead %.3f ms" % (
    G, shard.eng.exchange_doubles() and (n + m + 1 + G - 1) // G, m * ((n + m + 1) // G + 1) * 8 / 1e9, el / args.steps * 1e3, args.steps / el,
    ms / launches, 16.0 * m * active / (ms / launches * 1e-3) / 1e9, active, el / args.steps * 1e3 - ms / launches))
dist.destroy_process_group()
