#!/usr/bin/env python3
"""Benchmark of the dense simplex-tableau hot path on MI355X.

One "step" = one simplex pivot (pricing + ratio test + row gather + rank-1
tableau update) of a synthetic dense LP that is generated on the device and is
resident in HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--strategy max_cost|max_reduction]

N == 1 runs the device-resident loop of the C ABI (lpamd_run_pivots); N > 1 is
launched by torch.distributed.run with one rank per GPU and runs the
column-sharded loop (RCCL all-gather of candidates + integer sum-all-reduce of
the pivot column, both on a second stream beside the rank-1 update for
max_cost).  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (m constraints, n structural variables)
    "synthetic_m4096_n8192": (4096, 8192),
    "synthetic_m32768_n65536": (32768, 65536),
    "synthetic_m65536_n262144": (65536, 262144),
    "synthetic_m1024_n2048": (1024, 2048),     # quick self-test size
    # shape studies of the update kernel (same dense bytes, different column length)
    "synthetic_m65536_n65536": (65536, 65536),
    "synthetic_m32768_n131072": (32768, 131072),
    "synthetic_m16384_n131072": (16384, 131072),
    "synthetic_m131072_n16384": (131072, 16384),
    "synthetic_m65536_n131072": (65536, 131072),
    "synthetic_m65536_n196608": (65536, 196608),
}
DEFAULT_CONFIG = "synthetic_m32768_n65536"
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak


def pmc_traffic(config, world, pivots_per_launch=1):
    """HBM bytes per k_update launch from the committed rocprofv3 PMC passes of this
    same command (profiles/*/pmc_hbm_traffic_*.json: FETCH_SIZE doubled as gfx950
    requires, plus WRITE_SIZE).  None when no pass exists for the configuration."""
    import glob
    if world != 1:
        return None
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_hbm_traffic_*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("config") == config and d.get("pivots_per_launch", 1) == pivots_per_launch:
            best = d.get("traffic_bytes_per_launch")
    return best


def cpu_baseline(m, n, seed, strategy_name, budget_s=20.0):
    """Times the CPU restatement (oracle/, OpenMP over columns) on the same
    synthetic instance: a bounded number of pivots from the same start."""
    import psutil
    from oracle import lp_oracle as lo

    need = (m * (n + m + 2) * 8) * 1.15
    note = ""
    if psutil.virtual_memory().available < need + 8e9:
        m, n = CONFIGS["synthetic_m4096_n8192"]
        note = " (host RAM too small for the full instance: m=%d n=%d used instead)" % (m, n)
    t0 = time.time()
    dt = lo.DenseTableau.synthetic(m, n, seed, trace_cap=4096)
    dt.s.skip_zero_cols = 1   # same work as the GPU path: unit columns are left alone
    dt.s.fast_basis = 1
    t_fill = time.time() - t0
    strat = lo.STRATEGY[strategy_name]
    dt.solve(strat, max_pivots=1)            # warm-up pivot (page faults, thread pool)
    done0 = dt.n_pivots
    t0 = time.time()
    k = 0
    while time.time() - t0 < budget_s and k < 4096:
        step = max(1, min(64, k))            # 1, 1, 2, 4, ... pivots between clock reads
        dt.solve(strat, max_pivots=dt.n_pivots + step)
        k = dt.n_pivots - done0
    el = time.time() - t0
    cores = lo.lib().lpo_num_threads()
    return {
        "value": round(k / el, 4), "unit": "pivots/s", "cores": cores, "kind": "port",
        "sample": "first %d pivots (after 1 warm-up) of the same synthetic LP m=%d n=%d seed=%d, %s, CPU restatement with OpenMP on %d threads; "
                  "tableau fill %.1f s not timed%s" % (k, m, n, seed, strategy_name, cores, t_fill, note),
        "update_GBps": round(16.0 * m * (n + 1) * k / max(dt.s.update_seconds, 1e-9) / 1e9, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default=DEFAULT_CONFIG, choices=sorted(CONFIGS))
    ap.add_argument("--strategy", default="max_cost", choices=["max_cost", "max_reduction"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default=os.environ.get("LPAMD_EXCHANGE", "lookahead"), choices=["lookahead", "allreduce", "broadcast"],
                    help="N > 1: how the shards exchange the entering column (see linear-programming_amd/sharded.py)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LPAMD_BENCH_FORCE_SHARDED=1 runs the multi-rank code path (RCCL collectives included)
    # even for a world of one, to rehearse it on a one-GPU box.
    force_sharded = os.environ.get("LPAMD_BENCH_FORCE_SHARDED", "0") == "1"
    use_dist = world > 1 or (force_sharded and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    lp = importlib.import_module("linear-programming_amd")
    m, n = CONFIGS[args.config]
    N = n + m
    strat = lp.STRATEGY[args.strategy]
    K, W = args.steps, args.warmup

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if world == 1 and not force_sharded:
        eng = lp.TableauEngine(m, N)
        eng.fill_synthetic(n, args.seed)
        run = lambda k: eng.run_pivots(strat, k)
    else:
        from importlib import import_module
        sharded = import_module("linear-programming_amd.sharded")
        shard = sharded.EngineShard(m, N, rank, world, dev)
        shard.eng.fill_synthetic(n, args.seed)
        eng = shard.eng
        solver = sharded.ShardedSolver(shard, exchange=args.exchange)
        run = lambda k: solver.run_pivots(strat, k)

    res, done = run(W)
    assert done == W and res == lp.LIMIT, "warm-up ended early: result %d after %d pivots" % (res, done)
    sync_all()
    # HIP events around every rank-1 update launch of the timed region, recorded on the engine's own
    # stream (two event records per pivot; read back after the region)
    eng.set_profiling(True)
    t0 = time.perf_counter()
    res, done = run(K)
    sync_all()
    el = time.perf_counter() - t0
    upd_ms, launches, active = eng.update_stats()
    last_pivots, pass_streamed, _ = eng.pass_stats()
    eng.set_profiling(False)
    # pivots per launch over the timed region (the last pass may be shorter when K is no multiple of the chain depth)
    pass_pivots = max(1, round(K / launches)) if launches and last_pivots > 0 else 1
    assert done == K and res == lp.LIMIT, "timed region ended early: result %d after %d pivots" % (res, done)
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    avg_ms = upd_ms / max(launches, 1)
    # algorithmic bytes: every entry of an active local column is read once and written once PER PIVOT
    # (16*m*n_active); a pass that applies a chain of pivots counts the columns of each of them
    # (`active`), while it streams the tableau only once (`pass_streamed` columns)
    algo_bytes = 16.0 * m * active
    streamed_bytes = 16.0 * m * pass_streamed
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if launches else 0.0
    if use_dist:
        t = torch.tensor([achieved], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)   # report the slowest rank's kernel rate
        achieved = float(t.item())

    out = {
        "metric": "simplex_pivots_per_sec",
        "value": round(K / el, 3),
        "unit": "pivots/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(el / K * 1e3, 5),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": args.config, "m": m, "n": n, "stored_columns": N + 1, "strategy": args.strategy,
                   "seed": args.seed, "sharding": "columns round-robin over %d GPU(s), RHS replicated" % world,
                   "exchange": (args.exchange if (world > 1 or force_sharded) else "none (device-resident loop)")},
        "update_GBps_per_gpu": round(achieved, 1),
        "update_GBps_aggregate": round(achieved * world, 1),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args.config, world, pass_pivots),
                     "kernel": "k_update_slab" if pass_pivots > 1 else "k_update", "avg_launch_ms": round(avg_ms, 5),
                     "launches": int(launches), "algorithmic_bytes_per_launch": algo_bytes, "active_columns": int(active),
                     "pivots_per_launch": int(pass_pivots), "streamed_columns": int(pass_streamed),
                     "streamed_bytes_per_launch": streamed_bytes,
                     "streamed_GBps": round(streamed_bytes / (avg_ms * 1e-3) / 1e9, 1) if launches else 0.0,
                     "note": "achieved = algorithmic bytes (16*m*n_active per pivot x pivots per launch) / launch time; a launch "
                             "applies a chain of pivots in ONE pass over the tableau, so the HBM traffic of a launch is "
                             "streamed_bytes_per_launch (PMC: traffic) and achieved may exceed the HBM peak"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(m, n, args.seed, args.strategy)
        except Exception as ex:   # the baseline must never cost the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "pivots/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
