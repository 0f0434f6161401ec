#!/usr/bin/env python3
"""Benchmark of the dense simplex-tableau hot path on MI355X.

One "step" = one simplex pivot (pricing + ratio test + row gather + rank-1
tableau update) of a synthetic dense LP that is generated on the device and is
resident in HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--strategy max_cost|max_reduction]

N == 1 runs the device-resident loop of the C ABI (lpamd_run_pivots).  N > 1 runs the
column-sharded loop of the library (lpamd_group_*: chains of pivots per pass over the shard; the
ranks trade their candidates inside the launches over xGMI, or with one RCCL all-gather per pivot), one process per GPU: either the driver starts the ranks with
torch.distributed.run, or a plain `python bench.py --gpus N` starts them itself (as a child
torch.distributed.run, before this process touches a GPU).  Rank 0 prints ONE JSON line.

The loops apply a chain of pivots per pass over the tableau, so the warm-up and the timed region
are rounded UP to whole passes, and the timed region to at least three passes (the roofline figure
is an average over the launches of the pass kernel); the line reports the pivots actually timed
("steps") next to the request ("steps_requested").  A rank that cannot join the communicator of an
N-GPU run within four minutes prints a JSON line with an "error" key and exits non-zero.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
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
RMW_CEILING_GBS = 6360.0   # measured: a bare read-modify-write stream over two HBM placement classes (scripts/placement_microbench.hip)
INFINITY_CACHE_BYTES = 256 << 20
MIN_TIMED_PASSES = 3     # the roofline figure is an average over at least this many launches of the pass kernel
JOIN_TIMEOUT_S = 240     # creating the communicator of an N-GPU run may not hang the benchmark
FP64_VECTOR_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12   # one fp64 add or multiply per lane and clock, 16 lanes per SIMD


def pmc_traffic_ratio(config, world, chained, strategy):
    """HBM bytes the update kernel really moved per byte it is meant to stream, from the committed
    rocprofv3 PMC passes of this same command (profiles/*/pmc_hbm_traffic_*.json: FETCH_SIZE doubled
    as gfx950 requires, plus WRITE_SIZE; one pass per counter).  The newest file for the
    configuration, the pricing strategy and the kind of pass (chain of pivots or one pivot) wins; None if
    there is none."""
    import glob
    if world != 1:
        return None, None
    best, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_hbm_traffic_*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if (d.get("config") == config and d.get("strategy", "max_cost") == strategy
                and (d.get("pivots_per_launch", 1) > 1) == chained and d.get("traffic_over_streamed")):
            best, src = float(d["traffic_over_streamed"]), os.path.relpath(f, ROOT)
    return best, src


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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n_gpus, argv, port):
    """The command that starts one rank per GPU for a plain `python bench.py --gpus N`."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def round_up(x, q):
    return (x + q - 1) // q * q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default=DEFAULT_CONFIG, choices=sorted(CONFIGS))
    ap.add_argument("--strategy", default="max_cost", choices=["max_cost", "max_reduction"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-steps", action="store_true", help="time exactly --steps pivots even when that ends inside a pass")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # Not started by a launcher: start one rank per GPU as a child, before anything here touches a GPU.
        sys.exit(subprocess.call(launch_command(args.gpus, sys.argv[1:], free_port())))

    # The one JSON line must be the only thing on stdout, but libraries print there too (RCCL writes a version
    # banner when a communicator is created): keep the real stdout aside and send everything else to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LPAMD_BENCH_FORCE_SHARDED=1 runs the sharded loop (RCCL all-gather included) for a world of one, to
    # rehearse it on a one-GPU box; LPAMD_BENCH_SHARD_OF=w times rank 0 of a w-GPU run alone (its shard, real
    # RCCL launches, no wire time; the pivots are not the LP's, the work per pivot is that rank's).
    shard_of = int(os.environ.get("LPAMD_BENCH_SHARD_OF", "0"))
    force_sharded = os.environ.get("LPAMD_BENCH_FORCE_SHARDED", "0") == "1" or shard_of > 1
    # LPAMD_BENCH_FORCE_DIST=1 (under torch.distributed.run with one rank): the process-per-GPU path of N > 1 with a
    # world of one: torch's process group, the communicator id broadcast, ShardGroup.join.
    use_dist = world > 1 or (os.environ.get("LPAMD_BENCH_FORCE_DIST", "0") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    lp = importlib.import_module("linear-programming_amd")
    m, n = CONFIGS[args.config]
    N = n + m
    strat = lp.STRATEGY[args.strategy]

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def fail(message):
        """One JSON line that says what went wrong (stdout of rank 0, stderr of the others), then a hard exit."""
        line = json.dumps({"metric": "simplex_pivots_per_sec", "value": None, "unit": "pivots/s", "n_gpus": world, "rank": rank,
                           "error": message}) + "\n"
        os.write(real_stdout if rank == 0 else 2, line.encode())
        os._exit(3)

    group = None
    link = False
    rccl_ranks = 0
    if world == 1 and not force_sharded and not use_dist:
        eng = lp.TableauEngine(m, N)
        eng.fill_synthetic(n, args.seed)
        run = lambda k: eng.run_pivots(strat, k)
        depth = eng.loop_depth(strat)
        exchange = "none (device-resident loop)"
    else:
        def form_group():
            """The group of this run: joined over RCCL (one process per GPU), or a rehearsal on this GPU alone."""
            if not use_dist:
                if shard_of > 1:
                    return lp.ShardGroup(m, N, shard_of, devices=[local_rank] * shard_of, rehearse_rank0=True)
                return lp.ShardGroup(m, N, 1, devices=[local_rank])
            import threading
            ids = [lp.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            watchdog = threading.Timer(JOIN_TIMEOUT_S, fail, ["rank %d did not get through ncclCommInitRank within %d s (another rank "
                                                              "may have failed before it, or RCCL cannot reach its peers)" % (rank, JOIN_TIMEOUT_S)])
            watchdog.daemon = True
            watchdog.start()
            try:
                grp = lp.ShardGroup.join(m, N, rank, world, ids[0])
            except Exception as ex:
                fail("rank %d could not create its shard or join the communicator: %s" % (rank, ex))
            watchdog.cancel()
            if grp.comm_ranks() != world:
                fail("the communicator holds %d ranks, WORLD_SIZE is %d" % (grp.comm_ranks(), world))
            return grp

        group = form_group()
        rccl_ranks = group.comm_ranks()
        group.fill_synthetic(n, args.seed)
        eng = group.engine(0)
        run = lambda k: group.run_pivots(strat, k)
        depth = min(group.chain(), eng.loop_depth(strat))   # MAX_REDUCTION passes may be shorter than MAX_COST ones
        link = group.link_active()
        exchange = ("inside a pass every rank stores its slot (record + candidate column + its ratio-test partials, tagged words) into the "
                    "peers' receive buffers from inside the step's own launch, no collective; at the pass boundary one all-gather over "
                    "the same buffers (one launch) beside the pass kernel" if link else
                    "one RCCL all-gather per pivot (record + candidate column + its ratio-test partials per rank), inside the library")

    K, W = args.steps, args.warmup
    if not args.exact_steps:
        K, W = max(round_up(max(K, 1), depth), MIN_TIMED_PASSES * depth), round_up(W, depth)
    link_fallback = None
    if link:
        # The exchange inside the launches has a self-test behind it but no run on this machine yet: two passes decide.  If any
        # rank's do not come through, every rank forms a new group that keeps the collective for every step, and says so.
        try:
            res, done = run(2 * depth)
            good = done == 2 * depth and res == lp.LIMIT
            why = "result %d after %d pivots" % (res, done)
        except Exception as ex:
            good, why = False, str(ex)
        everybody = good
        if use_dist:
            flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            everybody = int(flag.item()) == 1
        if not everybody:
            link_fallback = "the exchange inside the launches failed its trial passes on some rank (rank %d: %s): RCCL all-gather per pivot instead" % (
                rank, why if not good else "fine here")
            group.close()
            os.environ["LPAMD_SHARD_LINK"] = "0"
            group = form_group()
            group.fill_synthetic(n, args.seed)
            eng = group.engine(0)
            link = group.link_active()
            exchange = "one RCCL all-gather per pivot (record + candidate column + its ratio-test partials per rank), inside the library"
    exchange_trial = None
    trial_both = os.environ.get("LPAMD_BENCH_TRIAL_BOTH", "1")      # "0": never; "force": also in the one-GPU rehearsals
    if link and ((use_dist and world > 1 and trial_both != "0") or trial_both == "force"):
        # Which exchange is faster on THIS node is a measurement, not an assumption (the stores into the peers' buffers have
        # been timed between processes on one GPU, never over xGMI): three passes of each, a second group beside the first,
        # the slower one is closed.  Every rank takes the same decision from the slowest rank's times.
        def timed(grp, k):
            sync_all()
            t = time.perf_counter()
            try:
                r, d = grp.run_pivots(strat, k)
            except Exception:
                r, d = lp.ERROR, -1
            sync_all()
            t = time.perf_counter() - t
            both = torch.tensor([t, 0.0 if (d == k and r == lp.LIMIT) else 1.0], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(both, op=dist.ReduceOp.MAX)
            return float(both[0].item()), float(both[1].item()) == 0.0
        t_link, ok_link = timed(group, 3 * depth)
        os.environ["LPAMD_SHARD_LINK"] = "0"
        other = form_group()
        os.environ["LPAMD_SHARD_LINK"] = "1"
        other.fill_synthetic(n, args.seed)
        timed(other, depth)
        t_coll, ok_coll = timed(other, 3 * depth)
        exchange_trial = {"pivots": 3 * depth, "inside_the_launches_ms": round(t_link * 1e3, 3), "collective_ms": round(t_coll * 1e3, 3)}
        if ok_coll and (not ok_link or t_coll < 0.97 * t_link):
            group.close()
            group = other
            eng = group.engine(0)
            link = group.link_active()
            exchange = "one RCCL all-gather per pivot (record + candidate column + its ratio-test partials per rank), inside the library"
            exchange_trial["kept"] = "collective"
        else:
            other.close()
            exchange_trial["kept"] = "inside the launches"
    res, done = run(W)
    assert done == W and res == lp.LIMIT, "warm-up ended early: result %d after %d pivots" % (res, done)
    sync_all()
    # HIP events around every rank-1 update launch of the timed region, recorded on the engine's own stream
    eng.set_profiling(True)
    t0 = time.perf_counter()
    res, done = run(K)
    sync_all()
    el = time.perf_counter() - t0
    upd_ms, passes, pivots, streamed_cols, touched_cols = eng.profile_totals()
    priced, evaluated = eng.pricing_stats()
    eng.set_profiling(False)
    assert done == K and res == lp.LIMIT, "timed region ended early: result %d after %d pivots" % (res, done)
    assert pivots == K or passes == 0, "device totals (%d pivots) disagree with the %d pivots timed" % (pivots, K)
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # HBM traffic of the pass kernel: every streamed column is read once and written once per PASS (16*m bytes);
    # the algorithmic figure counts 16*m bytes per column a PIVOT touches, i.e. once per pivot of the chain.
    streamed_bytes = 16.0 * m * streamed_cols
    algo_bytes = 16.0 * m * touched_cols
    secs = max(upd_ms, 1e-9) * 1e-3
    achieved = streamed_bytes / secs / 1e9 if passes else 0.0
    per_rank = None
    if use_dist:
        mine = torch.tensor([achieved, upd_ms / max(passes, 1), float(streamed_cols) / max(passes, 1)], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": k, "achieved_GBps": round(float(v[0]), 1), "frac": round(float(v[0]) / HBM_PEAK_GBS, 4),
                     "avg_launch_ms": round(float(v[1]), 5), "streamed_columns_per_launch": round(float(v[2]), 1)} for k, v in enumerate(every)]
        achieved = min(float(v[0]) for v in every)   # the line's figure is the slowest rank's kernel rate
    replicas_agree = None
    if use_dist:
        # every rank walks the same pivots on its replica of the state: a digest of each rank's trace says whether they did
        import hashlib
        digest = hashlib.sha256(repr(eng.trace()).encode()).digest()[:8]
        mine_d = torch.tensor([int.from_bytes(digest, "little") >> 1], dtype=torch.int64, device=dev)
        all_d = [torch.zeros_like(mine_d) for _ in range(world)]
        dist.all_gather(all_d, mine_d)
        replicas_agree = all(int(v.item()) == int(all_d[0].item()) for v in all_d)
    chained = passes > 0 and pivots > passes
    ratio, ratio_src = pmc_traffic_ratio(args.config, world, chained, args.strategy)
    per_launch = streamed_bytes / max(passes, 1)

    out = {
        "metric": "simplex_pivots_per_sec",
        "value": round(K / el, 3),
        "unit": "pivots/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "steps_requested": args.steps,
        "warmup_requested": args.warmup,
        "ms_per_step": round(el / K * 1e3, 5),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": args.config, "m": m, "n": n, "stored_columns": N + 1, "strategy": args.strategy,
                   "seed": args.seed, "sharding": "columns round-robin over %d GPU(s), RHS replicated" % world,
                   "exchange": exchange, "exchange_in_launch": bool(group is not None and link),
                   **({"exchange_fallback": link_fallback} if link_fallback else {}),
                   **({"exchange_trial": exchange_trial} if exchange_trial else {}),
                   **({"replicas_walked_the_same_pivots": replicas_agree} if replicas_agree is not None else {}),
                   "pivots_per_pass": depth, "rccl_ranks": rccl_ranks,
                   **({"rehearsal": "rank 0 of a %d-GPU run alone on one GPU" % shard_of} if shard_of > 1 else {})},
        "update_GBps_per_gpu": round(achieved, 1),
        "update_GBps_aggregate": round(achieved * world, 1),
        "pivots_per_pass": round(pivots / max(passes, 1), 3),
        "algorithmic_GBps": round(algo_bytes / secs / 1e9, 1) if passes else 0.0,
        **({"max_reduction_pricing": {"pricings": priced, "columns_evaluated_exactly": evaluated,
                                      "note": "since the engine was created (warm-up included)"}} if priced else {}),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "frac_of_measured_rmw_ceiling": round(achieved / RMW_CEILING_GBS, 4),
                     "applicable_roof": ("hbm" if 8.0 * m * (N + 2) / world > 2 * INFINITY_CACHE_BYTES else
                                         "infinity cache (the shard of %.0f MB fits or nearly fits the 256 MiB memory-side cache: the HBM peak is "
                                         "not what bounds this stream, and at this size the set-up of the pivots takes more of a pass than "
                                         "the stream does)" % (8.0 * m * (N + 2) / world / 1e6)),
                     **({"per_rank": per_rank} if per_rank else {}),
                     "traffic": round(ratio * per_launch) if ratio else None,
                     "traffic_source": ratio_src,
                     "kernel": ("k_update_deep" if pivots > 16 * passes else "k_update_slab") if chained else "k_update",
                     "avg_launch_ms": round(upd_ms / max(passes, 1), 5),
                     "launches": int(passes), "streamed_bytes_per_launch": per_launch,
                     "algorithmic_bytes_per_launch": algo_bytes / max(passes, 1),
                     "fp64_vector": {"achieved_Tops": round(2.0 * m * touched_cols / secs / 1e12, 2) if passes else 0.0,
                                     "peak_Tops": FP64_VECTOR_PEAK_TOPS, "frac": round(2.0 * m * touched_cols / secs / 1e12 / FP64_VECTOR_PEAK_TOPS, 4) if passes else 0.0,
                                     "note": "separately rounded multiply + add per entry and chained pivot (no FMA: the reference rounds twice); "
                                             "peak = 256 CUs x 4 SIMDs x 16 fp64 lanes per clock x 2.4 GHz; beyond about 20 pivots per pass this "
                                             "resource shows in the pass time beside HBM (plus one v_min3_f32 per two entries and step for the 1e-6 drop)"},
                     "note": "achieved = HBM bytes one launch streams (16*m per column it reads and writes, once per pass) / its "
                             "duration (HIP events on the engine's stream); a launch applies pivots_per_pass pivots to every entry, "
                             "so the per-pivot algorithmic bytes (16*m*n_active per pivot) give algorithmic_GBps, which may exceed the "
                             "peak; traffic = PMC bytes per launch (committed profile, scaled to this run's streamed columns)"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(m, n, args.seed, args.strategy)
        except Exception as ex:   # the baseline must never cost the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "pivots/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if group is not None:
        group.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
