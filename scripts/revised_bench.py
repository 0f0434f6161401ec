"""Times the device-resident revised-simplex loop on the synthetic dense LP.

    python scripts/revised_bench.py --m 32768 --n 65536 --pivots 20 --warmup 3

Prints one JSON line: pivots per second over `pivots` iterations after `warmup`
(both go through lpamd_rev_solve with a pivot limit, so every iteration is the
full pricing + B^-1 column + ratio test + rank-1 update of the inverse)."""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lp = importlib.import_module("linear-programming_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--pivots", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seed", type=int, default=20240607)
    a = ap.parse_args()
    eng = lp.RevisedTableauEngine(a.m, a.n + a.m)
    eng.fill_synthetic(a.n, a.seed)
    res, done = eng.solve(max_pivots=a.warmup)
    assert res == lp.LIMIT and done == a.warmup, (res, done)
    t0 = time.perf_counter()
    res, done = eng.solve(max_pivots=a.warmup + a.pivots)
    dt = time.perf_counter() - t0
    k = done - a.warmup
    m, W = a.m, a.n + a.m + 1
    # HBM bytes every iteration streams on a dense LP: the pricing product reads the tableau once and
    # B^-1 * column reads the inverse once; c_B^T * B^-1 and the rank-1 update only touch the parts
    # with non-zero multipliers (up to 8 m^2 and 16 m^2 more when the inverse has filled in)
    dense_bytes = 8.0 * (m * W + m * m)
    n_upd, n_rows, n_dense = eng.list_sizes()
    print(json.dumps({"update_rows": n_upd, "pricing_rows": n_rows, "dense_rows": n_dense, "metric": "revised_simplex_pivots_per_second", "value": k / dt, "ms_per_pivot": 1e3 * dt / max(k, 1),
                      "pivots": k, "result": res, "m": a.m, "n": a.n, "dense_products_GB_per_pivot": dense_bytes / 1e9,
                      "dense_products_GBps": dense_bytes * k / dt / 1e9}))
    eng.close()


if __name__ == "__main__":
    main()
