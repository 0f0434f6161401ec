#!/usr/bin/env python3
"""Timeline of k_shard_step over the steps of one pass of the rank-0 shard of an N-GPU deal (rehearsal on one GPU), from
the wall-clock checkpoints of a library built with -DLPAMD_DBG_CLOCK (make dbg-clock): microseconds relative to the
moment row workgroup 0 entered the step.

  LPAMD_LIBRARY=bin_scratch/liblpamd_dbgclock.so python scripts/shard_timeline.py [shards] [config]
"""
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

lp = importlib.import_module("linear-programming_amd")
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
config = sys.argv[2] if len(sys.argv) > 2 else bench.DEFAULT_CONFIG
m, n = bench.CONFIGS[config]
g = lp.ShardGroup(m, n + m, shards, devices=[0] * shards, rehearse_rank0=True)
g.fill_synthetic(n, 1)
depth = g.chain()
g.run_pivots(lp.MAX_COST, depth)
g.run_pivots(lp.MAX_COST, 2 * depth)
L = lp.engine.library()
buf = (ctypes.c_ulonglong * (32 * 4 * 8))()
L.lpamd_dbg_clock.restype = ctypes.c_int
assert L.lpamd_dbg_clock(buf) == 0
t = lambda j, who, cp: buf[(j * 4 + who) * 8 + cp]
print("config %s rank 0 of %d, depth %d, exchange inside the launches: %s" % (config, shards, depth, g.link_active()))
print(" j | row wg 0: winner  pivot-known  col'-written  priced  column-rebuilt  end | col wg 0: entry winner pivot-known loop-done published | last col wg: entry loop-done published | next step")
for j in range(1, depth):
    base = t(j, 0, 0)
    if not base:
        continue
    us = lambda who, cp: (t(j, who, cp) - base) / 100.0 if t(j, who, cp) else float("nan")
    nxt = t(j + 1, 0, 0) if j + 1 < depth else 0
    print("%2d | %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f | %5.2f %5.2f %5.2f %5.2f %5.2f | %5.2f %5.2f %5.2f | %5.2f" % (
        j, us(0, 1), us(0, 2), us(0, 3), us(0, 4), us(0, 5), us(0, 6), us(2, 0), us(2, 1), us(2, 2), us(2, 3), us(2, 4),
        us(3, 0), us(3, 3), us(3, 4), (nxt - base) / 100.0 if nxt else float("nan")))
g.close()
