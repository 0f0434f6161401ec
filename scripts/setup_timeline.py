#!/usr/bin/env python3
"""Timeline of the set-up kernels of one chained pass, from the wall-clock checkpoints of a library built with
-DLPAMD_DBG_CLOCK (make dbg-clock; LPAMD_LIBRARY points at it).  Prints, per step of the last pass, the
checkpoints in microseconds relative to the entry of k_chain_ratio of that step.

  LPAMD_LIBRARY=bin_scratch/liblpamd_dbgclock.so python scripts/setup_timeline.py [config] [shard_of]
"""
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

lp = importlib.import_module("linear-programming_amd")
config = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_CONFIG
shard_of = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m, n = bench.CONFIGS[config]
N = n + m
if shard_of > 1:
    group = lp.ShardGroup(m, N, shard_of, devices=[0] * shard_of, rehearse_rank0=True)
    group.fill_synthetic(n, 1)
    depth = group.chain()
    run = lambda k: group.run_pivots(lp.MAX_COST, k)
else:
    eng = lp.TableauEngine(m, N)
    eng.fill_synthetic(n, 1)
    depth = eng.loop_depth(lp.MAX_COST)
    run = lambda k: eng.run_pivots(lp.MAX_COST, k)
run(depth)
run(2 * depth)
L = lp.engine.library()
buf = (ctypes.c_ulonglong * (32 * 4 * 8))()
L.lpamd_dbg_clock.restype = ctypes.c_int
assert L.lpamd_dbg_clock(buf) == 0
t = lambda j, who, cp: buf[(j * 4 + who) * 8 + cp]
print("config %s shard_of %d depth %d; microseconds after the entry of the step's first set-up kernel" % (config, shard_of, depth))
print(" j | k_chain_ratio: first load back | gating state and pricing partials back")
for j in range(1, depth):
    base = t(j, 0, 0)
    if base:
        print("%2d | %5.2f | %5.2f" % (j, (t(j, 0, 7) - base) / 100.0, (t(j, 0, 4) - base) / 100.0))
print(" j | ratio: gate+fold  loop  end | finalize: entry  r2  a2  rows_end | cols: entry r2 a2 d2 loop end | last wg: entry loop_end | next entry")
for j in range(1, depth):
    base = t(j, 0, 0) or t(j, 2, 0)
    us = lambda who, cp: (t(j, who, cp) - base) / 100.0 if t(j, who, cp) else float("nan")
    nxt = (t(j + 1, 0, 0) or t(j + 1, 2, 0)) if j + 1 < depth else 0
    print("%2d | %5.2f %5.2f %5.2f | %5.2f %5.2f %5.2f %5.2f | %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f | %5.2f %5.2f | %5.2f" % (
        j, us(0, 1), us(0, 2), us(0, 3), us(1, 0), us(1, 1), us(1, 2), us(1, 3),
        us(2, 0), us(2, 1), us(2, 2), us(2, 3), us(2, 4), us(2, 5), us(3, 0), us(3, 1), (nxt - base) / 100.0 if nxt else float("nan")))
