#!/usr/bin/env python3
"""Timeline of the MAX_REDUCTION set-up kernels (k_prb_rows, k_prb_bounds, k_prb_tail) over the steps of one pass, from
the wall-clock checkpoints of a library built with -DLPAMD_DBG_CLOCK (make dbg-clock; LPAMD_LIBRARY points at it):
microseconds relative to the moment k_prb_rows entered the step.

  LPAMD_LIBRARY=bin_scratch/liblpamd_dbgclock.so python scripts/prb_timeline.py [config]
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
m, n = bench.CONFIGS[config]
eng = lp.TableauEngine(m, n + m)
eng.fill_synthetic(n, 1)
depth = eng.loop_depth(lp.MAX_REDUCTION)
eng.run_pivots(lp.MAX_REDUCTION, depth)
eng.run_pivots(lp.MAX_REDUCTION, 2 * depth)
L = lp.engine.library()
buf = (ctypes.c_ulonglong * (32 * 4 * 8))()
L.lpamd_dbg_clock.restype = ctypes.c_int
assert L.lpamd_dbg_clock(buf) == 0
t = lambda j, who, cp: buf[(j * 4 + who) * 8 + cp]
print("config %s depth %d" % (config, depth))
print(" j | k_prb_rows: go minima tau collected ranked col' end | k_prb_bounds wg 0: entry rows-known tiles-done end | "
      "k_prb_tail row wg 0: entry alive selected evaluated polled decided end | column wg 0: entry alive selected evaluated polled decided scalars end | next step")
for j in range(1, depth):
    base = t(j, 0, 0)
    if not base:
        continue
    us = lambda who, cp: (t(j, who, cp) - base) / 100.0 if t(j, who, cp) else float("nan")
    nxt = t(j + 1, 0, 0) if j + 1 < depth else 0
    row = "%2d |" % j
    row += " ".join("%5.2f" % us(0, cp) for cp in range(1, 8)) + " | "
    row += " ".join("%5.2f" % us(1, cp) for cp in range(0, 4)) + " | "
    row += " ".join("%5.2f" % us(2, cp) for cp in (0, 1, 2, 3, 4, 5, 7)) + " | "
    row += " ".join("%5.2f" % us(3, cp) for cp in range(0, 8)) + " | "
    row += "%5.2f" % ((nxt - base) / 100.0 if nxt else float("nan"))
    print(row)
