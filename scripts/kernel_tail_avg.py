"""Average duration of the LAST `--last` launches of every kernel in a rocprofv3 kernel trace
(`*_kernel_trace.csv`), for loops whose cost per iteration drifts as they run.

    python scripts/kernel_tail_avg.py <kernel_trace.csv> --last 20
"""
import argparse
import collections
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--last", type=int, default=20)
a = ap.parse_args()
runs = collections.defaultdict(list)
with open(a.trace, newline="") as f:
    for row in csv.DictReader(f):
        runs[row["Kernel_Name"]].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
out = []
for name, spans in runs.items():
    spans.sort()
    tail = spans[-a.last:] if len(spans) > a.last else spans
    avg = sum(e - s for s, e in tail) / len(tail) / 1e3
    short = re.sub(r"\(.*", "", name)
    out.append((avg, short, len(tail), len(spans)))
total = 0.0
for avg, name, n, n_all in sorted(out, reverse=True):
    print("%-70s last %4d of %6d launches: avg %10.1f us" % (name[:70], n, n_all, avg))
