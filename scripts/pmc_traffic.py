"""Reduces two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one run each, as the hardware
cannot hold both with their derived counters) to HBM bytes per launch of the pass kernel.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 32 --warmup 16 --no-cpu-baseline > fetch.json
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 32 --warmup 16 --no-cpu-baseline > write.json
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write fetch.json > profiles/roundN/pmc_hbm_traffic_<config>.json

FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced reads and is doubled.  The bench line of one of the two runs supplies what a launch is
meant to move (streamed / algorithmic bytes per launch, pivots per launch)."""
import csv
import glob
import json
import os
import sys


def per_launch(directory, counter):
    vals = []
    name = None
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and "k_update" in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
                name = row["Kernel_Name"].split("(")[0]
    return (sum(vals) / len(vals) if vals else None), len(vals), name


def main():
    fetch_dir, write_dir, bench_json = sys.argv[1:4]
    line = [l for l in open(bench_json).read().splitlines() if l.startswith("{")][-1]
    bench = json.loads(line)
    f, nf, name = per_launch(fetch_dir, "FETCH_SIZE")
    w, nw, _ = per_launch(write_dir, "WRITE_SIZE")
    read_b, write_b = 2.0 * f * 1024, w * 1024
    roof = bench["roofline"]
    streamed, algo = roof["streamed_bytes_per_launch"], roof["algorithmic_bytes_per_launch"]
    print(json.dumps({
        "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps %d --warmup %d --no-cpu-baseline%s (one pass per counter)" % (
            bench["steps_requested"], bench["warmup_requested"],
            "" if bench["config"]["strategy"] == "max_cost" else " --strategy " + bench["config"]["strategy"]),
        "config": bench["config"]["workload"], "strategy": bench["config"]["strategy"], "kernel": name,
        "note": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read, so it is doubled",
        "FETCH_SIZE_avg_KiB_per_launch": f, "FETCH_SIZE_launches": nf,
        "WRITE_SIZE_avg_KiB_per_launch": w, "WRITE_SIZE_launches": nw,
        "read_bytes_per_launch_corrected": read_b, "write_bytes_per_launch": write_b,
        "traffic_bytes_per_launch": read_b + write_b, "algorithmic_bytes_per_launch": algo,
        "pivots_per_launch": bench["pivots_per_pass"], "streamed_bytes_per_launch": streamed,
        "traffic_over_streamed": (read_b + write_b) / streamed,
        "traffic_over_algorithmic": (read_b + write_b) / algo}, indent=1))


if __name__ == "__main__":
    main()
