#!/usr/bin/env python3
"""The table of DESIGN.md's measurement section from the bench lines in profiles/roundN/ (in brackets: the value of the same
file at a git revision, when one is given):  python scripts/profile_table.py profiles/round3 [REV]"""
import json
import subprocess
import sys

ROWS = [
    ("m=32768 n=65536 max_cost, default `bench.py` (depth measured at create)", "bench_default_m32768_n65536.json", "3351"),
    ("same, as the driver runs it (`--steps 20 --warmup 5`: 3 timed passes)", "bench_default_steps20_warmup5.json", "3329 (one pass, 0.716)"),
    ("same under `rocprofv3 --kernel-trace --stats` (`…_kernel_stats.csv`)", "bench_default_m32768_n65536_under_rocprof.json", "0.708"),
    ("same, `LPAMD_CHAIN=32`", "bench_32_pivots_per_pass_m32768_n65536.json", "3787"),
    ("same, `LPAMD_CHAIN=16`: col' in registers", "bench_16_pivots_per_pass_m32768_n65536.json", "2663"),
    ("same, `LPAMD_CHAIN=1`", "bench_one_pivot_per_pass_m32768_n65536.json", "180.3"),
    ("same, max_reduction", "bench_max_reduction_m32768_n65536.json", "2730"),
    ("m=4096 n=8192 max_cost (0.40 GB stored: Infinity-Cache size, col' in registers)", "bench_m4096_n8192_max_cost.json", "50 107"),
    ("same, max_reduction", "bench_m4096_n8192_max_reduction.json", "16 291"),
    ("m=65536 n=262144 on ONE GPU (172 GB stored), max_cost", "bench_m65536_n262144_1gpu_max_cost.json", "445.8 (0.709)"),
    ("same, max_reduction (BASELINE config 5 on one GPU)", "bench_m65536_n262144_1gpu_max_reduction.json", "418.1"),
    ("rank-0 shard of an 8-GPU run of the default config, exchange inside the launches, per rank", "shard_of_8_max_cost.json", "18 512"),
    ("same with the all-gather in front of every step", "shard_of_8_max_cost_collective.json", ""),
    ("rank-0 shard of 4", "shard_of_4_max_cost.json", "11 446"),
    ("rank-0 shard of 2", "shard_of_2_max_cost.json", "6431"),
    ("rank-0 shard of 8, max_reduction, all-gathers over the receive buffers", "shard_of_8_max_reduction.json", "10 354"),
    ("same with RCCL's all-gather (a device copy with a world of one)", "shard_of_8_max_reduction_collective.json", ""),
    ("rank-0 shard of config 5 on 8 GPUs (m=65536 n=262144, max_reduction)", "shard_of_8_m65536_n262144_max_reduction.json", "2682"),
]


def main():
    where = sys.argv[1]
    rev = sys.argv[2] if len(sys.argv) > 2 else None
    print("| config | file in `%s/` | pivots/s | pivots per pass | ms/pivot | pass kernel | streamed GB/s (`frac`) | round 2 |" % where)
    print("|---|---|---|---|---|---|---|---|")
    for label, name, round2 in ROWS:
        d = json.load(open("%s/%s" % (where, name)))
        before = ""
        if rev:
            try:
                before = " (%.0f)" % json.loads(subprocess.check_output(["git", "show", "%s:%s/%s" % (rev, where, name)]))["value"]
            except Exception:
                pass
        r = d["roofline"]
        print("| %s | `%s` | **%.0f**%s | %s | %.4f | %.3f ms × %d | %.0f (%.3f) | %s |" % (
            label, name, d["value"], before, d["config"]["pivots_per_pass"], d["ms_per_step"], r["avg_launch_ms"], r["launches"],
            r["achieved"], r["frac"], round2))


if __name__ == "__main__":
    main()
