#!/bin/bash
# Rank-0 rehearsal: all-gathers over the receive buffers (one launch) against RCCL's (a device copy with a world of one).
out=gpurun_out/linkgather; mkdir -p $out; : > $out/sweep.txt
for lg in 0 1; do
  for args in "--steps 520 --warmup 78" "--strategy max_reduction --steps 264 --warmup 66" "--config synthetic_m65536_n262144 --strategy max_reduction --steps 78 --warmup 26"; do
    LPAMD_LINK_GATHER=$lg LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>$out/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gather over the link $lg |', '$args', '|', d['value'], 'pivots/s', d['ms_per_step'], 'ms/pivot', d['roofline']['avg_launch_ms'], 'ms/pass')" | tee -a $out/sweep.txt
  done
done
