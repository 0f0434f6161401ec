#!/bin/bash
cfg=$1; shift
for pad in "$@"; do
  LPAMD_LDA_PAD=$pad timeout -k 10 200 python -u bench.py --config $cfg --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/sweep.json 2>gpurun_out/sweep.err || { echo FAILED; tail -3 gpurun_out/sweep.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/sweep.json'))
print('$cfg pad=$pad', d['value'], 'piv/s', d['roofline']['avg_launch_ms'], 'ms', d['roofline']['achieved'], 'GB/s', d['roofline']['frac'])"
done
