#!/bin/bash
cfg=$1; nt=$2; grid=$3; shift 3
for u in "$@"; do
  LPAMD_UPDATE_NT=$nt LPAMD_UPDATE_GRID=$grid LPAMD_UPDATE_UNROLL=$u timeout -k 10 200 python -u bench.py --config $cfg --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/sweep.json 2>gpurun_out/sweep.err || { echo "FAILED"; tail -3 gpurun_out/sweep.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/sweep.json'))
print('$cfg nt=$nt grid=$grid unroll=$u', d['value'], 'piv/s', d['roofline']['avg_launch_ms'], 'ms', d['roofline']['achieved'], 'GB/s', d['roofline']['frac'])"
done
