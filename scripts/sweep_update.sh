#!/bin/bash
# usage: sweep_update.sh <config> <nt> <grid...>
cfg=$1; nt=$2; shift 2
for g in "$@"; do
  LPAMD_UPDATE_NT=$nt LPAMD_UPDATE_GRID=$g timeout -k 10 200 python -u bench.py --config $cfg --steps 128 --warmup 8 --no-cpu-baseline > gpurun_out/sweep.json 2>gpurun_out/sweep.err || { echo "FAILED $cfg nt=$nt grid=$g"; tail -3 gpurun_out/sweep.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/sweep.json'))
print('$cfg nt=$nt grid=$g', d['value'], 'piv/s', d['roofline']['avg_launch_ms'], 'ms', d['roofline']['achieved'], 'GB/s', d['roofline']['frac'])"
done
