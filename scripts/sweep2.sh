#!/bin/bash
# usage: sweep2.sh cfg "NT PREC UNROLL GRID" ...
cfg=$1; shift
for spec in "$@"; do
  set -- $spec
  LPAMD_UPDATE_NT=$1 LPAMD_UPDATE_PREC=$2 LPAMD_UPDATE_UNROLL=$3 LPAMD_UPDATE_GRID=$4 timeout -k 10 200 python -u bench.py --config $cfg --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/sweep.json 2>gpurun_out/sweep.err || { echo "FAILED $spec"; tail -3 gpurun_out/sweep.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/sweep.json'))
print('$cfg nt=$1 prec=$2 unroll=$3 grid=$4 :', d['value'], 'piv/s', d['roofline']['avg_launch_ms'], 'ms', d['roofline']['achieved'], 'GB/s', d['roofline']['frac'])"
done
