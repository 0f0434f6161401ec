#!/bin/bash
# depth sweep of the LDS-resident pass, and its grid
set -e
export TMPDIR=/tmp
o=gpurun_out/r2p10
rm -rf $o; mkdir -p $o
for d in 18 20 22 24 28 32; do
  LPAMD_CHAIN=$d timeout -k 10 200 python3 -u bench.py --steps 256 --warmup 64 --no-cpu-baseline > $o/d${d}.json 2> $o/sweep.err
  python3 -c "import json;d=json.load(open('$o/d${d}.json'));print('depth',$d,d['value'],'pivots/s',d['roofline']['avg_launch_ms'],'ms/pass',d['roofline']['achieved'],'GB/s',d['pivots_per_pass'])" | tee -a $o/sweep.txt
done
for g in 1024 4096 8192; do
  LPAMD_CHAIN=24 LPAMD_DEEP_GRID=$g timeout -k 10 200 python3 -u bench.py --steps 192 --warmup 48 --no-cpu-baseline > $o/g${g}.json 2> $o/sweep.err
  python3 -c "import json;d=json.load(open('$o/g${g}.json'));print('depth 24 grid',$g,d['value'],'pivots/s',d['roofline']['avg_launch_ms'],'ms/pass')" | tee -a $o/sweep.txt
done
for d in 16 20 24 32; do
  LPAMD_CHAIN=$d timeout -k 10 200 python3 -u bench.py --strategy max_reduction --steps 256 --warmup 64 --no-cpu-baseline > $o/mr_d${d}.json 2> $o/sweep.err
  python3 -c "import json;d=json.load(open('$o/mr_d${d}.json'));print('max_reduction depth',$d,d['value'],'pivots/s',d['roofline']['avg_launch_ms'],'ms/pass')" | tee -a $o/sweep.txt
done
