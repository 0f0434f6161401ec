#!/bin/bash
# chains of 24 and 32 pivots per pass: correctness of the chain tests, then a depth x kernel sweep
set -e
export TMPDIR=/tmp
o=gpurun_out/r2p7
rm -rf $o; mkdir -p $o
timeout -k 10 600 python -u -m pytest tests/test_chain_gpu.py tests/test_reduction_chain_gpu.py -m gpu -x -q --timeout 400 --timeout-method thread > $o/tests.log 2>&1 || { tail -40 $o/tests.log; exit 1; }
tail -2 $o/tests.log
for d in 16 24 32 40 48; do for deep in 0 8 4; do
  if [ $d -gt 32 ] && [ $deep -eq 0 ]; then continue; fi
  LPAMD_CHAIN=$d LPAMD_DEEP=$deep timeout -k 10 200 python3 -u bench.py --steps 192 --warmup 96 --no-cpu-baseline > $o/sweep_d${d}_deep${deep}.json 2> $o/sweep.err
  python3 -c "import json;d=json.load(open('$o/sweep_d${d}_deep${deep}.json'));print('depth',$d,'deep',$deep,d['value'],'pivots/s',d['roofline']['avg_launch_ms'],'ms/pass',d['roofline']['achieved'],'GB/s',d['pivots_per_pass'])" | tee -a $o/sweep.txt
done; done
