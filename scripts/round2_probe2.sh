#!/bin/bash
# correctness of the reworked chain kernels, then depth x slab-shape sweep of the default configuration
set -e
export TMPDIR=/tmp
o=gpurun_out/r2p2
rm -rf $o; mkdir -p $o
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $o/tests.log 2>&1 || { tail -40 $o/tests.log; exit 1; }
tail -2 $o/tests.log
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 100 --timeout-method thread > $o/noalloc_alone.log 2>&1 || { tail -30 $o/noalloc_alone.log; exit 1; }
tail -1 $o/noalloc_alone.log
for d in 8 12 16; do for sh in 1 3 4 5; do
  LPAMD_CHAIN=$d LPAMD_SLAB=$sh timeout -k 10 200 python3 -u bench.py --steps 96 --warmup 48 --no-cpu-baseline > $o/sweep_d${d}_s${sh}.json 2> $o/sweep.err
  python3 -c "import json;d=json.load(open('$o/sweep_d${d}_s${sh}.json'));print('depth',$d,'shape',$sh,d['value'],'pivots/s',d['roofline']['avg_launch_ms'],'ms/pass',d['roofline']['achieved'],'GB/s')" | tee -a $o/sweep.txt
done; done
