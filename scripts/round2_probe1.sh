#!/bin/bash
# no-alloc test alone in a fresh process; rehearsal of the sharded loop (rank 0 of N GPUs, RCCL world of one)
set -e
export TMPDIR=/tmp
o=gpurun_out/r2p1
rm -rf $o; mkdir -p $o
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 100 --timeout-method thread > $o/noalloc_alone.log 2>&1
tail -2 $o/noalloc_alone.log
for w in 8 4 2; do
  LPAMD_BENCH_SHARD_OF=$w timeout -k 10 200 python3 -u bench.py --steps 256 --warmup 16 --no-cpu-baseline > $o/shard_of_$w.json 2> $o/shard_of_$w.err
  python3 -c "import json;d=json.load(open('$o/shard_of_$w.json'));print($w,d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['achieved'])"
done
LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats8 -- python3 bench.py --steps 256 --warmup 16 --no-cpu-baseline > $o/shard_of_8_rocprof.json 2> $o/rocprof8.err
find $o/stats8 -name "*kernel_stats.csv" -exec cp {} $o/shard_of_8_kernel_stats.csv \;
rm -rf $o/stats8
head -14 $o/shard_of_8_kernel_stats.csv
LPAMD_BENCH_FORCE_SHARDED=1 timeout -k 10 200 python3 -u bench.py --no-cpu-baseline > $o/sharded_world1.json 2> $o/sharded_world1.err
python3 -c "import json;d=json.load(open('$o/sharded_world1.json'));print('world1 sharded',d['value'],d['ms_per_step'])"
