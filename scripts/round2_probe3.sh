#!/bin/bash
# per-kernel breakdown of the default loop (depth 16), other sizes, and the 8-GPU shard rehearsal
set -e
export TMPDIR=/tmp
o=gpurun_out/r2p3
rm -rf $o; mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --no-cpu-baseline > $o/bench_default_under_rocprof.json 2> $o/rocprof.err
find $o/stats -name "*kernel_stats.csv" -exec cp {} $o/bench_default_kernel_stats.csv \;
rm -rf $o/stats
head -12 $o/bench_default_kernel_stats.csv | cut -c1-150
for cfg in synthetic_m4096_n8192; do for d in 8 16; do for sh in 3 4 5; do
  LPAMD_CHAIN=$d LPAMD_SLAB=$sh timeout -k 10 200 python3 -u bench.py --config $cfg --steps 960 --warmup 96 --no-cpu-baseline > $o/${cfg}_d${d}_s${sh}.json 2>> $o/sweep.err
  python3 -c "import json;d=json.load(open('$o/${cfg}_d${d}_s${sh}.json'));print('$cfg depth',$d,'shape',$sh,d['value'],'pivots/s',d['roofline']['avg_launch_ms'],'ms/pass',d['roofline']['achieved'],'GB/s')" | tee -a $o/sweep.txt
done; done; done
for w in 8 2; do for d in 8 16; do for sh in 3 4; do
  LPAMD_BENCH_SHARD_OF=$w LPAMD_CHAIN=$d LPAMD_SLAB=$sh timeout -k 10 200 python3 -u bench.py --steps 512 --warmup 64 --no-cpu-baseline > $o/shard_of_${w}_d${d}_s${sh}.json 2>> $o/sweep.err
  python3 -c "import json;d=json.load(open('$o/shard_of_${w}_d${d}_s${sh}.json'));print('shard of',$w,'depth',$d,'shape',$sh,d['value'],'pivots/s',d['ms_per_step'],'ms/pivot',d['roofline']['avg_launch_ms'],'ms/pass',d['roofline']['achieved'],'GB/s')" | tee -a $o/sweep.txt
done; done; done
LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats8 -- python3 bench.py --steps 512 --warmup 64 --no-cpu-baseline > $o/shard_of_8_under_rocprof.json 2>> $o/rocprof.err
find $o/stats8 -name "*kernel_stats.csv" -exec cp {} $o/shard_of_8_kernel_stats.csv \;
rm -rf $o/stats8
head -14 $o/shard_of_8_kernel_stats.csv | cut -c1-150
