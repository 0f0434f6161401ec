#!/bin/bash
set -e
export TMPDIR=/tmp
o=gpurun_out/r2p4
rm -rf $o; mkdir -p $o
timeout -k 10 500 python -u -m pytest tests/test_chain_gpu.py tests/test_group_gpu.py tests/test_sharded_gpu.py tests/test_engine_parity.py tests/test_cli_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $o/tests.log 2>&1 || { tail -40 $o/tests.log; exit 1; }
tail -2 $o/tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --no-cpu-baseline > $o/bench_default_under_rocprof.json 2> $o/rocprof.err
find $o/stats -name "*kernel_stats.csv" -exec cp {} $o/bench_default_kernel_stats.csv \;
rm -rf $o/stats
cat $o/bench_default_under_rocprof.json | cut -c1-200
LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats8 -- python3 bench.py --steps 512 --warmup 64 --no-cpu-baseline > $o/shard_of_8_under_rocprof.json 2>> $o/rocprof.err
find $o/stats8 -name "*kernel_stats.csv" -exec cp {} $o/shard_of_8_kernel_stats.csv \;
rm -rf $o/stats8
for w in 8 4 2; do
  LPAMD_BENCH_SHARD_OF=$w timeout -k 10 200 python3 -u bench.py --steps 512 --warmup 64 --no-cpu-baseline > $o/shard_of_${w}.json 2>> $o/sweep.err
  python3 -c "import json;d=json.load(open('$o/shard_of_${w}.json'));print('shard of',$w,d['value'],'pivots/s',d['ms_per_step'],'ms/pivot',d['roofline']['avg_launch_ms'],'ms/pass',d['roofline']['achieved'],'GB/s')" | tee -a $o/sweep.txt
done
timeout -k 10 250 python3 -u bench.py > $o/bench_default.json 2>> $o/sweep.err
python3 scripts/kernel_tail_avg.py 2>/dev/null || true
python3 - <<'PY'
import csv
for f in ('bench_default_kernel_stats.csv','shard_of_8_kernel_stats.csv'):
    print(f)
    for row in csv.DictReader(open('gpurun_out/r2p4/'+f)):
        n=row['Name'].split('(')[0].replace('void ','').replace('lpamd::','')
        print('  %-40s calls %5s avg %10.1f us  total %8.2f ms'%(n[:40],row['Calls'],float(row['AverageNs'])/1e3,float(row['TotalDurationNs'])/1e6))
PY
