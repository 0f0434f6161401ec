#!/bin/bash
set -e
export TMPDIR=/tmp
o=gpurun_out/r2p5
rm -rf $o; mkdir -p $o
timeout -k 10 600 python -u -m pytest tests/test_reduction_chain_gpu.py tests/test_chain_gpu.py tests/test_engine_parity.py -m gpu -x -q --timeout 300 --timeout-method thread > $o/tests.log 2>&1 || { tail -60 $o/tests.log; exit 1; }
tail -2 $o/tests.log
for cfg in synthetic_m32768_n65536 synthetic_m4096_n8192; do
  timeout -k 10 250 python3 -u bench.py --config $cfg --strategy max_reduction --steps 64 --warmup 16 --no-cpu-baseline > $o/mr_$cfg.json 2>> $o/err.log
  python3 -c "import json;d=json.load(open('$o/mr_$cfg.json'));print('$cfg max_reduction',d['value'],'pivots/s',d['roofline']['avg_launch_ms'],'ms/pass',d['roofline']['achieved'],'GB/s', d['pivots_per_pass'])"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --strategy max_reduction --no-cpu-baseline > $o/mr_under_rocprof.json 2>> $o/err.log
find $o/stats -name "*kernel_stats.csv" -exec cp {} $o/mr_kernel_stats.csv \;
rm -rf $o/stats
python3 - <<'PY'
import csv
for row in csv.DictReader(open('gpurun_out/r2p5/mr_kernel_stats.csv')):
    n=row['Name'].split('(')[0].replace('void ','').replace('lpamd::','')
    print('  %-40s calls %5s avg %10.1f us  total %8.2f ms'%(n[:40],row['Calls'],float(row['AverageNs'])/1e3,float(row['TotalDurationNs'])/1e6))
PY
