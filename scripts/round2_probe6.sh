#!/bin/bash
set -e
export TMPDIR=/tmp
o=gpurun_out/r2p6
rm -rf $o; mkdir -p $o
timeout -k 10 600 python -u -m pytest tests/test_reduction_chain_gpu.py tests/test_engine_parity.py -m gpu -x -q --timeout 300 --timeout-method thread > $o/tests.log 2>&1 || { tail -60 $o/tests.log; exit 1; }
tail -2 $o/tests.log
timeout -k 10 120 python3 scripts/prb_list_sizes.py 8192 16384 $(python3 -c "print(' '.join(['1']*40))") 16 16 16 16 2>&1 | tail -1
timeout -k 10 200 python3 scripts/prb_list_sizes.py 32768 65536 1 1 1 1 1 1 1 1 16 16 16 16 16 16 16 16 16 16 2>&1 | tail -1
for cfg in synthetic_m32768_n65536 synthetic_m4096_n8192; do
  timeout -k 10 250 python3 -u bench.py --config $cfg --strategy max_reduction --steps 128 --warmup 32 --no-cpu-baseline > $o/mr_$cfg.json 2>> $o/err.log
  python3 -c "import json;d=json.load(open('$o/mr_$cfg.json'));print('$cfg max_reduction',d['value'],'pivots/s',d['roofline']['avg_launch_ms'],'ms/pass',d['roofline']['achieved'],'GB/s', d['pivots_per_pass'], d.get('max_reduction_pricing'))"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --strategy max_reduction --steps 128 --warmup 32 --no-cpu-baseline > $o/mr_under_rocprof.json 2>> $o/err.log
find $o/stats -name "*kernel_stats.csv" -exec cp {} $o/mr_kernel_stats.csv \;
rm -rf $o/stats
python3 - <<'PY'
import csv
for row in csv.DictReader(open('gpurun_out/r2p6/mr_kernel_stats.csv')):
    n=row['Name'].split('(')[0].replace('void ','').replace('lpamd::','')
    print('  %-32s calls %5s avg %9.1f min %8.1f max %9.1f us  total %8.2f ms'%(n[:32],row['Calls'],float(row['AverageNs'])/1e3,float(row['MinNs'])/1e3,float(row['MaxNs'])/1e3,float(row['TotalDurationNs'])/1e6))
PY
for r in 16 24; do
  LPAMD_PRB_ROWS=$r timeout -k 10 250 python3 -u bench.py --strategy max_reduction --steps 128 --warmup 32 --no-cpu-baseline > $o/mr_rows$r.json 2>> $o/err.log
  python3 -c "import json;d=json.load(open('$o/mr_rows$r.json'));print('rows $r max_reduction',d['value'],'pivots/s', d.get('max_reduction_pricing'))"
done
