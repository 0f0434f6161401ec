#!/bin/bash
# bench + rocprofv3 kernel stats + the two PMC passes of the default configuration (run on the GPU box)
set -e
export TMPDIR=/tmp
o=gpurun_out/prof_chain
rm -rf $o; mkdir -p $o
timeout -k 10 300 python3 -u bench.py > $o/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --no-cpu-baseline > $o/bench_under_rocprof.json
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline > $o/pmc_fetch.json
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline > $o/pmc_write.json
python3 scripts/pmc_traffic.py $o/pmc_fetch $o/pmc_write synthetic_m32768_n65536 8 > $o/pmc_hbm_traffic_m32768_n65536.json
find $o/stats -name "*kernel_stats.csv" -exec cp {} $o/kernel_stats.csv \;
rm -rf $o/stats/*/*kernel_trace.csv $o/pmc_fetch $o/pmc_write
cat $o/bench_default.json; head -8 $o/kernel_stats.csv; cat $o/pmc_hbm_traffic_m32768_n65536.json
