# Per-kernel averages of rank 0 of an N-GPU run alone on one GPU (N = first argument, default 8) under rocprofv3.
export TMPDIR=/tmp; w=${1:-8}; shift || true; o=gpurun_out/sh$w; rm -rf $o; mkdir -p $o
LPAMD_BENCH_SHARD_OF=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/st -- python3 bench.py --no-cpu-baseline --steps 512 --warmup 64 "$@" > $o/bench.json 2> $o/err.log
find $o/st -name "*kernel_stats.csv" -exec cp {} $o/kernel_stats.csv \; ; rm -rf $o/st
python3 -c "
import csv,json
d=json.load(open('$o/bench.json')); print(d['value'], d['ms_per_step'])
for r in csv.DictReader(open('$o/kernel_stats.csv')):
    print(r['Name'].split('(')[0][-44:], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
"
