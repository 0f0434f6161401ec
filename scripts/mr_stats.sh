# Per-kernel averages of the max-reduction bench under rocprofv3 (default shard unless arguments say otherwise).
export TMPDIR=/tmp; o=gpurun_out/mr; rm -rf $o; mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/st -- python3 bench.py --no-cpu-baseline --strategy max_reduction "$@" > $o/bench.json 2> $o/err.log
find $o/st -name "*kernel_stats.csv" -exec cp {} $o/kernel_stats.csv \; ; rm -rf $o/st
python3 -c "
import csv,json
print(json.load(open('gpurun_out/mr/bench.json'))['value'])
for r in csv.DictReader(open('gpurun_out/mr/kernel_stats.csv')):
    print(r['Name'].split('(')[0][-40:], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
"
