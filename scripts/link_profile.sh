#!/bin/bash
# Rank-0-of-8 rehearsal under rocprofv3 with and without the exchange inside the launch.
set -u
out=gpurun_out/linkprof
mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
for link in 0 1; do
  LPAMD_SHARD_LINK=$link LPAMD_BENCH_SHARD_OF=8 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/l$link -- python3 bench.py --no-cpu-baseline --steps 520 --warmup 78 > $out/bench_l$link.json 2> $out/err_l$link.log || { tail -5 $out/err_l$link.log; exit 1; }
  f=$(find $out/l$link -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/kernel_stats_l$link.csv
  echo "== link $link"; cut -c1-60 $out/bench_l$link.json; awk -F'","' 'NR>1 {n=$1; gsub(/"/,"",n); printf "%-60s %6s %10.1f\n", substr(n,1,60), $2, $4/1000}' $out/kernel_stats_l$link.csv | head -8
done
