#!/bin/bash
# Everything profiles/roundN/ holds, in one run on the GPU box (about 8 minutes):
#   bash scripts/profile_round.sh && cp gpurun_out/prof_round/* profiles/roundN/
# Prints a progress line per leg.
set -e
export TMPDIR=/tmp
o=gpurun_out/prof_round
rm -rf $o; mkdir -p $o
# The chain depth is measured when an engine is created (a few passes of the pass kernel per candidate depth).  The
# profiled runs take the depth the plain run chose and skip that probe, so that the kernel averages are those of the
# timed region's own launches.
depth_of() { python3 -c "import json,sys; print(json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])['config']['pivots_per_pass'])" "$1"; }
stats() {   # stats <name> <bench args...>: rocprofv3 kernel stats of one bench command
  local name=$1; shift
  echo "stats $name"
  LPAMD_DEPTH_PROBE=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/st_$name -- python3 bench.py --no-cpu-baseline "$@" > $o/${name}_under_rocprof.json 2>> $o/err.log
  find $o/st_$name -name "*kernel_stats.csv" -exec cp {} $o/${name}_kernel_stats.csv \;
  rm -rf $o/st_$name
}
pmc() {     # pmc <name> <bench args...>: the two PMC passes and their reduction
  local name=$1; shift
  echo "pmc $name"
  LPAMD_DEPTH_PROBE=0 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 44 --warmup 22 --no-cpu-baseline "$@" > $o/pmc_fetch.json 2>> $o/err.log
  LPAMD_DEPTH_PROBE=0 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 44 --warmup 22 --no-cpu-baseline "$@" > $o/pmc_write.json 2>> $o/err.log
  python3 scripts/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/pmc_fetch.json > $o/pmc_hbm_traffic_$name.json
  rm -rf $o/pmc_fetch $o/pmc_write $o/pmc_fetch.json $o/pmc_write.json
}
# the default line (with the CPU baseline) and the driver's own invocation
timeout -k 10 300 python3 -u bench.py > $o/bench_default_m32768_n65536.json 2>> $o/err.log
timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_default_steps20_warmup5.json 2>> $o/err.log
export LPAMD_CHAIN=$(depth_of $o/bench_default_m32768_n65536.json)
echo "depth chosen by the probe: $LPAMD_CHAIN"
stats bench_default_m32768_n65536
pmc m32768_n65536
unset LPAMD_CHAIN
# max-reduction, same configuration
timeout -k 10 300 python3 -u bench.py --strategy max_reduction --no-cpu-baseline > $o/bench_max_reduction_m32768_n65536.json 2>> $o/err.log
export LPAMD_CHAIN=$(depth_of $o/bench_max_reduction_m32768_n65536.json)
stats bench_max_reduction_m32768_n65536 --strategy max_reduction
pmc max_reduction_m32768_n65536 --strategy max_reduction
unset LPAMD_CHAIN
# one pivot per pass (the roofline of a single rank-1 update), 16 per pass (the longest chain that hides under the
# HBM time of the pass) and 32 per pass (bound by the fp64 vector rate)
LPAMD_CHAIN=1 timeout -k 10 300 python3 -u bench.py --no-cpu-baseline > $o/bench_one_pivot_per_pass_m32768_n65536.json 2>> $o/err.log
LPAMD_CHAIN=16 timeout -k 10 300 python3 -u bench.py --steps 96 --warmup 32 --no-cpu-baseline > $o/bench_16_pivots_per_pass_m32768_n65536.json 2>> $o/err.log
LPAMD_CHAIN=32 timeout -k 10 300 python3 -u bench.py --steps 96 --warmup 32 --no-cpu-baseline > $o/bench_32_pivots_per_pass_m32768_n65536.json 2>> $o/err.log
# the other BASELINE sizes on one GPU
for strat in max_cost max_reduction; do
  timeout -k 10 300 python3 -u bench.py --config synthetic_m4096_n8192 --strategy $strat --steps 960 --warmup 96 --no-cpu-baseline > $o/bench_m4096_n8192_$strat.json 2>> $o/err.log
  timeout -k 10 400 python3 -u bench.py --config synthetic_m65536_n262144 --strategy $strat --steps 32 --warmup 16 --no-cpu-baseline > $o/bench_m65536_n262144_1gpu_$strat.json 2>> $o/err.log
done
# rank 0 of an N-GPU run alone on this GPU (RCCL world of one)
for w in 2 4 8; do
  LPAMD_BENCH_SHARD_OF=$w timeout -k 10 300 python3 -u bench.py --steps 512 --warmup 64 --no-cpu-baseline > $o/shard_of_${w}_max_cost.json 2>> $o/err.log
done
LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 python3 -u bench.py --strategy max_reduction --steps 256 --warmup 64 --no-cpu-baseline > $o/shard_of_8_max_reduction.json 2>> $o/err.log
LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 python3 -u bench.py --config synthetic_m65536_n262144 --strategy max_reduction --steps 64 --warmup 32 --no-cpu-baseline > $o/shard_of_8_m65536_n262144_max_reduction.json 2>> $o/err.log
LPAMD_BENCH_SHARD_OF=8 stats shard_of_8_max_cost --steps 512 --warmup 64
# the same with the RCCL all-gather in front of every chained step instead of the exchange inside the launches
LPAMD_SHARD_LINK=0 LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 python3 -u bench.py --steps 512 --warmup 64 --no-cpu-baseline > $o/shard_of_8_max_cost_collective.json 2>> $o/err.log
LPAMD_SHARD_LINK=0 LPAMD_BENCH_SHARD_OF=8 stats shard_of_8_max_cost_collective --steps 512 --warmup 64
LPAMD_SHARD_LINK=0 LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 python3 -u bench.py --strategy max_reduction --steps 256 --warmup 64 --no-cpu-baseline > $o/shard_of_8_max_reduction_collective.json 2>> $o/err.log
# two processes on this GPU trading tagged words through IPC-mapped uncached memory while both kernels run
[ -x bin_scratch/link_pingpong ] && { timeout -k 10 60 ./bin_scratch/link_pingpong 2000 4096; timeout -k 10 60 ./bin_scratch/link_pingpong 2000 64; } > $o/link_pingpong_two_processes.txt 2>> $o/err.log
# depth of the chain against the pass kernel (k_update_deep from 17 steps on; 16 and below: k_update_slab)
for d in 12 16 18 20 22 23 24 26 28 32; do
  LPAMD_CHAIN=$d timeout -k 10 200 python3 -u bench.py --steps $((d*6)) --warmup $d --no-cpu-baseline 2>> $o/err.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth $d', r['value'], 'pivots/s', r['roofline']['avg_launch_ms'], 'ms/pass', r['roofline']['achieved'], 'GB/s', r['roofline']['kernel'])
" >> $o/sweep_depth_m32768_n65536.txt
done
for d in 16 22 24 28 32; do
  LPAMD_CHAIN=$d timeout -k 10 200 python3 -u bench.py --strategy max_reduction --steps $((d*4)) --warmup $d --no-cpu-baseline 2>> $o/err.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('max_reduction depth $d', r['value'], 'pivots/s', r['roofline']['avg_launch_ms'], 'ms/pass', r['roofline']['achieved'], 'GB/s')
" >> $o/sweep_depth_m32768_n65536.txt
done
LPAMD_BENCH_SHARD_OF=8 stats shard_of_8_max_reduction --strategy max_reduction --steps 256 --warmup 64
stats bench_m4096_n8192_max_reduction --config synthetic_m4096_n8192 --strategy max_reduction --steps 960 --warmup 96
stats bench_m4096_n8192_max_cost --config synthetic_m4096_n8192 --steps 960 --warmup 96
# timelines of the set-up kernels from the build with wall-clock checkpoints (make dbg-clock), if it is there
if [ -f bin_scratch/liblpamd_dbgclock.so ]; then
  for cfg in synthetic_m32768_n65536 synthetic_m4096_n8192; do
    LPAMD_LIBRARY=bin_scratch/liblpamd_dbgclock.so timeout -k 10 200 python3 scripts/fused_timeline.py $cfg > $o/fused_setup_timeline_${cfg#synthetic_}.txt 2>> $o/err.log
    LPAMD_LIBRARY=bin_scratch/liblpamd_dbgclock.so timeout -k 10 200 python3 scripts/prb_timeline.py $cfg > $o/max_reduction_setup_timeline_${cfg#synthetic_}.txt 2>> $o/err.log
  done
  for l in 1 0; do
    LPAMD_SHARD_LINK=$l LPAMD_LIBRARY=bin_scratch/liblpamd_dbgclock.so timeout -k 10 200 python3 scripts/shard_timeline.py 8 2>> $o/err.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $o/shard_step_timeline_rank0_of_8_link$l.txt
  done
fi
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/prof_round/*.json")):
    try:
        text = open(f).read()
        d = json.loads(text) if text.lstrip().startswith("{\n") else json.loads([l for l in text.splitlines() if l.startswith("{")][-1])
    except Exception as ex:
        print(f, "unreadable", ex); continue
    if "metric" in d:
        r = d["roofline"]
        print("%-62s %10.1f pivots/s  %8.4f ms/pivot  pass %8.3f ms  %7.1f GB/s frac %.3f traffic %s" % (
            f.split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"], r["traffic"]))
    else:
        print("%-62s traffic/streamed %.4f" % (f.split("/")[-1], d["traffic_over_streamed"]))
PY
