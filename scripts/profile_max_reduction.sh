#!/bin/bash
# The MAX_REDUCTION legs of scripts/profile_round.sh alone (about 4 minutes on the GPU box):
#   bash scripts/profile_max_reduction.sh && cp gpurun_out/prof_mr/* profiles/roundN/
set -e
export TMPDIR=/tmp
o=gpurun_out/prof_mr
rm -rf $o; mkdir -p $o
stats() {
  local name=$1; shift
  echo "stats $name"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/st_$name -- python3 bench.py --no-cpu-baseline "$@" > $o/${name}_under_rocprof.json 2>> $o/err.log
  find $o/st_$name -name "*kernel_stats.csv" -exec cp {} $o/${name}_kernel_stats.csv \;
  rm -rf $o/st_$name
}
pmc() {
  local name=$1; shift
  echo "pmc $name"
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 44 --warmup 22 --no-cpu-baseline "$@" > $o/pmc_fetch.json 2>> $o/err.log
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 44 --warmup 22 --no-cpu-baseline "$@" > $o/pmc_write.json 2>> $o/err.log
  python3 scripts/pmc_traffic.py $o/pmc_fetch $o/pmc_write $o/pmc_fetch.json > $o/pmc_hbm_traffic_$name.json
  rm -rf $o/pmc_fetch $o/pmc_write $o/pmc_fetch.json $o/pmc_write.json
}
timeout -k 10 300 python3 -u bench.py --strategy max_reduction --no-cpu-baseline > $o/bench_max_reduction_m32768_n65536.json 2>> $o/err.log
stats bench_max_reduction_m32768_n65536 --strategy max_reduction
pmc max_reduction_m32768_n65536 --strategy max_reduction
timeout -k 10 300 python3 -u bench.py --config synthetic_m4096_n8192 --strategy max_reduction --steps 960 --warmup 96 --no-cpu-baseline > $o/bench_m4096_n8192_max_reduction.json 2>> $o/err.log
timeout -k 10 400 python3 -u bench.py --config synthetic_m65536_n262144 --strategy max_reduction --steps 32 --warmup 16 --no-cpu-baseline > $o/bench_m65536_n262144_1gpu_max_reduction.json 2>> $o/err.log
LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 python3 -u bench.py --strategy max_reduction --steps 256 --warmup 64 --no-cpu-baseline > $o/shard_of_8_max_reduction.json 2>> $o/err.log
LPAMD_BENCH_SHARD_OF=8 timeout -k 10 300 python3 -u bench.py --config synthetic_m65536_n262144 --strategy max_reduction --steps 64 --warmup 32 --no-cpu-baseline > $o/shard_of_8_m65536_n262144_max_reduction.json 2>> $o/err.log
LPAMD_BENCH_SHARD_OF=8 stats shard_of_8_max_reduction --strategy max_reduction --steps 256 --warmup 64
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/prof_mr/*.json")):
    d = json.load(open(f))
    if "metric" in d:
        r = d["roofline"]
        print("%-62s %10.1f pivots/s  %8.4f ms/pivot  pass %8.3f ms  %7.1f GB/s frac %.3f" % (
            f.split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"]))
    else:
        print("%-62s traffic/streamed %.4f" % (f.split("/")[-1], d["traffic_over_streamed"]))
PY
