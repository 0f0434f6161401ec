# Pass kernel of long chains: automatic grid (about 48 batches per wavefront) against fixed grids, on the shard sizes of the benchmark.
run() { python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys;r=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$TAG', r['value'], 'pivots/s', r['ms_per_step'], 'ms/pivot, pass', r['roofline']['avg_launch_ms'], 'ms', r['roofline']['frac'])"; }
for g in 0 8192; do
  export LPAMD_DEEP_GRID=$g
  TAG="full grid $g" run --steps 184 --warmup 23
  TAG="shard-of-8 grid $g" LPAMD_BENCH_SHARD_OF=8 run --steps 512 --warmup 64
  TAG="shard-of-4 grid $g" LPAMD_BENCH_SHARD_OF=4 run --steps 512 --warmup 64
  TAG="shard-of-2 grid $g" LPAMD_BENCH_SHARD_OF=2 run --steps 256 --warmup 64
  TAG="m65536 n262144 grid $g" run --config synthetic_m65536_n262144 --steps 46 --warmup 23
  TAG="m65536 n262144 shard-of-8 grid $g" LPAMD_BENCH_SHARD_OF=8 run --config synthetic_m65536_n262144 --steps 92 --warmup 23
done
