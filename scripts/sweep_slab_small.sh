# Shards inside the Infinity Cache (m=4096 n=8192): register shape / grid of the pass kernel, and the LDS variant.
run() { timeout -k 10 200 python3 bench.py --config synthetic_m4096_n8192 --steps 960 --warmup 96 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;r=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$TAG', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])"; }
TAG="shape 1 (512-row slabs)" LPAMD_SLAB=1 run
TAG="shape 1 grid 512" LPAMD_SLAB=1 LPAMD_SLAB_GRID=512 run
TAG="shape 1 grid 1024" LPAMD_SLAB=1 LPAMD_SLAB_GRID=1024 run
TAG="shape 2 (512-row slabs, 4 columns)" LPAMD_SLAB=2 run
TAG="shape 4 (1024-row slabs)" LPAMD_SLAB=4 run
TAG="LDS variant, 16 steps" LPAMD_DEEP=8 run
TAG="LDS variant, 16 steps, 4 columns" LPAMD_DEEP=4 run
TAG="LDS variant, 23 steps" LPAMD_CHAIN=23 LPAMD_DEEP=8 run
