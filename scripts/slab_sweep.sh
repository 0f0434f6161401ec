#!/bin/bash
# pivots/s of the default bench for the variants of the chained pass (col' from L2 / in registers)
set -e
out=gpurun_out/slab_sweep.jsonl
: > $out
cfg=${1:-synthetic_m32768_n65536}
steps=${2:-96}
for d in ${3:-4 8}; do
  for sl in 0 1 2 3 4; do
    LPAMD_CHAIN=$d LPAMD_SLAB=$sl timeout -k 10 120 python -u bench.py --config $cfg --steps $steps --warmup 8 --no-cpu-baseline | sed "s/^{/{\"chain\": $d, \"slab\": $sl, /" >> $out
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/slab_sweep.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"], "chain", d["chain"], "slab", d["slab"], "pivots/s", d["value"], "pass ms", d["roofline"]["avg_launch_ms"])
PY
