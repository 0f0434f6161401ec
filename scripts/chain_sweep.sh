#!/bin/bash
# pivots/s of the default bench for every chain depth (pivots per pass over the tableau)
set -e
out=gpurun_out/chain_sweep.jsonl
: > $out
for cfg in synthetic_m32768_n65536 synthetic_m4096_n8192; do
  for d in 1 2 4 6 8; do
    steps=96; [ $cfg = synthetic_m4096_n8192 ] && steps=960
    LPAMD_CHAIN=$d timeout -k 10 120 python -u bench.py --config $cfg --steps $steps --warmup 8 --no-cpu-baseline | sed "s/^{/{\"chain\": $d, /" >> $out
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/chain_sweep.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"], "chain", d["chain"], "pivots/s", d["value"], "pass ms", d["roofline"]["avg_launch_ms"], "cols", d["roofline"]["active_columns"])
PY
