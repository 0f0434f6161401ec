#!/bin/bash
# The exchange inside the step's launch (LPAMD_SHARD_LINK): group tests, then the rank-0 rehearsal with and without it.
set -u
out=gpurun_out/link
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_group_gpu.py tests/test_sharded_gpu.py -x -q -m gpu --timeout 180 --timeout-method thread > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
: > $out/sweep.txt
for link in 0 1; do
  for w in 8 4 2; do
    LPAMD_SHARD_LINK=$link LPAMD_BENCH_SHARD_OF=$w timeout -k 10 200 python bench.py --no-cpu-baseline --steps 520 --warmup 78 > $out/b.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
    python - "$link" "$w" $out/b.json >> $out/sweep.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print("link", sys.argv[1], "shard of", sys.argv[2], d["value"], "pivots/s", d["ms_per_step"], "ms/pivot", d["roofline"]["avg_launch_ms"], "ms/pass", d["pivots_per_pass"], d["config"].get("exchange_in_launch"))
PY
  done
done
cat $out/sweep.txt
