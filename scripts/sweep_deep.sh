# Pass kernel of long chains: depth x columns per batch (LPAMD_DEEP) x grid on the default benchmark shard.
set -e
for cfg in "24 8 4096" "24 4 4096" "24 4 8192" "32 8 4096" "32 4 4096" "32 4 8192" "28 4 4096" "20 4 4096" "16 4 4096"; do
  set -- $cfg
  LPAMD_CHAIN=$1 LPAMD_DEEP=$2 LPAMD_DEEP_GRID=$3 timeout -k 10 120 python bench.py --no-cpu-baseline --steps $(($1*4)) --warmup $1 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth $1 batch $2 grid $3:', r['value'], 'pivots/s', r['roofline']['avg_launch_ms'], 'ms/pass', r['roofline']['achieved'], 'GB/s')
"
done
