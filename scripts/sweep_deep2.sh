set -e
for cfg in "23 8 8192" "23 8 16384" "24 8 16384" "24 8 32768" "32 8 16384" "16 8 8192"; do
  set -- $cfg
  LPAMD_CHAIN=$1 LPAMD_DEEP=$2 LPAMD_DEEP_GRID=$3 timeout -k 10 120 python bench.py --no-cpu-baseline --steps $(($1*6)) --warmup $1 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth $1 batch $2 grid $3:', r['value'], 'pivots/s', r['roofline']['avg_launch_ms'], 'ms/pass', r['roofline']['achieved'], 'GB/s')
"
done
