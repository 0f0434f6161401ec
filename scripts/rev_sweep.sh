# Revised simplex on the tableau: parity tests, then early / late-phase cost per pivot.
set -e
timeout -k 10 400 python -u -m pytest tests/test_revised_gpu.py -x -q --timeout 150 --timeout-method thread > gpurun_out/rev_test.log 2>&1 || { tail -20 gpurun_out/rev_test.log; exit 1; }
tail -2 gpurun_out/rev_test.log
for cfg in "--m 32768 --n 65536 --pivots 40 --warmup 3" "--m 32768 --n 65536 --pivots 200 --warmup 300" "--m 32768 --n 65536 --pivots 200 --warmup 3000" "--m 4096 --n 8192 --pivots 200 --warmup 20"; do
  timeout -k 10 300 python scripts/revised_bench.py $cfg
done
