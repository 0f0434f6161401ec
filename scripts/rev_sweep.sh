set -e
timeout -k 10 400 python -u -m pytest tests/test_revised_gpu.py -x -q --timeout 150 --timeout-method thread > gpurun_out/rev_test.log 2>&1 || { tail -20 gpurun_out/rev_test.log; exit 1; }
tail -2 gpurun_out/rev_test.log
for R in 64 16; do
  echo "ROWS=$R"
  LPAMD_REV_ROWDOT_ROWS=$R timeout -k 10 200 python scripts/revised_bench.py --m 32768 --n 65536 --pivots 40 --warmup 3
  LPAMD_REV_ROWDOT_ROWS=$R timeout -k 10 300 python scripts/revised_bench.py --m 32768 --n 65536 --pivots 200 --warmup 3000
done
