#!/bin/bash
# chain tests with the rebuilt library, then everything profiles/round2 holds
set -e
export TMPDIR=/tmp
o=gpurun_out/r2p11
rm -rf $o; mkdir -p $o
timeout -k 10 600 python -u -m pytest tests/test_chain_gpu.py tests/test_reduction_chain_gpu.py tests/test_group_gpu.py -m gpu -x -q --timeout 400 --timeout-method thread > $o/tests.log 2>&1 || { tail -40 $o/tests.log; exit 1; }
tail -2 $o/tests.log
bash scripts/profile_round.sh
