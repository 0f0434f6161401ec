#!/bin/bash
set -e
export TMPDIR=/tmp
o=gpurun_out/suite
rm -rf $o; mkdir -p $o
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 500 --timeout-method thread --durations=8 > $o/tests.log 2>&1 || { tail -60 $o/tests.log; exit 1; }
tail -12 $o/tests.log
