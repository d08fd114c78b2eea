#!/bin/bash
# round 3, GPU call N: Schur fallback for revisited landmarks, error codes through the C handle
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_badyn_gpu.py tests/test_system_gpu.py tests/test_facade_gpu.py -q -x > $OUT/pytest.txt 2>&1; tail -8 $OUT/pytest.txt
