#!/bin/bash
# round 3, GPU call Q: threaded host set-up of large BA graphs
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_badyn_gpu.py -q -x > $OUT/pytest.txt 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest.txt | tail -6
VIDO_BA_VERBOSE=1 python tools/prof_ba_global.py 2>&1 | tail -11
VIDO_BA_HOST_THREADS=1 VIDO_BA_VERBOSE=1 python tools/prof_ba_global.py 2>&1 | tail -11 | grep -E "setup|iters" | head -12
