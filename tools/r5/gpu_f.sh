#!/bin/bash
# round 5, call f: the full-size parity tests + the two bench shapes of the Winograd list
set -u
OUT=gpurun_out/r5f; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -s 2>&1 | grep -v "amdgpu.ids\|Warning\|warnings.warn" | tail -25 | tee $OUT/pytest_fullsize.txt
timeout 900 python -m pytest tests/test_wino_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest_wino.txt
