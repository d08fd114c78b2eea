#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4l; export TMPDIR=/tmp
timeout 300 python tools/scratch/wino_slope.py > gpurun_out/r4l/slope.txt 2>gpurun_out/r4l/slope.err; cat gpurun_out/r4l/slope.txt; tail -3 gpurun_out/r4l/slope.err
timeout 300 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -3
WINO_ONLY=1 timeout 300 python tools/prof_wino.py 2>/dev/null | tee gpurun_out/r4l/prof_wino.txt
