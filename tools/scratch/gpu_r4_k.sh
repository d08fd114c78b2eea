#!/bin/bash
# round 4, call K: first contact of csrc/wino.hip: parity tests, then the microbenchmark
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4k; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r4k/pytest_wino.txt
cat gpurun_out/r4k/pytest_wino.txt
timeout 300 python tools/prof_wino.py > gpurun_out/r4k/prof_wino.txt 2>gpurun_out/r4k/prof_wino.err
cat gpurun_out/r4k/prof_wino.txt; tail -5 gpurun_out/r4k/prof_wino.err
