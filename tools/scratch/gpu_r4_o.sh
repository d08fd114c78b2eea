#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4o; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -5
for a in 0 7 8; do timeout 60 tools/ubench/wino_ablate_$a.bin; done
timeout 300 python tools/scratch/wino_slope.py 2>/dev/null | tee gpurun_out/r4o/slope.txt
WINO_ONLY=1 timeout 300 python tools/prof_wino.py 2>/dev/null | tee gpurun_out/r4o/prof_wino.txt
