#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4p; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -5
timeout 60 tools/ubench/wino_ablate_prof.bin | grep -v "wg \(0  \|mid\) wave"
WINO_ONLY=1 timeout 300 python tools/prof_wino.py 2>/dev/null | tee gpurun_out/r4p/prof_wino.txt
