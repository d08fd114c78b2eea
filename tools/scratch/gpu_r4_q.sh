#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4q; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_maskrcnn_gpu.py -x -q -k "conv1x1 or bottleneck" 2>&1 | tail -8
timeout 300 python tools/prof_conv1x1.py 2>/dev/null | tee gpurun_out/r4q/conv1x1.txt
