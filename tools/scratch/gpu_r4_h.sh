#!/bin/bash
# round 4, GPU call H: conv1x1 chunk-size variants
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_maskrcnn_gpu.py -q -x -k "conv1x1 or matrix_core_1x1" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
for n in 1 2; do echo "NSUB $n"; VIDO_CONV1X1_NSUB=$n timeout 300 python tools/prof_conv1x1.py 2>/dev/null | cut -c1-60,150-; done
