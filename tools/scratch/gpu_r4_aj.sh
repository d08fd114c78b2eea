#!/bin/bash
# the eight-wave Winograd kernel (VIDO_WINO_W8=1: waves w, w + 4 pair up; 2: waves 2 k, 2 k + 1): parity tests + microbench against the four-wave one
VIDO_WINO_W8=2 timeout 300 python -m pytest tests/test_wino_gpu.py -q -x 2>&1 | tail -3
for m in 2 1 0; do echo "--- w8 mode $m"; VIDO_WINO_W8=$m WINO_ONLY=1 timeout 200 python tools/prof_wino.py 2>&1 | grep " x " | cut -c1-120 | grep "P2\|regularisation 2\|mask head"; done
