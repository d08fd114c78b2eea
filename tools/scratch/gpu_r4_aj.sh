#!/bin/bash
# the eight-wave Winograd kernel (VIDO_WINO_W8=1): parity tests + microbench against the four-wave one
VIDO_WINO_W8=1 timeout 300 python -m pytest tests/test_wino_gpu.py -q -x 2>&1 | tail -6
echo "--- w8"; VIDO_WINO_W8=1 WINO_ONLY=1 timeout 200 python tools/prof_wino.py 2>&1 | grep " x " | cut -c1-120
echo "--- w4"; WINO_ONLY=1 timeout 200 python tools/prof_wino.py 2>&1 | grep " x " | cut -c1-120
