#!/bin/bash
# m16d grouped convolution: parity tests + microbench with / without (VIDO_GCONV_NO_DMA16=1)
mkdir -p gpurun_out/r4ae
timeout 300 python -m pytest tests/test_maskrcnn_gpu.py -q -k "grouped_conv or bottleneck" 2>&1 | tail -4
timeout 200 python tools/prof_gconv.py 2>&1 | tee gpurun_out/r4ae/new.txt | tail -5
VIDO_GCONV_NO_DMA16=1 timeout 200 python tools/prof_gconv.py 2>&1 | tee gpurun_out/r4ae/old.txt | tail -5
