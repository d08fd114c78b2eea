#!/bin/bash
# round 5, call c: the 128 x 112 conv1x1 form: parity (library's choice and both forced forms), microbench, detector time
set -u
OUT=gpurun_out/r5c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_maskrcnn_gpu.py -m gpu -q -x -k "conv1x1 or bottleneck" 2>&1 | tail -5 | tee $OUT/pytest_c1.txt
for tn in 0 128; do echo "== VIDO_CONV1X1_TN=$tn" | tee -a $OUT/conv1x1.txt; VIDO_CONV1X1_TN=$tn timeout 300 python tools/prof_conv1x1.py 2>&1 | grep " -> " | cut -c1-270 | tee -a $OUT/conv1x1.txt; done
for tn in 0 128; do echo "== VIDO_CONV1X1_TN=$tn" | tee -a $OUT/bench.txt; VIDO_CONV1X1_TN=$tn timeout 600 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 2> $OUT/bench_$tn.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: d['stage_ms'][k] for k in ('liteflownet_ms', 'monodepth2_ms', 'maskrcnn_x101_fpn_ms', 'track_total_ms')})" | tee -a $OUT/bench.txt; done
