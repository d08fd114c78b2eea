#!/bin/bash
# round 5, call x: LiteFlowNet's dense 3x3 layers on the library's vector-ALU Winograd (VIDO_LFN_NO_WINO=1) beside the detector's matrix-pipe convolutions: headline A/B
set -u
OUT=gpurun_out/r5x; mkdir -p $OUT
export VIDO_CONVDIRECT_SET=novalu
for rep in 1 2; do for m in "" 1; do echo "== rep $rep VIDO_LFN_NO_WINO=$m" | tee -a $OUT/ab.txt
  VIDO_LFN_NO_WINO=$m timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2> $OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']; print(d['value'], d['ms_per_step'], {k: s[k] for k in s if 'ms' in k and ('flow' in k or 'depth' in k or 'mask' in k)})" | tee -a $OUT/ab.txt; done; done
