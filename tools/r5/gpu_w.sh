#!/bin/bash
# round 5, call w: which layers the direct convolution should take beside the detector (headline A/B)
set -u
OUT=gpurun_out/r5w; mkdir -p $OUT
for rep in 1 2; do for m in off all novalu; do echo "== rep $rep direct=$m" | tee -a $OUT/ab.txt
  if [ $m = off ]; then export VIDO_NO_CONVDIRECT=1; else unset VIDO_NO_CONVDIRECT; export VIDO_CONVDIRECT_SET=$m; fi
  timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2> $OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']; print(d['value'], d['ms_per_step'], {k: s[k] for k in s if 'ms' in k and ('flow' in k or 'depth' in k or 'mask' in k)})" | tee -a $OUT/ab.txt; done; done
