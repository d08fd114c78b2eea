#!/bin/bash
# round 5, call k2: the K-split Winograd form with TWO channel slices x two tile blocks (half the U traffic per matrix instruction, twice the chain): parity, microbench, headline A/B
set -u
OUT=gpurun_out/r5k2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -3 | tee $OUT/pytest_wino.txt
WINO_ONLY=1 timeout 600 python tools/prof_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $OUT/wino_microbench.txt
for rep in 1 2; do for m in 4 2; do echo "== rep $rep VIDO_WINO_KSPLIT=$m" | tee -a $OUT/ab.txt
  VIDO_WINO_KSPLIT=$m timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2> $OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']; print(d['value'], d['ms_per_step'], {k: s[k] for k in s if 'ms' in k and ('flow' in k or 'depth' in k or 'mask' in k)})" | tee -a $OUT/ab.txt; done; done
