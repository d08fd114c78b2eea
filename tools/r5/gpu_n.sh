#!/bin/bash
# round 5, call n: the Winograd kernel on the layers it does NOT fill the chip with (VIDO_WINO_MIN_WGS): stand-alone it is no faster than the library there, but beside the
# other two streams what counts is CU time, not latency.  Headline A/B.
set -u
OUT=gpurun_out/r5n; mkdir -p $OUT
for rep in 1 2; do for m in 0 48 1; do echo "== rep $rep VIDO_WINO_MIN_WGS=$m" | tee -a $OUT/ab.txt; VIDO_WINO_MIN_WGS=$m timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2> $OUT/err_$m.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']; print(d['value'], d['ms_per_step'], {k: s[k] for k in s if 'ms' in k and ('flow' in k or 'depth' in k or 'mask' in k)})" | tee -a $OUT/ab.txt; done; done
