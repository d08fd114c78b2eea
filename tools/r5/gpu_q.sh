#!/bin/bash
# round 5, call q: full-size parity + Winograd parity after the per-form weight cache; headline A/B of the K-split form against the round-4 rule
set -u
OUT=gpurun_out/r5q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_wino_gpu.py -x -q 2>&1 | tail -4 | tee $OUT/pytest.txt
for rep in 1 2; do for m in 128 0; do echo "== rep $rep VIDO_WINO_MIN_WGS=$m" | tee -a $OUT/ab.txt; VIDO_WINO_MIN_WGS=$m timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2> $OUT/err_$m.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']; print(d['value'], d['ms_per_step'], {k: s[k] for k in s if 'ms' in k and ('flow' in k or 'depth' in k or 'mask' in k)}, d['config']['net_optimisations']['hip_graphs'])" | tee -a $OUT/ab.txt; done; done
