#!/bin/bash
# round 5, call o: the K-split form of the Winograd kernel: parity on every shape, microbenchmark against the tile form and the library, headline A/B
set -u
OUT=gpurun_out/r5o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_wino.txt
timeout 600 python tools/prof_wino.py 2>&1 | tee $OUT/wino_microbench.txt
for m in 128 0; do echo "== VIDO_WINO_MIN_WGS=$m" | tee -a $OUT/ab.txt; VIDO_WINO_MIN_WGS=$m timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2> $OUT/err_$m.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']; print(d['value'], d['ms_per_step'], {k: s[k] for k in s if 'ms' in k and ('flow' in k or 'depth' in k or 'mask' in k)})" | tee -a $OUT/ab.txt; done
