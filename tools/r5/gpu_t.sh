#!/bin/bash
# round 5, call t: barrier-free K-split Winograd kernel: parity, microbenchmark, LiteFlowNet / detector parity at full size, headline
set -u
OUT=gpurun_out/r5t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wino_gpu.py tests/test_convdirect_gpu.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_wino.txt
timeout 600 python tools/prof_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $OUT/wino_microbench.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_nets_modules_gpu.py -x -q 2>&1 | tail -4 | tee $OUT/pytest_nets.txt
for rep in 1 2; do timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2> $OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']; print(d['value'], d['ms_per_step'], {k: s[k] for k in s if 'ms' in k and ('flow' in k or 'depth' in k or 'mask' in k)}, d['config']['net_optimisations']['hip_graphs'])" | tee -a $OUT/ab.txt; done
