#!/bin/bash
# round 5, call r: the direct implicit-GEMM convolution: parity, microbenchmark, full-size LiteFlowNet parity, headline A/B
set -u
OUT=gpurun_out/r5r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_convdirect_gpu.py -x -q 2>&1 | tail -8 | tee $OUT/pytest_convdirect.txt
timeout 600 python tools/prof_convdirect.py 2>&1 | grep -v amdgpu.ids | tee $OUT/convdirect_microbench.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_nets_modules_gpu.py -x -q 2>&1 | tail -4 | tee $OUT/pytest_nets.txt
for m in 1 ""; do echo "== VIDO_NO_CONVDIRECT=$m" | tee -a $OUT/ab.txt; VIDO_NO_CONVDIRECT=$m timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2> $OUT/err_x$m.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']; print(d['value'], d['ms_per_step'], {k: s[k] for k in s if 'ms' in k and ('flow' in k or 'depth' in k or 'mask' in k)}, d['config']['net_optimisations']['hip_graphs'])" | tee -a $OUT/ab.txt; done
