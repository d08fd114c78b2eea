#!/bin/bash
# round 5, call a: parity of the new blur / pyramid kernels, front-end timings old vs new, conv1x1 microbench after the range-check fix, the headline as the round starts
set -u
OUT=gpurun_out/r5a; mkdir -p $OUT
timeout 600 python -m pytest tests/test_orb_gpu.py tests/test_system_gpu.py tests/test_track_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest_orb.txt
for mode in "new" "VIDO_BLUR_TILES=1" "VIDO_PYR_TILES=0" "VIDO_BLUR_NB=2" "VIDO_BLUR_NB=5" "VIDO_PYR_NX=6 VIDO_PYR_NY=10" "VIDO_PYR_NX=4 VIDO_PYR_NY=6"; do
  echo "== $mode" | tee -a $OUT/frontend.txt
  if [ "$mode" = new ]; then timeout 120 python tools/prof_frontend_batch.py 2>&1 | grep "pyramid_ms" | tee -a $OUT/frontend.txt
  else env $mode timeout 120 python tools/prof_frontend_batch.py 2>&1 | grep "pyramid_ms" | tee -a $OUT/frontend.txt; fi
done
timeout 300 python tools/prof_conv1x1.py 2>&1 | grep " -> " | cut -c1-250 | tee $OUT/conv1x1.txt
timeout 900 python bench.py --steps 60 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench_60.json 2> $OUT/bench_60.err; echo "bench rc $?"; cut -c1-400 $OUT/bench_60.json
