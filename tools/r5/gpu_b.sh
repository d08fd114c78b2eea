#!/bin/bash
# round 5, call b: pyramid tile-count sweep
set -u
OUT=gpurun_out/r5b; mkdir -p $OUT
for mode in "VIDO_PYR_NX=6 VIDO_PYR_NY=10" "VIDO_PYR_NX=8 VIDO_PYR_NY=12" "VIDO_PYR_NX=10 VIDO_PYR_NY=15" "VIDO_PYR_NX=8 VIDO_PYR_NY=8" "VIDO_PYR_NX=5 VIDO_PYR_NY=12" "VIDO_PYR_NX=12 VIDO_PYR_NY=20"; do
  echo "== $mode" | tee -a $OUT/frontend.txt
  env $mode timeout 120 python tools/prof_frontend_batch.py 2>&1 | grep "pyramid_ms" | cut -c1-60 | tee -a $OUT/frontend.txt
done
