#!/bin/bash
# round 4, GPU call J: band pyramid in the batched front end; host threads of the global-BA set-up
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4j; mkdir -p $OUT
for m in 0 1; do echo "VIDO_ORB_BANDS=$m"; VIDO_ORB_BANDS=$m timeout 200 python tools/prof_frontend_batch.py 2>/dev/null | head -1; done
timeout 300 python -m pytest tests/test_orb_gpu.py -q -x 2>&1 | tail -2
VIDO_ORB_BANDS=1 timeout 300 python -m pytest tests/test_orb_gpu.py -q -x 2>&1 | tail -2
for t in 8 16 32 64; do echo "VIDO_BA_HOST_THREADS=$t"; VIDO_BA_HOST_THREADS=$t VIDO_BA_VERBOSE=1 timeout 200 python tools/prof_ba_global.py 2>&1 | grep "ba setup\|iters" | tail -9; done
