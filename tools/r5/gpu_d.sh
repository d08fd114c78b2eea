#!/bin/bash
# round 5, call d: ORB parity after the compass identity, kernel trace of the batched front end
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r5d; mkdir -p $OUT
timeout 600 python -m pytest tests/test_orb_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest_orb.txt
timeout 120 python tools/prof_frontend_batch.py 2>&1 | grep "pyramid_ms" | tee $OUT/frontend.txt
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fe -o fe -- python $REPO/tools/prof_frontend_batch.py > $OUT/fe.log 2>&1
f=$(find $OUT/fe -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200 | tee $OUT/frontend_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
