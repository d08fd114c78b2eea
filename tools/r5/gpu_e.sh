#!/bin/bash
# round 5, call e: front end after the fused ingest, the two-launch scan and the short sincos: parity + timings + kernel trace
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r5e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_system_gpu.py tests/test_track_gpu.py tests/test_facade_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest.txt
for mode in "new" "VIDO_PYR_TILES=2" "VIDO_ORB_SCAN1=1"; do
  echo "== $mode" | tee -a $OUT/frontend.txt
  if [ "$mode" = new ]; then timeout 120 python tools/prof_frontend_batch.py 2>&1 | grep "pyramid_ms" | tee -a $OUT/frontend.txt
  else env $mode timeout 120 python tools/prof_frontend_batch.py 2>&1 | grep "pyramid_ms" | tee -a $OUT/frontend.txt; fi
done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fe -o fe -- python $REPO/tools/prof_frontend_batch.py > $OUT/fe.log 2>&1
f=$(find $OUT/fe -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-60,150-330 | tee $OUT/frontend_kernel_stats.txt
find $OUT -name "*kernel_trace.csv" -delete
