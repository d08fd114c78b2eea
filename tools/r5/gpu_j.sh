#!/bin/bash
# round 5, call j: async window solve vs the number of hardware queues
set -u
OUT=gpurun_out/r5j; mkdir -p $OUT
for q in 4 8 16; do for mode in "VIDO_LBA_ASYNC_DUMMY=1" "VIDO_LBA_SYNC=1"; do echo "== GPU_MAX_HW_QUEUES=$q $mode" | tee -a $OUT/nodet.txt
  env GPU_MAX_HW_QUEUES=$q $mode timeout 300 python tools/prof_nodet.py 80 2>&1 | grep frames_per_s | cut -c1-420 | tee -a $OUT/nodet.txt; done; done
