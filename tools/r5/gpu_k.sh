#!/bin/bash
# the tracker alone: window solve async vs sync
for mode in "VIDO_LBA_ASYNC_DUMMY=1" "VIDO_LBA_SYNC=1"; do echo "== $mode"
  env $mode VIDO_CALL_PROF=1 timeout 300 python tools/prof_tracker.py 80 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | grep "frames\|PartialBatch\|bawin\|orb_extract\|PoseOptimizationFlow2Cam" | cut -c1-330; done
