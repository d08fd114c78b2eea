#!/bin/bash
# async window solve beside the networks: stream priority / queue slot of the BA context
for mode in "VIDO_BA_CTX_PRIO=normal" "VIDO_BA_CTX_PRIO=least" "VIDO_BA_CTX_PRIO=skip1" "VIDO_BA_CTX_PRIO=skip2" "VIDO_BA_CTX_PRIO=skip3"; do echo "== $mode"
  env $mode VIDO_CALL_PROF=1 timeout 300 python tools/prof_nodet.py 80 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | grep "frames_per\|PartialBatch\|bawin_set\|bawin_push" | cut -c1-130; done
