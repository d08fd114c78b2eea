#!/bin/bash
# round 5, call h: where the literal chain (no detector) spends its 5.8 ms: stage ms, C-ABI call profile, kernel stats
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r5h; mkdir -p $OUT
VIDO_CALL_PROF=1 timeout 300 python tools/prof_nodet.py 60 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee $OUT/nodet_call_profile.txt | cut -c1-400
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $REPO/tools/prof_nodet.py 60 > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" | cut -c1-70,140-400 > $OUT/nodet_kernel_stats.txt
find $OUT -name "*kernel_trace.csv" -delete
grep "frames_per_s" $OUT/kt.log | cut -c1-200
cut -c1-200 $OUT/nodet_kernel_stats.txt | head -45
