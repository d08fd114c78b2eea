#!/bin/bash
# round 3, GPU call L: kernel timeline of the LiteFlowNet / MonoDepth2 graphs
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3l; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python $REPO/tools/prof_lfn_timeline.py > $OUT/tl.log 2>&1; cd $REPO; tail -2 $OUT/tl.log
python tools/summarize_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) flow,flow,flow,flow,depth,depth,depth,depth 40 > $OUT/tl_summary.txt; awk '/^phase 3/,/^phase 4/' $OUT/tl_summary.txt | cut -c1-160; awk '/^phase 7/,0' $OUT/tl_summary.txt | cut -c1-160
