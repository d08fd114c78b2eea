#!/bin/bash
set -u
OUT=gpurun_out/r5s; mkdir -p $OUT; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/tl -o tl -- python $REPO/tools/prof_lfn_timeline.py > $REPO/$OUT/tl.log 2>&1
cd $REPO
python tools/summarize_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) "flow,flow,flow,flow,depth,depth,depth,depth" 60 > $OUT/nets_timeline_summary.txt 2>&1
rm -rf $OUT/tl
head -70 $OUT/nets_timeline_summary.txt | cut -c1-150
