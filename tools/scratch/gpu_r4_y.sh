#!/bin/bash
cd /tmp; export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r4y; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tld -o tl -- python $REPO/tools/prof_det_timeline.py > $OUT/tld.log 2>&1
cd $REPO
python tools/summarize_timeline.py $(find $OUT/tld -name "*kernel_trace.csv" | head -1) "det,det,det,det,det,det" 60 > $OUT/det_timeline_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
grep -n "phase 2 det" -A60 $OUT/det_timeline_summary.txt | cut -c1-140
