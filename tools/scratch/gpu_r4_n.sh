#!/bin/bash
# round 4, call N: kernel timelines of the flow / depth graphs and of the one-graph detector with wino.hip in place
cd /tmp; export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r4n; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python $REPO/tools/prof_lfn_timeline.py > $OUT/tl.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tld -o tl -- python $REPO/tools/prof_det_timeline.py > $OUT/tld.log 2>&1
cd $REPO
python tools/summarize_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) "flow,flow,flow,flow,depth,depth,depth,depth" 40 > $OUT/nets_timeline_summary.txt 2>&1
python tools/summarize_timeline.py $(find $OUT/tld -name "*kernel_trace.csv" | head -1) "det,det,det,det,det,det" 45 > $OUT/det_timeline_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +2M -delete
sed -n 42,82p $OUT/nets_timeline_summary.txt | cut -c1-150
grep -n "phase 1 det" -A46 $OUT/det_timeline_summary.txt | cut -c1-150
