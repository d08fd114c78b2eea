#!/bin/bash
# round 3, GPU call A: the full -m gpu suite (new configs[3](b)/configs[4]/k1/shard tests included), a baseline bench line, and the per-phase kernel timeline of the network nodes
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3a; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest.txt
timeout 600 python bench.py --steps 40 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python $REPO/tools/prof_nets_timeline.py > $OUT/tl.log 2>&1
cd $REPO
N="flow,flow,flow,depth,depth,depth,trunk,trunk,trunk,rpn,rpn,rpn,box,box,box,mask,mask,mask,total,total,total"
python tools/summarize_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) "$N,$N" 14 > $OUT/tl_summary.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
tail -5 $OUT/pytest.txt; tail -3 $OUT/tl.log
