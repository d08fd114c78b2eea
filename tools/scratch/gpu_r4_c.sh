#!/bin/bash
# round 4, GPU call C: where the tracker's "ORB + lists" stage goes inside the pipeline (ms_orb / ms_lists), network-stream schedules under rocprof (queue ids), global-BA kernel trace, N = 2 on one GPU
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r4c; mkdir -p $OUT
timeout 900 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4c/bench.json")); print(d["value"], d["ms_per_step"], d["stage_ms"])
PY
timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -x -k "two_ranks" > $OUT/pytest_n2.txt 2>&1; tail -3 $OUT/pytest_n2.txt
cd /tmp && export TMPDIR=/tmp
for sch in "flow+depth" "flow,depth" "det" "none" "flow+depth!"; do
  tag=$(echo "$sch" | tr '+,!' 'pcx')
  rm -rf /tmp/sch_$tag
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/sch_$tag -o t -- python $REPO/bench.py --steps 20 --warmup 3 --no-extra --cpu-baseline 0 --net-streams "$sch" > $OUT/sched_$tag.json 2> $OUT/sched_$tag.err
  f=$(find /tmp/sch_$tag -name "*kernel_trace.csv" | head -1)
  python $REPO/tools/sched_timeline.py "$f" "$sch" > $OUT/sched_${tag}_summary.json 2>> $OUT/sched_$tag.err; cat $OUT/sched_${tag}_summary.json; echo
  python -c "
import json; d=json.load(open('$OUT/sched_$tag.json')); print('$sch', d['value'], d['ms_per_step'])"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_gba -o gba -- python $REPO/tools/prof_ba_global.py > $OUT/prof_gba.log 2>&1
rm -f $OUT/prof_gba/*kernel_trace.csv
cd $REPO
