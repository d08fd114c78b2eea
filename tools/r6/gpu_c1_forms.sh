#!/bin/bash
# kernel-trace durations of the conv1x1 forms on the detector's four bottleneck shapes: tools/r6/gpu_c1_forms.sh [arith list] (f16x2 bf16x3 f32)
cd /tmp; export TMPDIR=/tmp; REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/c1forms; mkdir -p $OUT
for AR in ${*:-f16x2 bf16x3}; do
 for SH in "256 256 200 272" "512 512 100 136" "1024 1024 50 68" "2048 2048 25 34" "256 1024 50 68" "1024 256 50 68"; do
  T=$(echo $SH | tr ' ' '_')
  VIDO_CONV1X1_ARITH=$AR timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${AR}_$T -o p -- python $REPO/tools/r6/c1b3_run.py $SH 40 > $OUT/${AR}_$T.log 2>&1
  python - <<P
import csv,glob
for f in glob.glob('$OUT/${AR}_$T/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv1x1' in r['Name']: print('$AR', '$SH', r['Name'][:60], r['Calls'], 'avg %.1f us min %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
P
 done
done
