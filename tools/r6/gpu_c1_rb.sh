#!/bin/bash
cd /tmp; export TMPDIR=/tmp; REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/c1rb; mkdir -p $OUT
for RB in 5 4 3; do
 for SH in "64 256 200 272" "256 256 200 272" "256 64 200 272" "512 512 100 136" "256 512 100 136" "256 1024 50 68"; do
  T=$(echo $SH | tr ' ' '_')
  VIDO_CONV1X1_B3_FORM=3 VIDO_CONV1X1_H2_RB=$RB timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${RB}_$T -o p -- python $REPO/tools/r6/c1b3_run.py $SH 40 > $OUT/${RB}_$T.log 2>&1
  python - <<P
import csv,glob
for f in glob.glob('$OUT/${RB}_$T/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv1x1' in r['Name']: print('RB $RB', '$SH', r['Name'][38:60], r['Calls'], 'avg %.1f us min %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
P
 done
done
