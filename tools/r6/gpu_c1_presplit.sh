#!/bin/bash
cd /tmp; export TMPDIR=/tmp; REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/c1ps; mkdir -p $OUT
for V in base presplit; do
 for SH in "1024 1024 50 68" "512 512 100 136" "256 256 200 272" "2048 2048 25 34"; do
  T=$(echo $SH | tr ' ' '_')
  LV=""; [ $V != base ] && LV=$REPO/vido-slam_amd/variants/libvido_$V.so
  VIDO_LIB_VARIANT=$LV timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${V}_$T -o p -- python $REPO/tools/r6/c1b3_run.py $SH 40 > $OUT/${V}_$T.log 2>&1
  python - <<P
import csv,glob
for f in glob.glob('$OUT/${V}_$T/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv1x1' in r['Name']: print('$V', '$SH', r['Name'][38:60], r['Calls'], 'avg %.1f us min %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
P
 done
done
