#!/bin/bash
# k_ba_linearize: whole-record stores / non-temporal stores
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" linfull linnt linboth; do
  if [ -n "$v" ]; then export VIDO_LIB_VARIANT=$R/vido-slam_amd/variants/libvido_$v.so; else unset VIDO_LIB_VARIANT; fi
  rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o t -- python $R/tools/prof_ba_global.py > /tmp/kt.log 2>&1
  echo "== ${v:-base} $(grep -E 'iters' /tmp/kt.log | tail -1)"; f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); grep -E "linearize|k_ba_schur_mfma" $f | cut -c1-150
done
