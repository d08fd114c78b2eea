#!/bin/bash
# k_ba_schur_mfma: landmarks per window flush (BA_CHUNK) 64 / 128 / 256 / 512
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" chunk64 chunk128 chunk512; do
  if [ -n "$v" ]; then export VIDO_LIB_VARIANT=$R/vido-slam_amd/variants/libvido_$v.so; else unset VIDO_LIB_VARIANT; fi
  rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o t -- python $R/tools/prof_ba_global.py > /tmp/kt.log 2>&1; grep -E "iters|Error|error" /tmp/kt.log | tail -2
  echo "== ${v:-chunk256 (base)}"; f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); grep -E "schur|linearize" $f | cut -c1-150
done
