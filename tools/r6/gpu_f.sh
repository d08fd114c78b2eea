#!/bin/bash
# k_ba_schur_mfma: landmarks per unit (VIDO_BA_SCHUR_CHUNK; default = sized for one workgroup per CU)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in 0 256 192 224 320 416 512; do
  if [ $c != 0 ]; then export VIDO_BA_SCHUR_CHUNK=$c; else unset VIDO_BA_SCHUR_CHUNK; fi
  rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o t -- python $R/tools/prof_ba_global.py > /tmp/kt.log 2>&1; 
  echo "== chunk ${c} $(grep -E 'iters' /tmp/kt.log | tail -1)"; f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); grep -E "schur" $f | cut -c1-150
done
unset VIDO_BA_SCHUR_CHUNK
cd $R && timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_badyn_gpu.py -m gpu -x -q 2>&1 | tail -12 | grep -v -E "RCCL|HIP version|ROCm version|Hostname|Librccl"
