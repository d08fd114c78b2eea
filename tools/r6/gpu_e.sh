#!/bin/bash
# global BA: the reduced solve's kernels (round 6; VIDO_BCR_SCALAR=1 = round 5's), then the BA tests
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-new}; do
  if [ $v = scalar ]; then export VIDO_BCR_SCALAR=1; else unset VIDO_BCR_SCALAR; fi
  rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o t -- python $R/tools/prof_ba_global.py > /tmp/kt.log 2>&1; grep -E "iters|Error|error" /tmp/kt.log | tail -3
  echo "== $v"; f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); head -18 $f | cut -c1-150; cp $f $R/gpurun_out/global_ba_kernel_stats_$v.csv
done
unset VIDO_BCR_SCALAR
cd $R && timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_badyn_gpu.py -m gpu -x -q 2>&1 | tail -12 | grep -v -E "RCCL|HIP version|ROCm version|Hostname|Librccl"
