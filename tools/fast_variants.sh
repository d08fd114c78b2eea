#!/bin/bash
# times k_fast_strips for every library under vido-slam_amd/variants/ (tools/prof_frontend_batch.py) -> gpurun_out/<dir>/fast_variants.txt
out=${1:-gpurun_out/r2i}; mkdir -p $out; : > $out/fast_variants.txt
for lib in vido-slam_amd/libvido_slam_hip.so vido-slam_amd/variants/*.so; do
  echo "== $lib" >> $out/fast_variants.txt
  VIDO_LIB_PATH=/root/repo/$lib timeout 120 python tools/prof_frontend_batch.py 2>&1 | grep -v amdgpu.ids | tail -4 >> $out/fast_variants.txt
done
grep -v "variants/\*" $out/fast_variants.txt | grep -o "==.*\|'fast_ms': [0-9.]*\|fill.*\|est VALU.*\|{'a_iters.*" $out/fast_variants.txt
