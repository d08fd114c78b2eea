#!/bin/bash
for v in "" rot; do
  echo "=== variant: ${v:-base}"
  if [ -n "$v" ]; then export VIDO_LIB_VARIANT=$PWD/vido-slam_amd/variants/libvido_$v.so; else unset VIDO_LIB_VARIANT; fi
  timeout 300 python tools/r6/conv1x1_b3_check.py 2>&1 | grep -v amdgpu.ids | head -5 | sed -e 's/fp32 instr .layout 0.: //'
done
