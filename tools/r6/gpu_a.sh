#!/bin/bash
for f in 3 2 1 0; do
  echo "=== VIDO_CONV1X1_B3_FORM=$f"
  VIDO_CONV1X1_B3_FORM=$f timeout 300 python tools/r6/conv1x1_b3_check.py 2>&1 | grep -v amdgpu.ids | sed -e 's/fp32 instr .layout 0.: //'
done
