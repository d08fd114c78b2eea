#!/bin/bash
# round 3, GPU call O: the whole GPU suite twice (stability), then smoke()
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r3o; mkdir -p $OUT
for i in 1 2; do timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_$i.txt 2>&1; grep -E "passed|failed" $OUT/pytest_$i.txt | tail -2; done
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
