#!/bin/bash
# the whole GPU suite + smoke()
set -u
OUT=gpurun_out/r5m; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_full.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_full.txt | tail -3 | tee $OUT/pytest_gpu.txt; grep -E "^FAILED|^ERROR" $OUT/pytest_full.txt | head
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
