#!/bin/bash
# the whole GPU suite
OUT=gpurun_out/r5m; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest_full.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_full.txt | tail -5 | tee $OUT/pytest_gpu.txt; grep -B30 "Error\|FAILED" $OUT/pytest_full.txt | head -60
