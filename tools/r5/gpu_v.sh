#!/bin/bash
# round 5, call v: global BA host set-up with spinning pool workers: thread counts, phases, tests
set -u
OUT=gpurun_out/r5v; mkdir -p $OUT; rm -f $OUT/*
python tools/prof_ba_setup.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/setup.txt
for t in 4 16; do VIDO_BA_HOST_THREADS=$t python tools/prof_ba_setup.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/setup.txt; done
VIDO_BA_VERBOSE=1 python tools/prof_ba_global.py 2>&1 | grep -v amdgpu.ids | tail -11 | tee -a $OUT/setup.txt
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_badyn_gpu.py -x -q 2>&1 | grep -iE "passed|failed|error" | tee $OUT/pytest_ba.txt
