#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r2j
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2j/pytest.txt; cat gpurun_out/r2j/pytest.txt
timeout 400 python bench.py > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err; tail -3 gpurun_out/r2j/bench.err
tools/profile_round2.sh 2>&1 | tail -5
