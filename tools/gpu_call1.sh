#!/bin/bash
# round 2, GPU call 1: parity tests, the new end-to-end bench line, network variants, kernel trace of the chain
set -x
mkdir -p gpurun_out/r2a
cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2a/pytest.log
tail -15 gpurun_out/r2a/pytest.log
timeout 600 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc $?"
tail -c 3000 gpurun_out/r2a/bench.err
timeout 400 python tools/exp_nets.py base fold graphs > gpurun_out/r2a/exp_nets.jsonl 2> gpurun_out/r2a/exp_nets.err; echo "exp rc $?"
cat gpurun_out/r2a/exp_nets.jsonl
timeout 500 python tools/exp_nets.py find > gpurun_out/r2a/exp_find.jsonl 2> gpurun_out/r2a/exp_find.err; echo "find rc $?"
cat gpurun_out/r2a/exp_find.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r2a/prof -o e2e -- python /root/repo/bench.py --no-extra --cpu-baseline 0 --steps 10 > /root/repo/gpurun_out/r2a/bench_prof.json 2> /root/repo/gpurun_out/r2a/bench_prof.err; echo "prof rc $?"
cd /root/repo
find gpurun_out/r2a/prof -name "*kernel_stats*" | head; find gpurun_out/r2a/prof -name "*.csv" ! -name "*stats*" -delete; find gpurun_out/r2a/prof -name "*.db" -delete
cat gpurun_out/r2a/bench.json | head -c 6000
