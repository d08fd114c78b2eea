#!/bin/bash
# end-of-round validation on the GPU box: the whole -m gpu suite, smoke(), the default bench line, the round-2 profiles
cd /root/repo; mkdir -p gpurun_out/final
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/final/pytest.txt; cat gpurun_out/final/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 500 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -2 gpurun_out/final/bench.err
tools/profile_round2.sh 2>&1 | tail -3
