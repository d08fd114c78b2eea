#!/bin/bash
# The metric text's literal chain (flow + depth + track + local BA, no detector): synchronous local BA against VIDO_LBA_ASYNC=1 with and without the stats join.
mkdir -p gpurun_out
for i in 1 2; do
echo "== sync"; timeout 300 python tools/prof_nodet.py 120 2>&1 | tail -1
echo "== async (stats joins)"; VIDO_LBA_ASYNC=1 timeout 300 python tools/prof_nodet.py 120 2>&1 | tail -1
echo "== async, stats do not join"; VIDO_LBA_ASYNC=1 VIDO_STATS_NO_JOIN=1 timeout 300 python tools/prof_nodet.py 120 2>&1 | tail -1
done
echo "== headline async no-join"; VIDO_LBA_ASYNC=1 VIDO_STATS_NO_JOIN=1 timeout 600 python bench.py --steps 100 --warmup 10 --no-extra --cpu-baseline 0 2>&1 | tail -1 | cut -c1-400
echo "== headline sync"; timeout 600 python bench.py --steps 100 --warmup 10 --no-extra --cpu-baseline 0 2>&1 | tail -1 | cut -c1-400
