#!/bin/bash
# Profiles of the bench command for profiles/<round>/ (run on the GPU box through gpurun, from the repo root):
#   1. rocprofv3 --kernel-trace --stats     -> <round>_kernel_stats.csv (+ the bench line printed under the profiler)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md, HBM section; never with sys/hip traces)
# Usage: tools/profile_round.sh r1
set -u
ROUND=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$ROUND
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --cpu-frames 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python $REPO/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-extra > $OUT/bench_pmc_$C.json 2> $OUT/pmc_$C.err
done
cd $REPO
python tools/summarize_profiles.py $OUT $ROUND
