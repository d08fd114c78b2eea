#!/bin/bash
# Profiles kept under profiles/r2/ (run on the GPU box through gpurun, from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the headline command (bench.py, end-to-end chain)          -> e2e_kernel_stats.csv + the bench line printed under the profiler
#   2. rocprofv3 --kernel-trace --stats of the batched front end (tools/prof_frontend_batch.py, 64 frames)   -> frontend_kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md, HBM section; never with sys/hip traces) of the front end, the local BA
#      window and the global BA (tools/prof_ba_local.py, tools/prof_ba_global.py)                              -> pmc_traffic.json, pmc_traffic_ba.json, pmc_traffic_ba_global.json
#   4. SQ counters of k_fast_strips (tools/pmc_fast.sh)                                                 -> fast_sq_counters.txt
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r2; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -o e2e -- python $REPO/bench.py --steps 20 --warmup 3 --cpu-baseline 0 --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/e2e.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fe -o fe -- python $REPO/tools/prof_frontend_batch.py > $OUT/fe.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_fe_$C -o p -- python $REPO/tools/prof_frontend_batch.py > $OUT/pmc_fe_$C.log 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_bal_$C -o p -- python $REPO/tools/prof_ba_local.py > $OUT/pmc_bal_$C.log 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_bag_$C -o p -- python $REPO/tools/prof_ba_global.py > $OUT/pmc_bag_$C.log 2>&1
done
cd $REPO
tools/pmc_fast.sh $REPO/vido-slam_amd/libvido_slam_hip.so gpurun_out/prof_r2/sq > $OUT/fast_sq_counters.txt 2>&1
python tools/summarize_profiles2.py $OUT
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +2M -delete
