#!/bin/bash
# Profiles kept under profiles/r3/ (run on the GPU box through gpurun, from the repo root):
#   1. the headline line: python bench.py (defaults) -> bench_e2e.json; and --steps 200 -> bench_e2e_200.json
#   2. rocprofv3 --kernel-trace --stats of the headline command                                   -> e2e_kernel_stats.csv + bench_under_rocprof.json
#   3. rocprofv3 --kernel-trace --stats of the batched front end (tools/prof_frontend_batch.py)   -> frontend_kernel_stats.csv
#   4. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md; never with sys/hip traces) of the front end, the local BA window and the
#      global BA                                                                                    -> pmc_traffic.json, pmc_traffic_ba.json, pmc_traffic_ba_global.json
#   5. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of the three network nodes at 640x480 -> nets_mfma.json
#   6. kernel timelines of the LiteFlowNet / MonoDepth2 graphs and of the one-graph detector        -> nets_timeline_summary.txt, det_timeline_summary.txt
#      csrc/gconv.hip against the library convolution (tools/prof_gconv.py)                          -> gconv_kernel_stats.csv, gconv_microbench.txt
#   7. SQ counters of k_fast_strips (tools/pmc_fast.sh)                                             -> fast_sq_counters.txt
#   8. python -m pytest tests -m gpu                                                                -> pytest_gpu.txt
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r3; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_full.txt 2>&1; tail -12 $OUT/pytest_full.txt > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err; echo "bench rc $?"
timeout 900 python bench.py --steps 200 --warmup 10 --no-extra --cpu-baseline 0 > $OUT/bench_e2e_200.json 2> $OUT/bench_e2e_200.err; echo "bench200 rc $?"
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -o e2e -- python $REPO/bench.py --steps 20 --warmup 3 --cpu-baseline 0 --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/e2e.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fe -o fe -- python $REPO/tools/prof_frontend_batch.py > $OUT/fe.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_fe_$C -o p -- python $REPO/tools/prof_frontend_batch.py > $OUT/pmc_fe_$C.log 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_bal_$C -o p -- python $REPO/tools/prof_ba_local.py > $OUT/pmc_bal_$C.log 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_bag_$C -o p -- python $REPO/tools/prof_ba_global.py > $OUT/pmc_bag_$C.log 2>&1
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o p -- python $REPO/tools/nets_pmc3.py > $OUT/mfma.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python $REPO/tools/prof_lfn_timeline.py > $OUT/tl.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gc -o gc -- python $REPO/tools/prof_gconv.py > $OUT/gconv_microbench.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tld -o tl -- python $REPO/tools/prof_det_timeline.py > $OUT/tld.log 2>&1
cd $REPO
NETS_PMC_COMMAND="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/nets_pmc3.py (pipeline.NetNodes at 640x480, eager, 2 frames: LiteFlowNet, MonoDepth2, Mask R-CNN X-101-FPN, fp32)" python tools/nets_pmc.py --summarise $OUT/mfma $OUT/nets_mfma.json > $OUT/mfma_summary.txt 2>&1
python tools/summarize_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) "flow,flow,flow,flow,depth,depth,depth,depth" 40 > $OUT/nets_timeline_summary.txt 2>&1
python tools/summarize_timeline.py $(find $OUT/tld -name "*kernel_trace.csv" | head -1) "det,det,det,det,det,det" 45 > $OUT/det_timeline_summary.txt 2>&1
tools/pmc_fast.sh $REPO/vido-slam_amd/libvido_slam_hip.so gpurun_out/prof_r3/sq > $OUT/fast_sq_counters.txt 2>&1
python tools/summarize_profiles3.py $OUT
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +2M -delete
