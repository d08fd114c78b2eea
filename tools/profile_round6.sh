#!/bin/bash
# Profiles kept under profiles/r6/ (run on the GPU box through gpurun, from the repo root; sections selectable: tools/profile_round6.sh [tests] [bench] [e2e] [fe] [pmc] [sq] [nets] [ba] [nodet] [c1]):
#   tests  python -m pytest tests -m gpu                                                            -> pytest_gpu.txt
#   bench  the headline line: python bench.py (defaults) -> bench_e2e.json; and --steps 200 -> bench_e2e_200.json
#   e2e    rocprofv3 --kernel-trace --stats of the headline command                                   -> e2e_kernel_stats.csv + bench_under_rocprof.json
#   fe     rocprofv3 --kernel-trace --stats of the batched front end (tools/prof_frontend_batch.py)   -> frontend_kernel_stats.csv (k_fast_strips: the `roofline` kernel)
#   pmc    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md: never with sys / hip traces) of the front end and the global BA
#                                                                                                    -> pmc_traffic.json, pmc_traffic_ba_global.json   (read by bench.py)
#   sq     SQ counters of k_fast_strips (tools/pmc_fast.sh)                                          -> fast_sq_counters.txt                            (read by bench.py)
#   nets   kernel timelines of the LiteFlowNet / MonoDepth2 graphs and of the one-graph detector; matrix-pipe busy share of the three networks
#                                                                                                    -> nets_timeline_summary.txt, det_timeline_summary.txt, nets_mfma.json
#   ba     rocprofv3 --kernel-trace --stats of tools/prof_ba_global.py (configs[4] size)              -> global_ba_kernel_stats.csv
#   nodet  the literal chain of the metric text (tools/prof_nodet.py): stage ms + C-ABI call profile  -> nodet_call_profile.txt, nodet_kernel_stats.csv
#   c1     tools/r6/conv1x1_b3_check.py (split-fp16 and split-bf16 vs fp32-instruction 1x1 kernel vs float64) -> conv1x1_microbench.txt; tools/ubench/valu_int_issue -> valu_int_issue.txt
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r6; mkdir -p $OUT
WHAT="${*:-tests bench e2e fe pmc sq nets ba nodet c1}"
has() { case " $WHAT " in *" $1 "*) return 0;; *) return 1;; esac; }
if has tests; then timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_full.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_full.txt | tail -5 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt; fi
if has bench; then
  timeout 900 python bench.py > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err; echo "bench rc $?"
  timeout 900 python bench.py --steps 200 --warmup 10 --no-extra --cpu-baseline 0 > $OUT/bench_e2e_200.json 2> $OUT/bench_e2e_200.err; echo "bench200 rc $?"
fi
if has c1; then      # the split-fp16 / split-bf16 1x1 kernels against the fp32-instruction one and float64 (error ratio, us, fp32-equivalent TFLOP/s), the form the library picks per shape
  timeout 300 python tools/r6/conv1x1_b3_check.py 2>&1 | grep -v amdgpu.ids > $OUT/conv1x1_microbench.txt
  timeout 120 tools/ubench/valu_int_issue.bin > $OUT/valu_int_issue.txt 2>&1
fi
if has nodet; then VIDO_CALL_PROF=1 timeout 300 python tools/prof_nodet.py 80 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $OUT/nodet_call_profile.txt; fi
export TMPDIR=/tmp; cd /tmp
if has e2e; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -o e2e -- python $REPO/bench.py --steps 20 --warmup 3 --cpu-baseline 0 --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/e2e.err
fi
if has fe; then timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fe -o fe -- python $REPO/tools/prof_frontend_batch.py > $OUT/fe.log 2>&1; fi
if has ba; then timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bag -o bag -- python $REPO/tools/prof_ba_global.py > $OUT/bag.log 2>&1; fi
if has nodet; then timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nd -o nd -- python $REPO/tools/prof_nodet.py 60 > $OUT/nd.log 2>&1; fi
if has pmc; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_fe_$C -o p -- python $REPO/tools/prof_frontend_batch.py > $OUT/pmc_fe_$C.log 2>&1
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_bag_$C -o p -- python $REPO/tools/prof_ba_global.py > $OUT/pmc_bag_$C.log 2>&1
  done
fi
if has nets; then
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python $REPO/tools/prof_lfn_timeline.py > $OUT/tl.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tld -o tl -- python $REPO/tools/prof_det_timeline.py > $OUT/tld.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o p -- python $REPO/tools/nets_pmc3.py > $OUT/mfma.log 2>&1
  cd $REPO
  python tools/summarize_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) "flow,flow,flow,flow,depth,depth,depth,depth" 40 > $OUT/nets_timeline_summary.txt 2>&1
  python tools/summarize_timeline.py $(find $OUT/tld -name "*kernel_trace.csv" | head -1) "det,det,det,det,det,det" 45 > $OUT/det_timeline_summary.txt 2>&1
  NETS_PMC_COMMAND="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/nets_pmc3.py (pipeline.NetNodes at 640x480, eager, 2 frames: LiteFlowNet, MonoDepth2, Mask R-CNN X-101-FPN, fp32)" python tools/nets_pmc.py --summarise $OUT/mfma $OUT/nets_mfma.json > $OUT/mfma_summary.txt 2>&1
fi
cd $REPO
if has sq; then tools/pmc_fast.sh $REPO/vido-slam_amd/libvido_slam_hip.so gpurun_out/prof_r6/sq > $OUT/fast_sq_counters.txt 2>&1; fi
python tools/summarize_profiles5.py $OUT 6
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.csv" -size +2M -delete
ls gpurun_out/profiles_r6
