#!/bin/bash
# Profiles kept under profiles/r4/ (run on the GPU box through gpurun, from the repo root; sections selectable: tools/profile_round4.sh [tests] [bench] [e2e] [fe] [nets] [wino]):
#   tests  python -m pytest tests -m gpu                                                            -> pytest_gpu.txt
#   bench  the headline line: python bench.py (defaults) -> bench_e2e.json; and --steps 200 -> bench_e2e_200.json
#   e2e    rocprofv3 --kernel-trace --stats of the headline command                                   -> e2e_kernel_stats.csv + bench_under_rocprof.json
#   fe     rocprofv3 --kernel-trace --stats of the batched front end (tools/prof_frontend_batch.py)   -> frontend_kernel_stats.csv (k_fast_strips: the `roofline` kernel)
#   nets   kernel timelines of the LiteFlowNet / MonoDepth2 graphs and of the one-graph detector      -> nets_timeline_summary.txt, det_timeline_summary.txt
#          rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of the three network nodes         -> nets_mfma.json
#   wino   rocprofv3 --kernel-trace --stats of tools/prof_wino.py / tools/prof_conv1x1.py             -> wino_kernel_stats.csv, conv1x1_kernel_stats.csv (+ the tools' own tables)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r4; mkdir -p $OUT
WHAT="${*:-tests bench e2e fe nets wino}"
has() { case " $WHAT " in *" $1 "*) return 0;; *) return 1;; esac; }
stats() { f=$(find $1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $2; }
if has tests; then timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_full.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_full.txt | tail -5 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt; fi
if has bench; then
  timeout 900 python bench.py > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err; echo "bench rc $?"
  timeout 900 python bench.py --steps 200 --warmup 10 --no-extra --cpu-baseline 0 > $OUT/bench_e2e_200.json 2> $OUT/bench_e2e_200.err; echo "bench200 rc $?"
fi
export TMPDIR=/tmp; cd /tmp
if has e2e; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -o e2e -- python $REPO/bench.py --steps 20 --warmup 3 --cpu-baseline 0 --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/e2e.err
  stats $OUT/e2e $OUT/e2e_kernel_stats.csv
fi
if has fe; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fe -o fe -- python $REPO/tools/prof_frontend_batch.py > $OUT/fe.log 2>&1
  stats $OUT/fe $OUT/frontend_kernel_stats.csv
fi
if has nets; then
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python $REPO/tools/prof_lfn_timeline.py > $OUT/tl.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tld -o tl -- python $REPO/tools/prof_det_timeline.py > $OUT/tld.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o p -- python $REPO/tools/nets_pmc3.py > $OUT/mfma.log 2>&1
  cd $REPO
  python tools/summarize_timeline.py $(find $OUT/tl -name "*kernel_trace.csv" | head -1) "flow,flow,flow,flow,depth,depth,depth,depth" 40 > $OUT/nets_timeline_summary.txt 2>&1
  python tools/summarize_timeline.py $(find $OUT/tld -name "*kernel_trace.csv" | head -1) "det,det,det,det,det,det" 45 > $OUT/det_timeline_summary.txt 2>&1
  NETS_PMC_COMMAND="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/nets_pmc3.py (pipeline.NetNodes at 640x480, eager, 2 frames: LiteFlowNet, MonoDepth2, Mask R-CNN X-101-FPN, fp32)" python tools/nets_pmc.py --summarise $OUT/mfma $OUT/nets_mfma.json > $OUT/mfma_summary.txt 2>&1
  cd /tmp
fi
if has wino; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wn -o wn -- python $REPO/tools/prof_wino.py > $OUT/wino_microbench_under_rocprof.txt 2>&1
  stats $OUT/wn $OUT/wino_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c1 -o c1 -- python $REPO/tools/prof_conv1x1.py > $OUT/conv1x1_microbench_under_rocprof.txt 2>&1
  stats $OUT/c1 $OUT/conv1x1_kernel_stats.csv
fi
cd $REPO
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.csv" -size +2M -delete
ls $OUT
