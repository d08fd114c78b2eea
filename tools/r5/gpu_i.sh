#!/bin/bash
# round 5, call i: the window solve beside the next frame: facade / system / e2e / pipeline tests, the literal chain, the headline
set -u
OUT=gpurun_out/r5i; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_facade_gpu.py tests/test_system_gpu.py tests/test_e2e_gpu.py tests/test_pipeline_gpu.py tests/test_ba_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest.txt
for mode in "async" "VIDO_LBA_SYNC=1"; do echo "== $mode" | tee -a $OUT/nodet.txt
  if [ "$mode" = async ]; then timeout 300 python tools/prof_nodet.py 80 2>&1 | grep frames_per_s | cut -c1-600 | tee -a $OUT/nodet.txt
  else env $mode timeout 300 python tools/prof_nodet.py 80 2>&1 | grep frames_per_s | cut -c1-600 | tee -a $OUT/nodet.txt; fi; done
for mode in "async" "VIDO_LBA_SYNC=1"; do echo "== $mode" | tee -a $OUT/bench.txt
  if [ "$mode" = async ]; then timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['track_total_ms'], d['stage_ms']['local_ba_ms'], d['pose_translation_error_m'])" | tee -a $OUT/bench.txt
  else env $mode timeout 600 python bench.py --steps 100 --warmup 5 --no-extra --cpu-baseline 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['track_total_ms'], d['stage_ms']['local_ba_ms'], d['pose_translation_error_m'])" | tee -a $OUT/bench.txt; fi; done
