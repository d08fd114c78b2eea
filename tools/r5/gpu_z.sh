#!/bin/bash
# round 5, call z: run-to-run spread of the default bench line on one box (three default runs)
set -u
OUT=gpurun_out/r5z; mkdir -p $OUT
for i in 1 2 3; do timeout 900 python bench.py 2> $OUT/err$i.txt > $OUT/bench$i.json; python -c "
import json; d = json.load(open('$OUT/bench$i.json')); print('run $i:', d['value'], 'frames/s', d['ms_per_step'], 'ms | without detector', d['extra']['e2e_without_detector']['frames_per_s'], '| global BA', d['global_ba_iters_per_s']['wall'], d['global_ba_iters_per_s']['lm_loop'], '| front end', d['extra']['configs1_frontend_batched']['frames_per_s'], '| FAST frac', d['roofline']['frac'])" | tee -a $OUT/spread.txt; done
