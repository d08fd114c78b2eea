#!/bin/bash
# VIDO_CONVDIRECT_SET novalu (default) against all, now that the detector runs in split fp16: headline and the no-detector chain.
for i in 1 2; do
for s in novalu all; do
echo "== $s headline"; VIDO_CONVDIRECT_SET=$s timeout 600 python bench.py --steps 100 --warmup 10 --no-extra --cpu-baseline 0 2>&1 | tail -1 | cut -c90-130
echo "== $s no-detector chain"; VIDO_CONVDIRECT_SET=$s timeout 300 python tools/prof_nodet.py 120 2>&1 | tail -1 | cut -c1-60
done; done
