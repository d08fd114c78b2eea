#!/bin/bash
for s in 0 2 4 6 1 3 5 7; do echo "ablate $s"; VIDO_GCONV_ABLATE=$s timeout 200 python tools/prof_gconv.py 2>&1 | grep "cpg  8\|cpg 16" | cut -c60-105; done
