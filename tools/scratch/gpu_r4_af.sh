#!/bin/bash
for s in 0 1; do echo "ablate $s force8"; VIDO_GCONV_FORCE_NTW8=1 VIDO_GCONV_STAGGER=$s timeout 200 python tools/prof_gconv.py 2>&1 | grep "cpg  8\|cpg 16" | cut -c60-125; done
