#!/bin/bash
# oracle/_ref recipe for the one part of the reference that is ATen-only C++ (no OpenCV / Eigen / CXSparse):
#   src/thirdparty/mask_rcnn/maskrcnn_benchmark/csrc/cpu/ROIAlign_cpu.cpp  and  nms_cpu.cpp
# compiled WHERE THEY LIE under /root/reference against the torch headers of this image, objects only into oracle/_ref/.
# TEST INFRASTRUCTURE ONLY.  Nothing is copied, patched or stubbed: if the sources do not compile against this torch,
# the recipe says so (oracle/_ref/maskrcnn_cpu.log) and exits 3 — the ROI-Align / NMS restatements in oracle/nets_oracle.c
# then stay pinned by the reference's own known-answer tests only (tests/golden/maskrcnn_kats.npz), see DESIGN.md section 2.
#
# Outcome in this image (torch 2.10.0, g++ 11): BOTH files are rejected — they pass `tensor.type()` (at::DeprecatedTypeProperties)
# to AT_DISPATCH_FLOATING_TYPES, which since torch 2.x only takes a c10::ScalarType:
#   torch/headeronly/core/Dispatch.h:36:63: error: cannot convert 'const at::DeprecatedTypeProperties' to 'c10::ScalarType'
# so oracle/_ref cannot be built without editing the reference sources (not allowed) -> unbuildable here.
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${VIDO_REFERENCE:-/root/reference}/src/thirdparty/mask_rcnn/maskrcnn_benchmark/csrc
OUT=$HERE/_ref
mkdir -p "$OUT"
LOG=$OUT/maskrcnn_cpu.log
: > "$LOG"
if [ ! -d "$REF/cpu" ]; then echo "reference not present ($REF): nothing to build" | tee -a "$LOG"; exit 0; fi
TI=$(python3 -c "import torch, os; print(os.path.dirname(torch.__file__))")
ABI=$(python3 -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
PYI=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
ok=1
for f in nms_cpu ROIAlign_cpu; do
    if ! g++ -std=c++17 -O2 -fPIC -c "$REF/cpu/$f.cpp" -I"$REF" -I"$TI/include" -I"$TI/include/torch/csrc/api/include" -I"$PYI" \
            -D_GLIBCXX_USE_CXX11_ABI=$ABI -o "$OUT/$f.o" >> "$LOG" 2>&1; then
        ok=0; echo "[$f.cpp] does not compile against torch $(python3 -c 'import torch; print(torch.__version__)'):" | tee -a "$LOG"
        grep -m 3 "error" "$LOG"
    fi
done
if [ $ok = 1 ]; then
    g++ -shared -o "$OUT/libmaskrcnn_cpu_ref.so" "$OUT/nms_cpu.o" "$OUT/ROIAlign_cpu.o" -L"$TI/lib" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$TI/lib" >> "$LOG" 2>&1 && echo "built $OUT/libmaskrcnn_cpu_ref.so"
    exit 0
fi
exit 3
