#!/bin/bash
# 16-channel stride-2 grouped convolution: parity + microbench
timeout 400 python -m pytest tests/test_maskrcnn_gpu.py -q 2>&1 | tail -4
timeout 100 python - <<'PY'
import torch, torch.nn.functional as F, sys, os
sys.path.insert(0, os.getcwd())
import vido_slam_amd as vido
from vido_slam_amd.nets.ops import HipOps, pack_gconv3x3
ctx = vido.Context(); ops = HipOps(ctx)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for cpg, H, W in ((16, 200, 272), (32, 100, 136), (64, 50, 68)):
    G = 32; C = G * cpg
    x = torch.randn(1, C, H, W, device="cuda"); w = torch.randn(C, cpg, 3, 3, device="cuda") * 0.05; b = torch.randn(C, device="cuda")
    wp = pack_gconv3x3(w, G)
    t_conv = timeit(lambda: F.conv2d(x, w, None, 2, 1, 1, G))
    t_lib = timeit(lambda: ops.bias_res_act_(F.conv2d(x, w, None, 2, 1, 1, G), b, None, 0.0))
    t_new = timeit(lambda: ops.gconv3x3_s2_bias_act(x, wp, b, G, 0.0))
    print(f"stride 2 cpg {cpg} {H}x{W}: library conv {t_conv:6.1f} us (+ bias pass {t_lib:6.1f}) | gconv.hip {t_new:6.1f} us", flush=True)
PY
