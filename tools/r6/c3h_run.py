"""One shape of the direct split-fp16 3x3 kernel a few times (for rocprofv3): python tools/r6/c3h_run.py N cin cout H W [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv3x3_h
N, cin, cout, H, W = (int(a) for a in sys.argv[1:6]); reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
x = torch.randn(N, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5); b = torch.randn(cout, device="cuda")
wp = pack_conv3x3_h(w).cuda()
for _ in range(reps):
    y = ops.conv3x3_h_bias_act(x, wp, b, cout, 0.0)
torch.cuda.synchronize()
