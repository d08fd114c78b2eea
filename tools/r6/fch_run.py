"""fc6 through the split-fp16 FC kernel a few times (for rocprofv3): python tools/r6/fch_run.py rows k outs [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
rows, k, outs = (int(a) for a in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
x = torch.relu(torch.randn(rows, k, device="cuda")); w = torch.randn(outs, k) / k ** 0.5; b = torch.randn(outs, device="cuda")
wp = pack_conv1x1(w.reshape(outs, k, 1, 1), 3).cuda()
for _ in range(reps):
    y = ops.fc_h(x, wp, b, outs, 0.0)
torch.cuda.synchronize()
