import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import synth
B = 64
ctx = V.Context(width=640, height=480, max_batch=B)
seq = synth.Sequence(n_frames=16, w=640, h=480, seed=1)
g = np.stack([seq.frame(k % 16)[0] for k in range(B)])
gd = torch.from_numpy(g).cuda(); torch.cuda.synchronize()
for _ in range(4):
    ctx.orb_extract_batch((gd.data_ptr(), B, 480, 640, 480 * 640, 640), reuse=True)
print(ctx.orb_timing())
