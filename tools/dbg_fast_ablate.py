import sys, os; sys.path.insert(0,'.')
import numpy as np, vido_slam_amd as V, torch
from vido_slam_amd import synth
B=64
seq = synth.Sequence(n_frames=16, seed=1)
g = np.stack([seq.frame(k%16)[0] for k in range(B)])
gd = torch.from_numpy(g).cuda()
ctx = V.Context(max_batch=B, host_threads=32)
arg=(gd.data_ptr(), B, 480, 640, 480*640, 640)
for ab in (0,1,2,3,4,7):
    os.environ["VIDO_ABLATE"]=str(ab)
    ts=[]
    for _ in range(6):
        try: ctx.orb_extract_batch(arg)
        except Exception as e: pass
        ts.append(ctx.orb_timing()['fast_ms'])
    print('ablate',ab,'fast_ms %.3f'%min(ts))
