import sys, time; sys.path.insert(0,'.')
import numpy as np, vido_slam_amd as V, torch
from vido_slam_amd import synth
B=64
seq = synth.Sequence(n_frames=16, seed=1)
g = np.stack([seq.frame(k%16)[0] for k in range(B)])
gd = torch.from_numpy(g).cuda()
for nt in (1, 4, 8, 16, 32, 64):
    ctx = V.Context(max_batch=B, host_threads=nt)
    arg=(gd.data_ptr(), B, 480, 640, 480*640, 640)
    for _ in range(3): ctx.orb_extract_batch(arg)
    t=time.perf_counter(); 
    for _ in range(10): ctx.orb_extract_batch(arg)
    dt=(time.perf_counter()-t)/10
    print(nt, 'py wall %.2f ms'%(dt*1e3), {k: round(v,2) for k,v in ctx.orb_timing().items()})
    ctx.close()
