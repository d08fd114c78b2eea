import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import synth
B, W, H = 64, 640, 480
ctx = V.Context(width=W, height=H, max_batch=B)
tp = V.track_params(dataset=0, depth_map_factor=1.0, th_depth_bg=40.0, th_depth_obj=25.0)
ff = V.FrameFeatures(ctx, tp)
seq = synth.Sequence(n_frames=16, w=W, h=H, seed=1); fr = [seq.frame(k) for k in range(16)]; sel = np.arange(B) % 16
gray_d = torch.from_numpy(np.ascontiguousarray(np.stack([fr[i][0] for i in sel]))).cuda()
depth_d = torch.from_numpy(np.ascontiguousarray(np.stack([fr[i][2] for i in sel]).astype(np.float32))).cuda()
flow_d = torch.from_numpy(np.ascontiguousarray(np.stack([fr[i][3] for i in sel]))).cuda()
mask_d = torch.from_numpy(np.ascontiguousarray(np.stack([fr[i][4] for i in sel]))).cuda()
depth_work = torch.empty_like(depth_d)
dev_arg = (gray_d.data_ptr(), B, H, W, H * W, W)
T = np.zeros(4)
for it in range(25):
    t0 = time.perf_counter()
    kps, desc, cnt = ctx.orb_extract_batch(dev_arg, want_desc=True, reuse=True)
    t1 = time.perf_counter()
    depth_work.copy_(depth_d)
    ctx._check(ctx.lib.vido_frame_upload(ctx.h, 0, B, C.c_void_p(depth_work.data_ptr()), C.c_void_p(flow_d.data_ptr()), C.c_void_p(mask_d.data_ptr()), 1, C.byref(tp)))
    t2 = time.perf_counter()
    lists = ff.features(0, kps, cnt, reuse=True)
    t3 = time.perf_counter()
    if it >= 5:
        T += [t1 - t0, t2 - t1, t3 - t2, t3 - t0]
print("ms: orb %.3f upload %.3f features %.3f total %.3f" % tuple(T / 20 * 1e3))
print(ctx.orb_timing())
