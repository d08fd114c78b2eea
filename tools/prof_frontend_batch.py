import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import vido_slam_amd as V
if os.environ.get('VIDO_LIB_PATH'): V.host.LIB_PATH = os.environ['VIDO_LIB_PATH']
from vido_slam_amd import synth
B = 64
ctx = V.Context(width=640, height=480, max_batch=B)
seq = synth.Sequence(n_frames=16, w=640, h=480, seed=1)
g = np.stack([seq.frame(k % 16)[0] for k in range(B)])
gd = torch.from_numpy(g).cuda(); torch.cuda.synchronize()
for _ in range(4):
    ctx.orb_extract_batch((gd.data_ptr(), B, 480, 640, 480 * 640, 640), reuse=True)
print(ctx.orb_timing())

import ctypes
lib = ctypes.CDLL(V.host.LIB_PATH)
if hasattr(lib, "vido_debug_fs_prof"):
    out = (ctypes.c_ulonglong * 16)()
    lib.vido_debug_fs_prof(out, 1)
    ctx.orb_extract_batch((gd.data_ptr(), B, 480, 640, 480 * 640, 640), reuse=True); torch.cuda.synchronize()
    lib.vido_debug_fs_prof(out, 0)
    v = list(out)
    names = ["a_iters", "a_tasks", "exp_iters", "exp_quads", "b_iters", "b_cands", "nms_iters", "nms_corners", "chunks", "redo_aq", "redo_cand", "strips", "passes", "pass_px", "strip_px"]
    print({n: x for n, x in zip(names, v)})
    print("fill a %.2f exp %.2f b %.2f nms %.2f" % (v[1] / 64 / max(v[0], 1), v[3] / 64 / max(v[2], 1), v[5] / 64 / max(v[4], 1), v[7] / 64 / max(v[6], 1)))
    print("est VALU: a %d exp %d b %d nms %d" % (v[0] * 81, v[2] * 35, v[4] * 155, v[6] * 56))
