import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np
import vido_slam_amd as V
from vido_slam_amd import nets, synth
ctx = V.Context(width=640, height=480, max_batch=1); ops = nets.HipOps(ctx)
scene = synth.convoy_scene(n_frames=3) if hasattr(synth, "convoy_scene") else None
g = scene.frame(1)[0] if scene is not None else (np.random.rand(480, 640) * 255).astype(np.uint8)
bgr = torch.from_numpy(synth.gray_to_bgr(g)).cuda()
for rs in (0.05, 0.02, 0.01, 0.005):
    net = nets.fill_maskrcnn(nets.MaskRCNN(ops), 3, reg_scale=rs).eval().cuda()
    with torch.no_grad():
        out = net(nets.maskrcnn.image_to_feed(bgr, "cuda", (1088, 800)))
    s = out["scores"].float().cpu().numpy()
    print("reg_scale %.3f: detections %d  n_proposals %d  scores min %.6f max %.6f  distinct %d  >0.8: %d" % (rs, len(s), int(out["n_proposals"]), s.min() if len(s) else 0, s.max() if len(s) else 0, len(np.unique(s)), int((s > 0.8).sum())))
    del net
