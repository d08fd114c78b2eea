import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd import nets
ctx = V.Context(); ops = nets.HipOps(ctx)
mr = nets.fill_maskrcnn(nets.MaskRCNN(ops), 3).eval().cuda()
bgr = (np.random.RandomState(0).rand(375, 1242, 3) * 255).astype(np.uint8)
def T(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3, r
with torch.no_grad():
    ms, t = T(lambda: torch.as_tensor(bgr[:, :, ::-1].copy(), device="cuda").permute(2, 0, 1).float().unsqueeze(0)); print("upload+float", ms)
    ms, x = T(lambda: F.interpolate(t, size=(1088, 800), mode="area")); print("area resize", ms)
    ms, out = T(lambda: mr(x)); print("net", ms, "dets", len(out["labels"]), "props", len(out["proposals"]))
    ms, feats = T(lambda: mr.backbone(x)); print("backbone", ms)
    ms, pr = T(lambda: mr.rpn(feats, (800, 1088))); print("rpn", ms)
    ms, det = T(lambda: mr.roi_heads.box(feats[:4], pr[0], (800, 1088))); print("box head", ms)
    ms, mk = T(lambda: mr.roi_heads.mask(feats[:4], det[0], det[2])); print("mask head", ms)
    ms, _ = T(lambda: nets.paste_masks(out["masks"], out["boxes"], 375, 1242)); print("paste", ms)
    ms, _ = T(lambda: nets.analyse_image(mr, bgr)); print("analyse_image", ms)
