import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vido_slam_amd as V
from vido_slam_amd import nets
ctx = V.Context()
ops = nets.HipOps(ctx)
def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
x = torch.rand(1, 3, 1088, 800, device="cuda") * 255
mr = nets.fill_maskrcnn(nets.MaskRCNN(ops), 3).eval().cuda()
with torch.no_grad():
    print("maskrcnn backbone NCHW ms", timed(lambda: mr.backbone(x)))
    print("maskrcnn full NCHW ms", timed(lambda: mr(x)))
    mr2 = mr.to(memory_format=torch.channels_last); xc = x.contiguous(memory_format=torch.channels_last)
    print("maskrcnn backbone NHWC ms", timed(lambda: mr2.backbone(xc)))
    a = mr.backbone(x); b = mr2.backbone(xc)
    print("max rel diff", max(float((p - q).abs().max() / p.abs().max()) for p, q in zip(a, b)))
    for name, enabled in (("benchmark", True),):
        torch.backends.cudnn.benchmark = enabled
        print("maskrcnn backbone NCHW benchmark=True ms", timed(lambda: mr.backbone(x)))
lfn = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation), 1).eval().cuda()
a = torch.rand(1, 3, 384, 1248, device="cuda"); b = torch.rand(1, 3, 384, 1248, device="cuda")
print("liteflownet ms", timed(lambda: lfn(a, b)))
