"""Which bottlenecks take the strided grouped convolution of csrc/gconv.hip?  Runs the detector's backbone once, eagerly, and prints every stage's first block decision."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import vido_slam_amd as vido
from vido_slam_amd import nets as _nets
from vido_slam_amd.nets.ops import HipOps
from vido_slam_amd.nets.fuse import fold_batchnorm
from vido_slam_amd.nets import maskrcnn as M
ctx = vido.Context(); ops = HipOps(ctx)
net = _nets.fill_maskrcnn(_nets.MaskRCNN(ops), 3).eval().cuda()
fold_batchnorm(net, ops)
calls = []
orig = ops.gconv3x3_s2_bias_act
def spy(x, *a, **k):
    calls.append(tuple(x.shape)); return orig(x, *a, **k)
ops.gconv3x3_s2_bias_act = spy
for name, m in net.named_modules():
    if isinstance(m, M._Bottleneck) and tuple(m.conv2.stride) == (2, 2):
        print(name, "w2p", m._w2p is not None, "w1p", m._w1p is not None, "ops is", m._ops is ops, flush=True)
x = torch.randn(1, 3, 800, 1088, device="cuda")
with torch.no_grad():
    feats = net.backbone(x) if hasattr(net, "backbone") else None
print("s2 calls:", calls)
