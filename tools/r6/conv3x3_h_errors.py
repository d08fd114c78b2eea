import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv3x3_h, pack_wino3x3
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
torch.backends.cudnn.allow_tf32 = False
for N, cin, cout, H, W, scaled in ((1,256,256,50,68,True),(1,256,256,50,68,False),(1,256,256,100,136,True),(5,256,256,14,14,True),(1,64,384,33,17,True),(2,32,128,13,21,True)):
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(N, cin, H, W, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    if scaled: w = w * torch.exp(torch.randn(cout, 1, 1, 1, generator=g))
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    sc = ref.abs().mean((0, 2, 3), keepdim=True)
    xc, bc = x.cuda(), b.cuda()
    yh = ops.conv3x3_h_bias_act(xc, pack_conv3x3_h(w).cuda(), bc, cout, 1.0).cpu().double()
    form = ops.wino3x3_form(N, cin, cout, H, W)
    yw = ops.wino3x3_bias_act(xc, pack_wino3x3(w, form).cuda(), bc, cout, 1.0, form).cpu().double()
    yl = F.conv2d(xc, w.cuda(), bc, padding=1).cpu().double()
    yc = F.conv2d(x, w, b, padding=1).double()
    f = lambda y: (float(((y - ref) / sc).pow(2).mean().sqrt()), float((y - ref).abs().max()))
    print((N, cin, cout, H, W, scaled), "direct-f16 rms %.3e max %.3e | wino form %d rms %.3e max %.3e | library fp32 rms %.3e max %.3e | cpu fp32 rms %.3e max %.3e" % (f(yh) + (form,) + f(yw) + f(yl) + f(yc)))
