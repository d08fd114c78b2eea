"""Rows N1/N2 on the MI355X: the torch modules with the HIP cost-volume kernel (through the C-ABI, on torch's
stream) against the golden outputs of the reference modules."""
import os
import numpy as np
import pytest
import torch
from vido_slam_amd import nets

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "nets_kats.npz"))
TOL = 2e-4            # relative to the output's max magnitude (fp32 MIOpen convolutions vs the CPU reference run)


@pytest.fixture(scope="module")
def ctx(vido):
    c = vido.Context()
    yield c
    c.close()


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def test_liteflownet_hip_correlation_matches_reference(ctx):
    ops = nets.HipOps(ctx)
    net = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation), int(G["lfn_seed"])).eval().cuda()
    a = torch.from_numpy(G["lfn_first"].astype(np.float32) / 255.0)[None].cuda(); b = torch.from_numpy(G["lfn_second"].astype(np.float32) / 255.0)[None].cuda()
    flow = net(a, b).cpu().numpy()
    assert rel_err(flow, G["lfn_flow"]) < TOL
    # fused conv epilogue (bias + LeakyReLU as one HIP pass) gives the same flow
    net_f = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation, epilogue=ops.bias_act_), int(G["lfn_seed"])).eval().cuda()
    assert rel_err(net_f(a, b).cpu().numpy(), G["lfn_flow"]) < TOL
    # non-default torch stream: the library must enqueue on it (no sync between the torch convs and the HIP op)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        flow2 = net(a, b)
    s.synchronize()
    assert rel_err(flow2.cpu().numpy(), G["lfn_flow"]) < TOL


def test_liteflownet_fused_regularisation_passes(ctx):
    """vido_lfn_reg_front / vido_lfn_reg_tail (everything of Regularization.forward outside its convolutions, layers.py:213-262, as two HIP passes) against the torch
    expressions they replace, at the five levels' kernel sizes, and the whole network with them against the reference fixture."""
    import torch.nn.functional as F
    from vido_slam_amd.nets.liteflownet import backwarp
    ops = nets.HipOps(ctx)
    g = torch.Generator().manual_seed(11)
    for k, (H, W) in ((3, (8, 10)), (3, (15, 20)), (5, (30, 40)), (5, (60, 80)), (7, (97, 131))):
        nd = k * k
        dist = torch.randn((2, nd, H, W), generator=g).cuda(); flow = (torch.randn((2, 2, H, W), generator=g) * 3).cuda()
        sx = torch.nn.Conv2d(nd, 1, 1).cuda(); sy = torch.nn.Conv2d(nd, 1, 1).cuda()
        with torch.no_grad():
            d = dist.pow(2.0).neg(); d = (d - d.max(1, True)[0]).exp(); div = d.sum(1, True).reciprocal(); pad = (k - 1) // 2
            ref = torch.cat([sx(d * F.unfold(flow[:, 0:1], k, 1, pad).view_as(d)) * div, sy(d * F.unfold(flow[:, 1:2], k, 1, pad).view_as(d)) * div], 1)
            got = ops.lfn_reg_tail(dist, flow, sx, sy, k)
        assert float((got - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())), k
        im1 = torch.rand((2, 3, H, W), generator=g).cuda(); im2 = torch.rand((2, 3, H, W), generator=g).cuda(); feat = torch.randn((2, 5, H, W), generator=g).cuda()
        with torch.no_grad():
            diff = (im1 - backwarp(im2, flow * 2.5)).pow(2.0).sum(1, True).sqrt()
            ref = torch.cat([diff, flow - flow.flatten(2).mean(2, True).unsqueeze(-1), feat], 1)
            got = ops.lfn_reg_front(im1, im2, flow, 2.5, feat)
        assert float((got - ref).abs().max()) < 2e-5, k
    for Cc, (H, W) in ((2, (15, 20)), (49, (31, 41)), (49, (120, 160))):
        x = torch.randn((2, Cc, H, W), generator=g).cuda()
        dc = torch.nn.ConvTranspose2d(Cc, Cc, 4, 2, 1, bias=False, groups=Cc).cuda()
        with torch.no_grad():
            assert float((ops.deconv4s2_depthwise(x, dc.weight) - dc(x)).abs().max()) < 1e-5
            assert float((ops.deconv4s2_depthwise(x, dc.weight, 0.1) - dc(F.leaky_relu(x, 0.1))).abs().max()) < 1e-5
    net = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation, epilogue=ops.bias_act_, warp=ops.backwarp, fused=ops), int(G["lfn_seed"])).eval().cuda()
    a = torch.from_numpy(G["lfn_first"].astype(np.float32) / 255.0)[None].cuda(); b = torch.from_numpy(G["lfn_second"].astype(np.float32) / 255.0)[None].cuda()
    assert rel_err(net(a, b).cpu().numpy(), G["lfn_flow"]) < TOL
    net_p = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation, epilogue=ops.bias_act_, warp=ops.backwarp, fused=ops, pair_batch=True), int(G["lfn_seed"])).eval().cuda()
    assert rel_err(net_p(a, b).cpu().numpy(), G["lfn_flow"]) < TOL


def test_hip_correlation_matches_torch_reference(ctx):
    ops = nets.HipOps(ctx)
    g = torch.Generator().manual_seed(5)
    for stride, shape in ((1, (2, 96, 24, 78)), (2, (1, 64, 96, 312)), (2, (1, 32, 95, 311))):
        f1 = torch.randn(shape, generator=g); f2 = torch.randn(shape, generator=g)
        ref = nets.correlation_torch_reference(f1, f2, stride)
        got = ops.correlation(f1.cuda(), f2.cuda(), stride).cpu()
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-5


def test_monodepth2_decoder_matches_reference(ctx):
    dec = nets.fill_deterministic(nets.DepthDecoder(), int(G["md_seed"])).eval().cuda()
    rng = np.random.RandomState(int(G["md_feat_seed"]))
    shapes = [(1, 64, 32, 64), (1, 64, 16, 32), (1, 128, 8, 16), (1, 256, 4, 8), (1, 512, 2, 4)]
    feats = [torch.from_numpy(rng.uniform(0, 1.5, s).astype(np.float32)).cuda() for s in shapes]
    with torch.no_grad():
        out = dec(feats)
    for s in range(4):
        assert rel_err(out[("disp", s)].cpu().numpy(), G["md_disp%d" % s]) < TOL


def test_monodepth2_gpu_matches_cpu_run(ctx):
    net = nets.fill_deterministic(nets.MonoDepth2(), 9).eval()
    x = torch.rand(1, 3, 192, 640, generator=torch.Generator().manual_seed(1))
    ref = net(x)
    got = net.cuda()(x.cuda()).cpu()
    assert float((got - ref).abs().max()) < 2e-4


def test_area_feed_and_backwarp_equal_the_torch_forms(ctx):
    """vido_area_feed == flip + permute + float + interpolate(mode="area") [+ / 255] (bit for bit: same window rule, same summation order);
    vido_backwarp == layers.py Backward built from linspace + grid_sample (to rounding: the grid is evaluated in one expression)."""
    import torch.nn.functional as F
    from vido_slam_amd.nets import liteflownet as L
    ops = nets.HipOps(ctx)
    g = torch.Generator(device="cpu").manual_seed(3)
    for (H, W), feed, div in (((480, 640), (1088, 800), 1.0), ((480, 640), (192, 640), 255.0), ((375, 1242), (1088, 800), 1.0), ((37, 53), (16, 90), 255.0)):
        bgr = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).cuda()
        ref = F.interpolate(bgr.flip(-1).permute(2, 0, 1).float().unsqueeze(0), size=feed, mode="area")
        if div != 1.0:
            ref = ref.div(div)
        assert torch.equal(ops.area_feed(bgr, feed, div), ref), (H, W, feed)
    for B, Cc, H, W in ((1, 32, 60, 80), (2, 3, 30, 40), (1, 96, 15, 20), (1, 64, 120, 160)):
        x = torch.randn(B, Cc, H, W, generator=g).cuda()
        flow = (torch.randn(B, 2, H, W, generator=g) * 6.0).cuda()                 # many samples land outside the image
        ref = L.backwarp(x, flow); got = ops.backwarp(x, flow)
        assert float((got - ref).abs().max()) < 2e-4 * float(ref.abs().max())


def test_monodepth2_decoder_glue_kernels_and_fused_path(ctx):
    """vido_upcat_reflect / vido_bias_unary / vido_minmax_norm_u16 against their torch forms, and the decoder's fused forward (disp 0 only) against the module-by-module one
    and the reference fixture."""
    import torch.nn.functional as F
    from vido_slam_amd.nets.fuse import fold_batchnorm
    ops = nets.HipOps(ctx); g = torch.Generator().manual_seed(11)
    for C1, C2, h, w in ((16, 0, 5, 7), (32, 64, 3, 4), (256, 256, 6, 20), (1, 1, 1, 1), (16, 64, 96, 320)):
        x = torch.randn(1, C1, h, w, generator=g).cuda(); skip = torch.randn(1, C2, 2 * h, 2 * w, generator=g).cuda() if C2 else None
        up = F.interpolate(x, scale_factor=2, mode="nearest")
        ref = F.pad(torch.cat([up, skip], 1) if C2 else up, (1, 1, 1, 1), mode="reflect") if min(h, w) > 0 and 2 * min(h, w) > 1 else None
        assert torch.equal(ops.upcat_reflect(x, skip), ref), (C1, C2, h, w)
    for shape in ((1, 16, 9, 13), (1, 256, 12, 40), (2, 3, 5, 5)):
        x = (torch.randn(shape, generator=g) * 3).cuda(); b = torch.randn(shape[1], generator=g).cuda()
        assert float((ops.bias_unary_(x.clone(), b, "elu") - F.elu(x + b[None, :, None, None])).abs().max()) < 1e-6
        assert float((ops.bias_unary_(x.clone(), b, "sigmoid") - torch.sigmoid(x + b[None, :, None, None])).abs().max()) < 1e-6
    d = torch.rand(480, 640, generator=g).cuda() * 0.7 + 0.01
    lo, hi = d.min(), d.max()
    assert torch.equal(ops.minmax_norm_u16(d), ((d - lo) / (hi - lo + 1e-12) * 65536.0).clamp(0, 65535).to(torch.int32))
    dec = nets.fill_deterministic(nets.DepthDecoder(), int(G["md_seed"])).eval().cuda()
    rng = np.random.RandomState(int(G["md_feat_seed"]))
    shapes = [(1, 64, 32, 64), (1, 64, 16, 32), (1, 128, 8, 16), (1, 256, 4, 8), (1, 512, 2, 4)]
    feats = [torch.from_numpy(rng.uniform(0, 1.5, s).astype(np.float32)).cuda() for s in shapes]
    with torch.no_grad():
        plain = dec(feats)[("disp", 0)]
        fold_batchnorm(dec, ops); assert dec._ops is not None
        fused = dec(feats)
    assert list(fused.keys()) == [("disp", 0)] and float((fused[("disp", 0)] - plain).abs().max()) < 1e-6
    assert rel_err(fused[("disp", 0)].cpu().numpy(), G["md_disp0"]) < TOL
