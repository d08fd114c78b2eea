"""csrc/wino.hip on the MI355X: the dense 3x3 convolutions of the three network nodes as Winograd F(2x2, 3x3) with the channel contractions on the fp32 matrix pipe, against
conv2d in float64 (the arithmetic the reference's layers define: flow_net/src/layers.py:39-315, maskrcnn_benchmark/modeling/backbone/fpn.py, rpn/rpn.py:74-107,
roi_heads/mask_head/roi_mask_feature_extractors.py).  Tolerance: fp32 Winograd differs from an exactly rounded fp32 convolution by rounding only — 1e-4 of the output scale
(the library's own Winograd kernels, which these launches replace, sit in the same class)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from vido_slam_amd import nets

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def ctx(vido):
    c = vido.Context()
    yield c
    c.close()


SHAPES = [  # (N, cin, cout, H, W): LiteFlowNet heads at the small levels, ragged channel counts, odd maps, the 32-channel form (4-channel chunks), batches, the minimum
    (1, 49, 128, 30, 40), (1, 130, 128, 60, 80), (1, 131, 128, 15, 20), (1, 386, 128, 15, 20), (2, 32, 32, 24, 32), (1, 64, 32, 30, 40), (1, 64, 96, 17, 23),
    (1, 96, 96, 60, 80), (3, 256, 256, 14, 14), (1, 256, 256, 25, 34), (1, 8, 64, 2, 2), (1, 9, 32, 3, 5), (2, 16, 64, 7, 9), (1, 128, 64, 120, 160),
    (1, 256, 256, 200, 272), (1, 128, 128, 240, 320),      # the two shapes the bench prices: the FPN output convolution of P2 (852 workgroups), LiteFlowNet's level-2 regularisation
]


@pytest.mark.parametrize("form", [0, 1, 2])
@pytest.mark.parametrize("N,cin,cout,H,W", SHAPES)
def test_wino3x3_equals_conv2d(vido, ctx, N, cin, cout, H, W, form):
    """both launch forms on every shape: the tile form (a wave walks all input channels) and the K-split form (the four waves of a workgroup share them; what under-filled
    launches take — vido_wino3x3_form)"""
    from vido_slam_amd.nets.ops import HipOps, pack_wino3x3
    ops = HipOps(ctx)
    if form and N * cin * H * W > 40e6:
        pytest.skip("the K-split form is for small launches")
    g = torch.Generator().manual_seed(cin * 13 + cout + H)
    x = torch.randn(N, cin, H, W, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (3.0 * cin ** 0.5)); b = torch.randn(cout, generator=g)
    assert ops.wino3x3_supported(cin, cout, H, W)
    up = pack_wino3x3(w, form).cuda()
    ref0 = F.conv2d(x.double(), w.double(), None, 1, 1)
    for bias, slope in ((None, 1.0), (b, 0.1), (b, 0.0)):
        y = ops.wino3x3_bias_act(x.cuda(), up, bias.cuda() if bias is not None else None, cout, slope, form).cpu()
        ref = F.leaky_relu(ref0 + (bias.double()[None, :, None, None] if bias is not None else 0.0), slope)
        err = float((y.double() - ref).abs().max())
        assert tuple(y.shape) == (N, cout, H, W) and err < TOL * max(1.0, float(ref.abs().max())), (N, cin, cout, H, W, slope, form, err)


def test_wino3x3_refuses_what_it_has_no_form_for(vido, ctx):
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    assert not ops.wino3x3_supported(4, 64, 8, 8) and not ops.wino3x3_supported(64, 2, 8, 8) and not ops.wino3x3_supported(64, 64, 1, 8)
    with pytest.raises(vido.VidoError):
        ops.wino3x3_bias_act(torch.zeros(1, 64, 8, 8, device="cuda"), torch.zeros(16, device="cuda"), None, 2, 1.0)
    conv = torch.nn.Conv2d(64, 64, 3, 2, 1).cuda()                         # stride 2: not this kernel's; the caller keeps the library path
    assert ops.wino3x3_conv(conv, torch.zeros(1, 64, 8, 8, device="cuda"), 1.0) is None
    small = torch.nn.Conv2d(64, 64, 3, 1, 1).cuda()                        # 16 tiles: one workgroup of the tile form -> the K-split form
    assert ops.wino3x3_form(1, 64, 64, 8, 8) == 1 and ops.wino3x3_form(1, 256, 256, 200, 272) == 0 and ops.wino3x3_form(1, 256, 256, 50, 68) == 1 and ops.wino3x3_form(1, 128, 64, 120, 160) == 2
    with torch.no_grad():
        xs = torch.randn(1, 64, 8, 8, device="cuda"); ys = ops.wino3x3_conv(small, xs, 1.0)
        assert ys is not None and float((ys - small(xs)).abs().max()) < 1e-4
    assert ctx.lib.vido_wino3x3_fills_chip(1, 256, 200, 272, 0) == 1 and ctx.lib.vido_wino3x3_fills_chip(1, 256, 50, 68, 0) == 0 and ctx.lib.vido_wino3x3_fills_chip(1, 256, 50, 68, 40) == 1


def test_wino3x3_conv_follows_a_weight_update(vido, ctx, monkeypatch):
    """wino3x3_conv caches the packed transformed weight on the module: loading other weights (load_state_dict copies in place) must rebuild it."""
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    conv = torch.nn.Conv2d(16, 32, 3, 1, 1).cuda(); x = torch.randn(1, 16, 12, 10, device="cuda")
    with torch.no_grad():
        y0 = ops.wino3x3_conv(conv, x, 0.1)
        assert float((y0 - F.leaky_relu(conv(x), 0.1)).abs().max()) < 1e-4
        conv.weight.mul_(-0.5); conv.bias.add_(1.0)
        y1 = ops.wino3x3_conv(conv, x, 0.1)
        assert float((y1 - F.leaky_relu(conv(x), 0.1)).abs().max()) < 1e-4 and float((y1 - y0).abs().max()) > 1e-2


def test_liteflownet_with_and_without_the_winograd_launches(vido, ctx, monkeypatch):
    """The whole flow network with its dense 3x3 convolutions on csrc/wino.hip against the same network on the library convolutions + the bias / LeakyReLU pass."""
    ops = nets.HipOps(ctx)
    monkeypatch.setattr("vido_slam_amd.nets.ops._WINO_MIN_WGS", 1)          # take every layer that has the form, also the ones a 128 x 160 feed leaves too small to fill the chip
    torch.manual_seed(3)
    a = torch.rand(1, 3, 128, 160, device="cuda"); b = torch.rand(1, 3, 128, 160, device="cuda")
    net = nets.LiteFlowNet(ops.correlation, epilogue=ops.bias_act_, warp=ops.backwarp, fused=ops).eval().cuda()
    nets.fill_deterministic(net, 21)
    with torch.no_grad():
        y1 = net(a, b)
        taken = sum(1 for m in net.modules() if getattr(m, "_wino_u", None) is not None)
        monkeypatch.setenv("VIDO_NO_WINO", "1")
        ref = nets.LiteFlowNet(ops.correlation, epilogue=ops.bias_act_, warp=ops.backwarp, fused=ops).eval().cuda()
        ref.load_state_dict(net.state_dict())
        y0 = ref(a, b)
    assert taken >= 40 and sum(1 for m in ref.modules() if getattr(m, "_wino_u", None) is not None) == 0
    assert float((y1 - y0).abs().max()) < 2e-3 * max(1.0, float(y0.abs().max()))


@pytest.mark.parametrize("k,cin,H,W", [(7, 32, 60, 80), (5, 32, 33, 47), (3, 32, 15, 20), (7, 19, 17, 16), (5, 8, 1, 3)])
def test_conv_kxk_to_two_channels_equals_conv2d(vido, ctx, k, cin, H, W):
    """csrc/convsmall.hip (the last layer of LiteFlowNet's flow heads: k x k, 32 -> 2, + bias + the flow it refines) against conv2d in float64."""
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    torch.manual_seed(k * 100 + cin)
    conv = torch.nn.Conv2d(cin, 2, k, 1, k // 2).cuda(); x = torch.randn(1, cin, H, W, device="cuda"); r = torch.randn(1, 2, H, W, device="cuda")
    with torch.no_grad():
        ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), 1, k // 2)
        y0 = ops.conv_kxk_c2(conv, x); y1 = ops.conv_kxk_c2(conv, x, r)
    assert float((y0.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    assert float((y1.double() - (ref + r.double())).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    assert ops.conv_kxk_c2(torch.nn.Conv2d(cin, 3, k, 1, k // 2).cuda(), x) is None and ops.conv_kxk_c2(torch.nn.Conv2d(cin, 2, k, 2, k // 2).cuda(), x) is None


@pytest.mark.parametrize("cin,cout,H,W", [(32, 64, 60, 80), (32, 128, 33, 47), (64, 128, 15, 20), (96, 128, 30, 40), (6, 40, 5, 7), (128, 32, 9, 9)])
def test_conv1x1_skinny_equals_conv2d(vido, ctx, cin, cout, H, W):
    """csrc/convsmall.hip::k_conv1x1_skinny (LiteFlowNet's netFeat layers: 1x1, few input channels, bias + LeakyReLU) against conv2d in float64."""
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    torch.manual_seed(cin * 7 + cout)
    conv = torch.nn.Conv2d(cin, cout, 1).cuda(); x = torch.randn(1, cin, H, W, device="cuda")
    with torch.no_grad():
        for slope in (0.1, 1.0, 0.0):
            y = ops.conv1x1_skinny_conv(conv, x, slope)
            ref = F.leaky_relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double()), slope)
            assert float((y.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())), (cin, cout, slope)
    assert ops.conv1x1_skinny_conv(torch.nn.Conv2d(cin + 1, cout, 1).cuda(), torch.randn(1, cin + 1, H, W, device="cuda")) is None      # odd channel count: the library's
