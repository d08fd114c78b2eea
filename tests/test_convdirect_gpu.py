"""csrc/convdirect.hip on the MI355X: LiteFlowNet's stem, stride-2, separable and few-channel convolutions (flow_net/src/layers.py:39-73, 217-235) as one direct implicit-GEMM
launch on the fp32 matrix pipe, against conv2d in float64.  Tolerance: fp32 products and sums in another order than the library's — 1e-5 of the output scale."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def ctx(vido):
    c = vido.Context()
    yield c
    c.close()


@pytest.fixture(autouse=True)
def every_layer(monkeypatch):
    """the kernel on every layer it has an instance for (the callers' default leaves the 7x7 stem and the stride-2 3x3 layers to the library: nets/ops.py _CONVDIRECT_SET)"""
    monkeypatch.setattr("vido_slam_amd.nets.ops._CONVDIRECT_SET", "all")


SHAPES = [  # (N, cin, cout, H, W, (kh, kw), (sh, sw)): the layers of LiteFlowNet at 480 x 640 scaled down where the map is large, ragged sizes, odd channel counts, batches
    (2, 3, 32, 40, 56, (7, 7), (1, 1)), (2, 32, 32, 48, 64, (3, 3), (2, 2)), (1, 32, 64, 30, 40, (3, 3), (2, 2)), (2, 64, 96, 30, 40, (3, 3), (2, 2)), (1, 96, 128, 15, 20, (3, 3), (2, 2)),
    (2, 128, 192, 16, 20, (3, 3), (2, 2)), (1, 32, 49, 24, 32, (7, 1), (1, 1)), (1, 49, 49, 24, 32, (1, 7), (1, 1)), (1, 32, 25, 30, 40, (5, 1), (1, 1)), (1, 25, 25, 30, 40, (1, 5), (1, 1)),
    (1, 32, 9, 30, 40, (3, 3), (1, 1)), (1, 32, 9, 15, 20, (3, 3), (1, 1)), (1, 49, 1, 24, 32, (1, 1), (1, 1)), (1, 9, 1, 15, 20, (1, 1), (1, 1)), (1, 5, 33, 7, 9, (5, 5), (1, 1)),
    (3, 7, 70, 9, 11, (3, 3), (1, 1)), (1, 1, 1, 1, 1, (1, 1), (1, 1)), (1, 3, 32, 13, 17, (7, 7), (2, 2)), (1, 16, 32, 11, 13, (3, 3), (3, 2)), (1, 64, 64, 33, 47, (3, 3), (2, 2)),
    (2, 3, 32, 480, 640, (7, 7), (1, 1)), (2, 32, 32, 480, 640, (3, 3), (2, 2)), (1, 49, 49, 240, 320, (1, 7), (1, 1)),      # three layers at the benchmark's size
]


@pytest.mark.parametrize("N,cin,cout,H,W,k,s", SHAPES)
def test_conv_direct_equals_conv2d(vido, ctx, N, cin, cout, H, W, k, s):
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    g = torch.Generator().manual_seed(cin * 7 + cout + H + k[0])
    conv = torch.nn.Conv2d(cin, cout, k, s, (k[0] // 2, k[1] // 2))
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (cin * k[0] * k[1]) ** 0.5); conv.bias.copy_(torch.randn(cout, generator=g))
    x = torch.randn(N, cin, H, W, generator=g)
    ref0 = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), s, (k[0] // 2, k[1] // 2))
    conv = conv.cuda()
    with torch.no_grad():
        for slope in (1.0, 0.1):
            y = ops.conv_direct_conv(conv, x.cuda(), slope)
            assert y is not None, "no kernel"
            ref = F.leaky_relu(ref0, slope) if slope != 1.0 else ref0
            err = float((y.cpu().double() - ref).abs().max())
            assert tuple(y.shape) == tuple(ref.shape) and err < TOL * max(1.0, float(ref.abs().max())), (N, cin, cout, H, W, k, s, slope, err)
        nb = torch.nn.Conv2d(cin, cout, k, s, (k[0] // 2, k[1] // 2), bias=False).cuda()
        nb.weight.copy_(conv.weight)
        y = ops.conv_direct_conv(nb, x.cuda(), 1.0)
        err = float((y.cpu().double() - (ref0 - conv.bias.cpu().double()[None, :, None, None])).abs().max())
        assert err < TOL * max(1.0, float(ref0.abs().max())), ("no bias", err)


def test_conv_direct_refuses_what_it_has_no_kernel_for(vido, ctx):
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    x = torch.zeros(1, 8, 8, 8, device="cuda")
    assert ops.conv_direct_conv(torch.nn.Conv2d(8, 8, 3, 1, 1, groups=2).cuda(), x, 1.0) is None            # grouped
    assert ops.conv_direct_conv(torch.nn.Conv2d(8, 8, 3, 1, 2, dilation=2).cuda(), x, 1.0) is None          # dilated
    assert ops.conv_direct_conv(torch.nn.Conv2d(8, 8, (3, 5), 1, (1, 2)).cuda(), x, 1.0) is None            # a tap shape without an instance
    assert ops.conv_direct_conv(torch.nn.Conv2d(8, 8, 3, 1, 1, padding_mode="reflect").cuda(), x, 1.0) is None
    lib = ctx.lib
    assert lib.vido_conv_direct_supported(8, 8, 8, 8, 3, 3, 1, 1, 1, 1) == 1 and lib.vido_conv_direct_supported(8, 8, 8, 8, 3, 3, 5, 1, 1, 1) == 0 and lib.vido_conv_direct_supported(8, 8, 2, 2, 7, 7, 1, 1, 0, 0) == 0
    with pytest.raises(vido.VidoError):
        import ctypes as C
        ctx._check(lib.vido_conv_direct_bias_act(ctx.h, None, None, None, None, 1, 8, 8, 8, 8, 3, 3, 1, 1, 1, 1, C.c_float(1.0)))


def test_conv_direct_follows_a_weight_update(vido, ctx):
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    conv = torch.nn.Conv2d(16, 32, 3, 2, 1).cuda(); x = torch.randn(1, 16, 12, 10, device="cuda")
    with torch.no_grad():
        y0 = ops.conv_direct_conv(conv, x, 0.1)
        assert float((y0 - F.leaky_relu(conv(x), 0.1)).abs().max()) < 1e-4
        conv.weight.mul_(-0.5); conv.bias.add_(1.0)
        y1 = ops.conv_direct_conv(conv, x, 0.1)
        assert float((y1 - F.leaky_relu(conv(x), 0.1)).abs().max()) < 1e-4 and float((y1 - y0).abs().max()) > 1e-2


def test_default_set_leaves_the_vector_alu_winograd_layers_to_the_library(vido, ctx, monkeypatch):
    """beside the detector the stem and the stride-2 3x3 layers run faster on the library's vector-ALU Winograd kernels (profiles/r5/convdirect_ab.txt): the default set"""
    from vido_slam_amd.nets.ops import HipOps
    monkeypatch.setattr("vido_slam_amd.nets.ops._CONVDIRECT_SET", "novalu")
    ops = HipOps(ctx)
    assert ops.conv_direct_conv(torch.nn.Conv2d(3, 32, 7, 1, 3).cuda(), torch.zeros(1, 3, 16, 16, device="cuda"), 0.1) is None
    assert ops.conv_direct_conv(torch.nn.Conv2d(32, 64, 3, 2, 1).cuda(), torch.zeros(1, 32, 16, 16, device="cuda"), 0.1) is None
    assert ops.conv_direct_conv(torch.nn.Conv2d(32, 49, (7, 1), 1, (3, 0)).cuda(), torch.zeros(1, 32, 16, 16, device="cuda"), 1.0) is not None
    assert ops.conv_direct_conv(torch.nn.Conv2d(32, 9, 3, 1, 1).cuda(), torch.zeros(1, 32, 16, 16, device="cuda"), 1.0) is not None
