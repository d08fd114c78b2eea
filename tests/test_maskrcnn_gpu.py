"""Row N3 on the MI355X: the Mask R-CNN inference graph with the HIP ROI-Align / NMS / box-decode kernels (C-ABI, torch's
stream) against the stage-wise outputs of the reference detector."""
import os
import numpy as np
import pytest
import torch
from vido_slam_amd import nets

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "maskrcnn_graph.npz"))
TINY = nets.MaskRCNNConfig(blocks=(3, 4, 6, 3), groups=4, width_per_group=4, res2_out=32, stem_out=16, fpn_out=16, mlp_dim=64, num_classes=7,
                           mask_layers=(16, 16, 16, 16), detections_per_img=20)
TOL = 5e-4            # relative to the tensor's max magnitude: fp32 MIOpen convolutions vs the reference's CPU run


@pytest.fixture(scope="module")
def ctx(vido):
    c = vido.Context()
    yield c
    c.close()


def rel_err(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-12))


def test_tiny_graph_stage_by_stage(ctx):
    ops = nets.HipOps(ctx)
    net = nets.fill_maskrcnn(nets.MaskRCNN(ops, TINY), int(G["seed"])).eval().cuda()
    img = torch.from_numpy(G["image"])[None].cuda()
    with torch.no_grad():
        feats = net.backbone(img)
        for i, f in enumerate(feats):
            assert rel_err(f.cpu().numpy(), G["feat%d" % i]) < TOL, i
        # every later stage is fed the reference's inputs, so a rounding-level difference upstream cannot flip a discrete choice
        gfeats = [torch.from_numpy(G["feat%d" % i]).cuda() for i in range(5)]
        prop, obj = net.rpn(gfeats, (img.shape[-1], img.shape[-2]))
        nv = int((obj >= 0).sum())                                   # device-side RPN path: fixed-size lists, padding rows have objectness -1
        assert nv == len(G["proposals"]) and bool((obj[nv:] < 0).all()) and bool((prop[nv:] == 0).all())
        prop, obj = prop[:nv], obj[:nv]
        assert rel_err(prop.cpu().numpy(), G["proposals"]) < TOL and rel_err(obj.cpu().numpy(), G["objectness"]) < TOL
        gprop = torch.from_numpy(G["proposals"]).cuda()
        boxes, scores, labels = net.roi_heads.box(gfeats[:4], gprop, (img.shape[-1], img.shape[-2]))
        assert np.array_equal(labels.cpu().numpy(), G["det_labels"])
        assert rel_err(boxes.cpu().numpy(), G["det_boxes"]) < TOL and rel_err(scores.cpu().numpy(), G["det_scores"]) < TOL
        masks = net.roi_heads.mask(gfeats[:4], torch.from_numpy(G["det_boxes"]).cuda(), torch.from_numpy(G["det_labels"]).cuda())
        assert rel_err(masks.cpu().numpy(), G["det_masks"]) < TOL
    OW, OH = [int(v) for v in G["paste_size"]]
    pasted = nets.paste_masks(torch.from_numpy(G["det_masks"]).cuda(), torch.from_numpy(G["resized_boxes"]).cuda(), OH, OW)
    ref = np.unpackbits(G["pasted"], axis=-1)[..., :OW].astype(bool)
    assert (pasted.cpu().numpy() != ref).mean() < 1e-4           # bilinear resize at the 0.5 threshold: allow isolated pixels


def test_tiny_graph_end_to_end(ctx):
    net = nets.fill_maskrcnn(nets.MaskRCNN(nets.HipOps(ctx), TINY), int(G["seed"])).eval().cuda()
    out = net(torch.from_numpy(G["image"])[None].cuda())
    nv = int(out["n_proposals"])
    assert nv == len(G["proposals"]) and rel_err(out["proposals"][:nv].cpu().numpy(), G["proposals"]) < 5e-3
    assert np.array_equal(out["labels"].cpu().numpy(), G["det_labels"])
    assert rel_err(out["masks"].cpu().numpy(), G["det_masks"]) < 5e-3


def test_full_size_graph_runs(ctx):
    """X-101-32x8d-FPN at the node's 800x1088 feed (random-init weights): shapes, dtypes, finiteness."""
    net = nets.fill_maskrcnn(nets.MaskRCNN(nets.HipOps(ctx)), 1).eval().cuda()
    bgr = (np.random.RandomState(0).rand(375, 1242, 3) * 255).astype(np.uint8)
    img, labels = nets.analyse_image(net, bgr)
    assert tuple(img.shape) == (375, 1242) and img.dtype == torch.uint8
    out = net(torch.rand(1, 3, 1088, 800, device="cuda"))
    assert out["proposals"].shape[1] == 4 and 0 < int(out["n_proposals"]) <= 1000 and bool(torch.isfinite(out["proposals"]).all())
    assert out["masks"].shape[1:] == (1, 28, 28)


def test_device_side_head_ops_match_the_host_forms(ctx, oracle):
    """vido_nms_segments == layers.nms per segment (oracle), vido_roi_align_fpn == one vido_roi_align per level scattered back,
    vido_mask_label_image == Masker + the node's accumulation loop (torch paste_masks)."""
    ops = nets.HipOps(ctx)
    rng = np.random.RandomState(5)
    # segmented NMS, ragged segments incl. one longer than 64 x 16 boxes and an empty one
    ns = [1000, 663, 0, 70, 1300]
    K = 1344
    boxes = np.zeros((len(ns) * K, 4), np.float32); ref = []
    for g, n in enumerate(ns):
        # clustered boxes (40 centres, small jitter): heavy mutual suppression inside and across the 64-box blocks of the sweep, like RPN proposals
        cen = rng.uniform(20, 300, (40, 2)); c = cen[rng.randint(0, 40, n)] + rng.normal(0, 4, (n, 2)); wh = rng.uniform(30, 60, (n, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        boxes[g * K:g * K + n] = b
        ref.append(np.sort(oracle.nms(b, np.arange(n, 0, -1).astype(np.float32), 0.7)) if n else np.zeros(0, np.int64))
    keep, cnt = ops.nms_segments(torch.from_numpy(boxes).cuda(), (torch.arange(len(ns), dtype=torch.int32) * K).cuda(), torch.tensor(ns, dtype=torch.int32).cuda(), K, 0.7)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for g, n in enumerate(ns):
        assert cnt[g] == len(ref[g]) and np.array_equal(keep[g, :cnt[g]], ref[g]) and (keep[g, cnt[g]:] == -1).all(), g
    # FPN ROI-Align
    feats = [torch.randn(1, 16, 64 >> l, 80 >> l, device="cuda") for l in range(4)]
    scales = (0.25, 0.125, 0.0625, 0.03125)
    n = 300
    xy = rng.uniform(0, 200, (n, 2)); wh = rng.uniform(4, 120, (n, 2))
    b = torch.from_numpy(np.concatenate([xy, xy + wh], 1).astype(np.float32)).cuda()
    lvl = torch.from_numpy(rng.randint(0, 4, n).astype(np.int32)).cuda()
    got = ops.roi_align_fpn(feats, b, lvl, (7, 7), scales, 2)
    rois = torch.cat([b.new_zeros((n, 1)), b], 1)
    for l in range(4):
        idx = torch.nonzero(lvl == l).squeeze(1)
        assert torch.equal(got[idx], ops.roi_align(feats[l], rois[idx], (7, 7), scales[l], 2)), l
    # the channels-last form (what the detector calls: one transpose per level and frame) against the oracle, C = 256 and a C that is not a multiple of the 64-channel
    # groups; sampling_ratio 2 (the node's) and 0 (adaptive grid); boxes partly outside the maps
    for Cc, res, sr in ((256, 7, 2), (256, 14, 2), (100, 7, 0)):
        feats = [torch.randn(1, Cc, 64 >> l, 80 >> l, device="cuda") for l in range(4)]
        nh = [ops.to_nhwc(f) for f in feats]
        for f, t in zip(feats, nh):
            assert torch.equal(t, f.permute(0, 2, 3, 1).contiguous())
        xy = rng.uniform(-30, 300, (n, 2)); wh = rng.uniform(1, 160, (n, 2))
        b = torch.from_numpy(np.concatenate([xy, xy + wh], 1).astype(np.float32)).cuda()
        got = ops.roi_align_fpn_nhwc(nh, b, lvl, (res, res), scales, sr).cpu().numpy()
        assert np.array_equal(got, ops.roi_align_fpn(feats, b, lvl, (res, res), scales, sr).cpu().numpy())
        rois = np.concatenate([np.zeros((n, 1), np.float32), b.cpu().numpy()], 1)
        lv = lvl.cpu().numpy()
        for l in range(4):
            idx = np.nonzero(lv == l)[0]
            ref = oracle.roi_align(feats[l].cpu().numpy(), rois[idx], scales[l], res, res, sr)
            assert np.array_equal(got[idx], ref), (Cc, res, sr, l)
    # Masker + label image
    nd, Hh, Ww = 37, 120, 200
    masks = torch.rand(nd, 1, 28, 28, device="cuda")
    xy = rng.uniform(-20, 150, (nd, 2)); wh = rng.uniform(3, 90, (nd, 2))
    bx = torch.from_numpy(np.concatenate([xy, xy + wh], 1).astype(np.float32)).cuda()
    labels = torch.from_numpy(rng.randint(1, 81, nd).astype(np.int64)).cuda()
    img = ops.mask_label_image(masks, bx, labels, Hh, Ww)
    pasted = nets.paste_masks(masks, bx, Hh, Ww)
    ref_img = (pasted.to(torch.int32) * labels.view(-1, 1, 1).to(torch.int32)).sum(0).remainder(256).to(torch.uint8)
    assert (img != ref_img).float().mean().item() < 1e-4           # bilinear resize at the 0.5 threshold: isolated pixels may fall on the other side
    assert torch.equal(ops.mask_label_image(masks[:0], bx[:0], labels[:0], Hh, Ww), torch.zeros((Hh, Ww), dtype=torch.uint8, device="cuda"))


def test_fused_selection_kernels_equal_the_torch_forms(ctx):
    """csrc/detpost.hip (vido_rpn_select / vido_rpn_merge / vido_det_class_sort / vido_det_select) against the torch-op forms they replace (proposals_device /
    postprocess_static, themselves checked against the reference's data-dependent flow), on inputs full of TIES: quantised objectness logits (thousands of equal sigmoids,
    many saturated to exactly 1.0, a tie group straddling the top-k cut), a level with fewer anchors than pre_nms_top_n, duplicate class-score rows, and more than
    detections_per_img equal top scores (n_det > cap: the overflow the caller must detect)."""
    ops = nets.HipOps(ctx)
    net = nets.fill_maskrcnn(nets.MaskRCNN(ops, nets.MaskRCNNConfig(num_classes=9, detections_per_img=30)), 7).eval().cuda()
    g = torch.Generator().manual_seed(3)
    sizes = [(100, 136), (50, 68), (25, 34), (13, 17), (7, 9)]                      # 40 800 ... 189 anchors: the last two levels have fewer than 1000
    for trial, quant in enumerate((0.25, 2.0, 0.0)):
        logits = [torch.randn((1, 3, h, w), generator=g) * (6.0 if trial == 1 else 3.0) for h, w in sizes]
        if quant:
            logits = [torch.round(t / quant) * quant for t in logits]
        logits = [t.cuda() for t in logits]
        deltas = [(torch.randn((1, 12, h, w), generator=g) * 0.5).cuda() for h, w in sizes]
        feats = [torch.zeros((1, 1, h, w), device="cuda") for h, w in sizes]
        with torch.no_grad():
            a_b, a_s = net.rpn.proposals_device(feats, logits, deltas, (800, 1088))
            f_b, f_s = net.rpn.proposals_fused(feats, logits, deltas, (800, 1088))
        assert torch.equal(a_s, f_s), trial
        assert torch.equal(a_b, f_b), trial
    N, nc = 1000, 9
    for trial in range(3):
        lg = torch.randn((N, nc), generator=g) * 2.0
        if trial >= 1:
            lg = torch.round(lg)                                                    # equal logit rows -> equal probabilities: ties inside the classes
        if trial == 2:
            lg[:60] = torch.tensor([0.0, 9.0, 0, 0, 0, 0, 0, 0, 0])                  # 60 proposals with the same (top) score of class 1 ...
        dl = torch.randn((N, nc * 4), generator=g) * 0.3
        xy = torch.rand((N, 2), generator=g) * 600; wh = torch.rand((N, 2), generator=g) * 200 + 8
        if trial == 2:
            xy[:60] = torch.arange(60).float().view(-1, 1) * 13.0 % 700; wh[:60] = 10.0; dl[:60] = 0.0          # ... on small disjoint boxes: none of them suppresses another
        pr = torch.cat([xy, xy + wh], 1)
        obj = torch.rand((N,), generator=g); obj[900:] = -1.0; pr[900:] = 0.0      # padding rows of the device-side RPN path
        lg, dl, pr, obj = lg.cuda(), dl.cuda(), pr.cuda(), obj.cuda()
        with torch.no_grad():
            a = net.roi_heads.box.postprocess_static(lg, dl, pr, (800, 1088), obj, 30)
            f = net.roi_heads.box.postprocess_fused(lg, dl, pr, (800, 1088), obj, 30)
        assert int(a[3]) == int(f[3]) and int(a[3]) > 0, trial
        if trial == 2:
            assert int(f[3]) > 30                                                  # ties at the detections_per_img cut: the reference keeps them all
        for k in range(3):
            assert torch.equal(a[k], f[k]), (trial, k)


@pytest.mark.gpu
@pytest.mark.parametrize("cpg,H,W", [(32, 50, 68), (64, 25, 34), (16, 100, 136), (8, 200, 272), (32, 7, 9), (16, 9, 300), (8, 3, 5), (32, 1, 1), (16, 33, 61), (8, 6, 302), (16, 2, 4)])
def test_grouped_conv_matrix_core_kernel_equals_conv2d(vido, ctx, cpg, H, W):
    """csrc/gconv.hip against conv2d in float64: the detector's four bottleneck shapes (8 / 16 / 32 / 64 channels per group at their FPN-stage sizes) and ragged small ones
    (tiles that run across row ends, bands cut by the image border, a single pixel)."""
    from vido_slam_amd.nets.ops import HipOps, pack_gconv3x3
    ops = HipOps(ctx); groups = 32 if H * W > 2000 else 3
    g = torch.Generator().manual_seed(cpg * 1000 + H)
    x = torch.randn(1, groups * cpg, H, W, generator=g); w = torch.randn(groups * cpg, cpg, 3, 3, generator=g) * (1.0 / (3 * cpg ** 0.5)); b = torch.randn(groups * cpg, generator=g)
    assert ops.gconv3x3_supported(H, W, cpg, cpg)
    wp = pack_gconv3x3(w, groups).cuda()
    for slope in (0.0, 1.0):
        y = ops.gconv3x3_bias_act(x.cuda(), wp, b.cuda(), groups, slope).cpu()
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 1, 1, groups), slope)
        err = float((y.double() - ref).abs().max())
        assert err < 2e-5 * max(1.0, float(ref.abs().max())), (cpg, H, W, slope, err)
    ib = torch.randn(groups * cpg, generator=g)             # with the producer's bias + ReLU folded into the operand reads
    y = ops.gconv3x3_bias_act(x.cuda(), wp, b.cuda(), groups, 0.0, in_bias=ib.cuda()).cpu()
    ref = torch.relu(torch.nn.functional.conv2d(torch.relu(x.double() + ib.double()[None, :, None, None]), w.double(), b.double(), 1, 1, 1, groups))
    assert float((y.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), (cpg, H, W, "in_bias")
    # shapes outside the plan are refused, not mangled
    assert not ops.gconv3x3_supported(50, 100, 32, 32) and not ops.gconv3x3_supported(50, 68, 12, 12)
    with pytest.raises(vido.VidoError):
        ops.gconv3x3_bias_act(torch.zeros(1, 64, 50, 100, device="cuda"), torch.zeros(64 * 32 * 9, device="cuda"), torch.zeros(64, device="cuda"), 2, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("cpg,H,W", [(32, 100, 136), (64, 50, 68), (32, 11, 12), (32, 2, 4), (64, 7, 8), (32, 40, 200), (16, 200, 272), (16, 9, 12), (8, 31, 40), (16, 2, 4)])
def test_strided_grouped_conv_matrix_core_kernel_equals_conv2d(vido, ctx, cpg, H, W):
    """csrc/gconv.hip::k_gconv3x3_s2_m32 against conv2d(stride 2, padding 1) in float64: the first bottleneck of layer3 / layer4 at its FPN-stage size, odd heights, bands cut
    by the image border, a 2 x 4 map, a row wider than a tile."""
    from vido_slam_amd.nets.ops import HipOps, pack_gconv3x3
    ops = HipOps(ctx); groups = 32 if H * W > 2000 else 3
    g = torch.Generator().manual_seed(cpg * 1000 + H)
    x = torch.randn(1, groups * cpg, H, W, generator=g); w = torch.randn(groups * cpg, cpg, 3, 3, generator=g) * (1.0 / (3 * cpg ** 0.5)); b = torch.randn(groups * cpg, generator=g)
    assert ops.gconv3x3_s2_supported(H, W, cpg, cpg)
    wp = pack_gconv3x3(w, groups).cuda()
    for slope in (0.0, 1.0):
        y = ops.gconv3x3_s2_bias_act(x.cuda(), wp, b.cuda(), groups, slope).cpu()
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 2, 1, 1, groups), slope)
        assert y.shape == ref.shape, (y.shape, ref.shape)
        err = float((y.double() - ref).abs().max())
        assert err < 2e-5 * max(1.0, float(ref.abs().max())), (cpg, H, W, slope, err)
    assert not ops.gconv3x3_s2_supported(50, 70, 32, 32) and not ops.gconv3x3_s2_supported(50, 68, 24, 24) and not ops.gconv3x3_s2_supported(40, 300, 32, 32)      # W % 4, 24 channels per group, a band that does not fit two LDS buffers: the library keeps them
    with pytest.raises(vido.VidoError):
        ops.gconv3x3_s2_bias_act(torch.zeros(1, 64, 50, 70, device="cuda"), torch.zeros(64 * 32 * 9, device="cuda"), torch.zeros(64, device="cuda"), 2, 0.0)


@pytest.mark.gpu
def test_bottleneck_with_matrix_core_conv2_equals_library_path(vido, ctx):
    """_Bottleneck.forward with conv2 on csrc/gconv.hip against the same block on the library convolution + bias pass."""
    from vido_slam_amd.nets import maskrcnn as M
    from vido_slam_amd.nets.fuse import fold_batchnorm
    from vido_slam_amd.nets.ops import HipOps
    from vido_slam_amd.nets.weights import fill_maskrcnn
    blk = M._Bottleneck(256, 256, 256, 32, False, 1)
    fill_maskrcnn(blk); blk = blk.cuda().eval()
    x = torch.randn(1, 256, 40, 52, generator=torch.Generator().manual_seed(4)).cuda()
    with torch.no_grad():
        assert fold_batchnorm(blk, HipOps(ctx)) == 3 and blk._w2p is not None
        y_fast = blk(x).clone()
        blk._w2p = None
        y_lib = blk(x)
    assert float((y_fast - y_lib).abs().max()) < 1e-4 * max(1.0, float(y_lib.abs().max()))


_FORCED_C1 = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
import vido_slam_amd as V
from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
cin, cout, H, W, layout = (int(a) for a in sys.argv[2:7])
ctx = V.Context(width=640, height=480, max_batch=1); ops = HipOps(ctx)
assert ops.conv1x1_layout(cin, cout, H * W) == layout
g = torch.Generator().manual_seed(cin * 7 + H)
x = torch.randn(1, cin, H, W, generator=g); w = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin ** 0.5); b = torch.randn(cout, generator=g); r = torch.randn(1, cout, H, W, generator=g)
wp = pack_conv1x1(w, layout).cuda()
ref0 = torch.nn.functional.conv2d(x.double(), w.double())
for bias, res, slope in ((None, None, 1.0), (b, r, 0.0), (b, r, 0.1)):
    y = ops.conv1x1_bias_act(x.cuda(), wp, bias.cuda() if bias is not None else None, res.cuda() if res is not None else None, slope).cpu()
    ref = ref0 + (bias.double()[None, :, None, None] if bias is not None else 0.0) + (res.double() if res is not None else 0.0)
    ref = torch.nn.functional.leaky_relu(ref, slope)
    err = float((y.double() - ref).abs().max())
    assert err < 2e-5 * max(1.0, float(ref.abs().max())), (cin, cout, H, W, slope, err)
if H % 2 == 0 and W % 2 == 0:
    rh = torch.randn(1, cout, H // 2, W // 2, generator=g)
    y = ops.conv1x1_bias_act(x.cuda(), wp, b.cuda(), None, 1.0, residual_up2=rh.cuda()).cpu()
    ref = ref0 + b.double()[None, :, None, None] + torch.nn.functional.interpolate(rh.double(), scale_factor=2, mode="nearest")
    assert float((y.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
print("forced ok")
"""


def _run_forced_conv1x1(cin, cout, H, W, layout):
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VIDO_CONV1X1_TN="112" if layout == 1 else "128")
    out = subprocess.run([sys.executable, "-c", _FORCED_C1, root, str(cin), str(cout), str(H), str(W), str(layout)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "forced ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [None, 0, 1])      # None: the tile form the library picks for the shape; 0 / 1: forced 128 x 128 / 128 x 112 tiles (VIDO_CONV1X1_TN)
@pytest.mark.parametrize("cin,cout,H,W", [(64, 256, 40, 68), (256, 256, 50, 68), (256, 128, 37, 52), (1024, 1024, 10, 34), (512, 512, 25, 36), (32, 128, 12, 11), (2048, 256, 16, 16),
                                            (2048, 2048, 25, 34), (256, 128, 13, 11), (96, 128, 9, 15),      # the last three: H*W not a multiple of 4 (layer4: 850 positions)
                                            (256, 256, 200, 272), (1024, 1024, 50, 68), (512, 512, 100, 136)])      # the bench's own bottleneck shapes (layer1 / layer3 / layer2)
def test_conv1x1_matrix_core_gemm_equals_conv2d(vido, ctx, cin, cout, H, W, layout):
    """csrc/conv1x1.hip against conv2d in float64: bottleneck shapes of the detector (64 -> 256, 256 -> 256, 1024 -> 1024 ...), position counts that are not a multiple of
    the 128-wide tile or of 4 (rows then start at 4-byte-aligned addresses only), with and without bias / residual, ReLU / leaky / no activation."""
    from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
    ops = HipOps(ctx)
    g = torch.Generator().manual_seed(cin * 7 + H)
    x = torch.randn(1, cin, H, W, generator=g); w = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin ** 0.5); b = torch.randn(cout, generator=g); r = torch.randn(1, cout, H, W, generator=g)
    assert ops.conv1x1_supported(cin, cout, H * W)
    if layout is not None:
        # the form is read once per process from VIDO_CONV1X1_TN: a forced form runs in a child process (below); here only the library's own choice
        if cin % 64 and layout == 1:
            pytest.skip("the 112-wide form needs input channels in multiples of 64")
        _run_forced_conv1x1(cin, cout, H, W, layout); return
    lay = ops.conv1x1_layout(cin, cout, H * W)
    wp = pack_conv1x1(w, lay).cuda()
    ref0 = torch.nn.functional.conv2d(x.double(), w.double())
    for bias, res, slope in ((None, None, 1.0), (b, None, 0.0), (b, r, 0.0), (b, r, 0.1), (None, r, 1.0)):
        y = ops.conv1x1_bias_act(x.cuda(), wp, bias.cuda() if bias is not None else None, res.cuda() if res is not None else None, slope).cpu()
        ref = ref0 + (bias.double()[None, :, None, None] if bias is not None else 0.0) + (res.double() if res is not None else 0.0)
        ref = torch.nn.functional.leaky_relu(ref, slope)
        err = float((y.double() - ref).abs().max())
        assert err < 2e-5 * max(1.0, float(ref.abs().max())), (cin, cout, H, W, slope, err)
    # shapes outside the plan are refused, not mangled
    assert not ops.conv1x1_supported(48, 128, 1024) and not ops.conv1x1_supported(64, 64, 1024) and not ops.conv1x1_supported(64, 128, 64)
    with pytest.raises(vido.VidoError):
        ops.conv1x1_bias_act(torch.zeros(1, 48, 25, 36, device="cuda"), torch.zeros(4, 6, 64, 4, device="cuda"))      # 48 input channels: no form for it


_C1_SHAPES = [(64, 256, 40, 68), (256, 256, 50, 68), (256, 128, 37, 52), (1024, 1024, 10, 34), (512, 512, 25, 36), (32, 128, 12, 11), (2048, 256, 16, 16), (2048, 2048, 25, 34),
              (256, 128, 13, 11), (96, 128, 9, 15), (256, 256, 200, 272), (1024, 1024, 50, 68), (512, 512, 100, 136)]      # = the shapes of test_conv1x1_matrix_core_gemm_equals_conv2d


@pytest.mark.gpu
@pytest.mark.parametrize("form", [0, 1, 2, 3])      # VIDO_CONV1X1_B3_FORM: 0 = the form the library picks by shape; 1 / 2 / 3 = <1, 6> / <2, 4> / <1, 4> forced (a child process)
def test_split_bf16_conv1x1_is_no_less_accurate_than_the_fp32_instruction(vido, ctx, form):
    """The admissibility rule of the split-bf16 form: on EVERY shape of the 1x1 tests the max-abs error of k_conv1x1_b3 against float64 conv2d must
    not exceed 1.5 x the error of the fp32-matrix-instruction kernel on the same inputs — otherwise it would be narrower arithmetic than the reference's fp32 layers.
    Measured: 0.22 - 0.60 x (two accumulator sets: the leading bf16 product is rounded once per 16 input channels, the fp32 instruction once per channel).  Also: the
    three planes of a weight sum to it exactly, and both arithmetic settings answer through the same entry point (vido_conv1x1_set_arith)."""
    if form:
        import subprocess, sys, os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        code = ("import sys; sys.path.insert(0, %r); import pytest; sys.exit(pytest.main(['-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', "
                "%r + '::test_split_bf16_conv1x1_is_no_less_accurate_than_the_fp32_instruction[0]']))" % (root, os.path.abspath(__file__)))
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VIDO_CONV1X1_B3_FORM=str(form)), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        return
    from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
    ops = HipOps(ctx)
    worst = 0.0
    try:
        for cin, cout, H, W in _C1_SHAPES:
            g = torch.Generator().manual_seed(cin * 7 + H)
            x = torch.randn(1, cin, H, W, generator=g); w = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin ** 0.5); b = torch.randn(cout, generator=g); r = torch.randn(1, cout, H, W, generator=g)
            ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double()) + r.double())
            err = {}
            for arith in (1, 2, 0):
                ops.conv1x1_set_arith(arith)
                lay = ops.conv1x1_layout(cin, cout, H * W)
                assert lay == {0: 3, 1: 0, 2: 2}[arith]
                y = ops.conv1x1_bias_act(x.cuda(), pack_conv1x1(w, lay).cuda(), b.cuda(), r.cuda(), 0.0).cpu()
                err[arith] = float((y.double() - ref).abs().max())
            for a in (0, 2):                                             # split-fp16 (the default), split-bf16
                assert err[a] <= 1.5 * err[1], (cin, cout, H, W, err)
                assert err[a] < 2e-5 * max(1.0, float(ref.abs().max()))
                worst = max(worst, err[a] / err[1])
        assert ops.conv1x1_range_flag() == 0
    finally:
        ops.conv1x1_set_arith(0)
    assert worst <= 1.5


@pytest.mark.gpu
def test_split_fp16_conv1x1_over_its_range_and_past_it(vido, ctx):
    """The split-fp16 form takes activations as they are: the error bar of the test above holds with the activations scaled to 1e-3 and to 300 and with output channels whose
    weights differ by e^(2 N(0, 1)) (per-channel powers of two at pack time); an activation of 70 000 — outside fp16 — raises the context's range flag (once: reading resets)
    instead of passing an infinity on."""
    from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
    ops = HipOps(ctx)
    cin, cout, H, W = 256, 256, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, cin, H, W, generator=g); w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5 * torch.exp(2 * torch.randn(cout, 1, 1, 1, generator=g))
    try:
        for sc in (1e-3, 1.0, 300.0):
            xs = x * sc; ref = torch.nn.functional.conv2d(xs.double(), w.double()); e = {}
            for arith in (1, 0):
                ops.conv1x1_set_arith(arith)
                y = ops.conv1x1_bias_act(xs.cuda(), pack_conv1x1(w, ops.conv1x1_layout(cin, cout, H * W)).cuda(), None, None, 1.0)
                e[arith] = float(((y.cpu().double() - ref) / ref.abs().mean((0, 2, 3), keepdim=True)).pow(2).mean().sqrt())      # rms over outputs relative to their channel's scale
            assert e[0] <= 1.5 * e[1], (sc, e)
            assert ops.conv1x1_range_flag() == 0
        ops.conv1x1_set_arith(0)
        xb = x.clone(); xb[0, 17, 3, 5] = 70000.0
        ops.conv1x1_bias_act(xb.cuda(), pack_conv1x1(w, ops.conv1x1_layout(cin, cout, H * W)).cuda(), None, None, 1.0); torch.cuda.synchronize()
        assert ops.conv1x1_range_flag() == 1 and ops.conv1x1_range_flag() == 0
        ops.conv1x1_set_arith(2)                                         # the bf16 form has fp32's range
        y = ops.conv1x1_bias_act(xb.cuda(), pack_conv1x1(w, ops.conv1x1_layout(cin, cout, H * W)).cuda(), None, None, 1.0).cpu()
        ref = torch.nn.functional.conv2d(xb.double(), w.double())
        assert torch.isfinite(y).all() and float((y.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    finally:
        ops.conv1x1_set_arith(0)


@pytest.mark.gpu
def test_bottleneck_with_matrix_core_1x1_equals_library_path(vido, ctx, monkeypatch):
    """_Bottleneck.forward with conv1 / conv3 / the stride-1 shortcut on csrc/conv1x1.hip (bias, shortcut add and ReLU in the GEMM's epilogue) against the same block on the
    library convolutions + the bias / residual pass."""
    from vido_slam_amd.nets import maskrcnn as M
    from vido_slam_amd.nets.fuse import fold_batchnorm
    from vido_slam_amd.nets.ops import HipOps
    from vido_slam_amd.nets.weights import fill_maskrcnn
    monkeypatch.setattr("vido_slam_amd.nets.ops._C1X1_MIN_TILES", 0)          # (test-sized maps: by default a layer of fewer than 160 tiles stays with the library)
    for cin, mid, cout, stride in ((64, 256, 256, 1), (256, 256, 256, 1), (256, 512, 512, 2), (512, 1024, 1024, 2)):     # first block of layer1 (stride-1 shortcut convolution), a plain one, a stage's first block (stride-2 shortcut), layer3's (+ the strided conv2 on csrc/gconv.hip)
        blk = M._Bottleneck(cin, mid, cout, 32, False, stride)
        fill_maskrcnn(blk); blk = blk.cuda().eval()
        x = torch.randn(1, cin, 40, 52, generator=torch.Generator().manual_seed(4)).cuda()
        with torch.no_grad():
            os.environ["VIDO_CONV1X1"] = "all"                                       # every 1x1 convolution of the block, whatever the size rule of fuse.py says
            try:
                fold_batchnorm(blk, HipOps(ctx))
            finally:
                del os.environ["VIDO_CONV1X1"]
            assert blk._w1p is not None and blk._w3p is not None and (blk.downsample is None or blk._wdp is not None)
            y_fast = blk(x).clone()
            blk._w1p = blk._w3p = blk._wdp = None
            y_lib = blk(x)
        assert float((y_fast - y_lib).abs().max()) < 1e-4 * max(1.0, float(y_lib.abs().max()))


@pytest.mark.gpu
def test_fpn_lateral_with_upsampled_sum_equals_the_reference_form(vido, ctx, monkeypatch):
    """_FPN._inner on csrc/conv1x1.hip (lateral 1x1 convolution + bias + nearest-upsampled coarser level in the GEMM's epilogue, fpn.py:55-66) against conv2d + F.interpolate + add in
    float64, incl. a map whose size is not a multiple of 4 and the top level (no sum)."""
    from vido_slam_amd.nets.ops import HipOps, conv1x1_fills_chip
    ops = HipOps(ctx)
    assert conv1x1_fills_chip(256, 100 * 136) and not conv1x1_fills_chip(256, 50 * 68) and conv1x1_fills_chip(2048, 25 * 34)      # FPN P3 yes, P4 (54 tiles) no, layer4 (112 tiles) yes since round 6
    monkeypatch.setattr("vido_slam_amd.nets.ops._C1X1_MIN_TILES", 0)
    g = torch.Generator().manual_seed(17)
    for cin, cout, H, W in ((512, 256, 24, 36), (256, 256, 26, 34), (1024, 128, 10, 14)):
        conv = torch.nn.Conv2d(cin, cout, 1).cuda()
        x = torch.randn(1, cin, H, W, generator=g).cuda(); top = torch.randn(1, cout, H // 2, W // 2, generator=g).cuda()
        with torch.no_grad():
            y = ops.conv1x1_conv(conv, x, 1.0, residual_up2=top); y0 = ops.conv1x1_conv(conv, x, 1.0)
            ref0 = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double())
            ref = ref0 + torch.nn.functional.interpolate(top.double(), scale_factor=2, mode="nearest")
        assert y is not None and float((y.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
        assert float((y0.double() - ref0).abs().max()) < 2e-5 * max(1.0, float(ref0.abs().max()))
    with pytest.raises((vido.VidoError, AssertionError)):                   # odd map: no half-resolution residual for it
        ops.conv1x1_bias_act(torch.zeros(1, 64, 13, 12, device="cuda"), torch.zeros(4, 8, 64, 4, device="cuda"), None, None, 1.0, residual_up2=torch.zeros(1, 128, 6, 6, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [0, 8, 16])      # VIDO_CONV3X3_H_ROWS: 0 = the block form the library picks by launch size; 8 / 16 forced (a child process)
def test_direct_split_fp16_conv3x3_against_float64_and_the_winograd_kernel(vido, ctx, rows):
    """csrc/conv3x3h.hip (direct 3x3 in split-fp16 arithmetic: FPN outputs / RPN head on P2, P3, the mask head) against float64 conv2d on shapes that cover one block, odd
    sizes with blocks hanging over the edge, a batch, 16 .. 256 input channels, ReLU / leaky / no activation, no bias: error <= 1.5 x the fp32 Winograd kernel's
    (csrc/wino.hip) on the same input and small against the output's magnitude; the range flag stays down."""
    if rows:
        import subprocess, sys, os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        code = ("import sys; sys.path.insert(0, %r); import pytest; sys.exit(pytest.main(['-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', "
                "%r + '::test_direct_split_fp16_conv3x3_against_float64_and_the_winograd_kernel[0]']))" % (root, os.path.abspath(__file__)))
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VIDO_CONV3X3_H_ROWS=str(rows)), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        return
    from vido_slam_amd.nets.ops import HipOps, pack_conv3x3_h, pack_wino3x3
    ops = HipOps(ctx)
    F = torch.nn.functional
    for N, cin, cout, H, W, slope, with_bias in ((1, 16, 128, 16, 16, 0.0, True), (2, 32, 128, 13, 21, 0.1, True), (3, 48, 256, 7, 35, 1.0, False), (1, 256, 256, 50, 68, 0.0, True),
                                                 (5, 256, 256, 14, 14, 0.0, True), (1, 128, 128, 100, 136, 1.0, True), (1, 64, 384, 33, 17, 0.1, True), (2, 49, 128, 30, 40, 0.1, True), (3, 131, 128, 9, 20, 0.1, True),
                                                 (1, 128, 64, 40, 50, 0.1, True), (2, 64, 64, 17, 33, 0.1, True), (1, 64, 32, 50, 70, 0.1, True), (2, 32, 32, 9, 16, 1.0, False)):
        assert ops.ctx.lib.vido_conv3x3_h_supported(N, cin, cout, H, W)
        g = torch.Generator().manual_seed(cin + H)
        x = torch.randn(N, cin, H, W, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5) * torch.exp(torch.randn(cout, 1, 1, 1, generator=g)); b = torch.randn(cout, generator=g) if with_bias else None
        pre = F.conv2d(x.double(), w.double(), b.double() if with_bias else None, padding=1); ref = F.leaky_relu(pre, slope)
        xc = x.cuda(); bc = b.cuda() if with_bias else None
        yh = ops.conv3x3_h_bias_act(xc, pack_conv3x3_h(w).cuda(), bc, cout, slope).cpu()
        assert float((yh.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), (N, cin, cout, H, W)
        form = ops.wino3x3_form(N, cin, cout, H, W)
        yw = ops.wino3x3_bias_act(xc, pack_wino3x3(w, form).cuda(), bc if with_bias else torch.zeros(cout, device="cuda"), cout, slope, form).cpu()
        # the error measure: rms over all outputs, each relative to its channel's mean magnitude (the channels' weights differ by e^N(0,1); a max over 10^5 .. 10^7 outputs of
        # the largest channel is a tail statistic that moves by 3x between kernels of equal rms error: tools/r6/conv3x3_h_errors.py)
        scale = pre.abs().mean((0, 2, 3), keepdim=True).clamp_min(1e-30)                 # (before the activation: a ReLU can leave a channel all zeros)
        eh, ew = (float(((y.double() - ref) / scale).pow(2).mean().sqrt()) for y in (yh, yw))
        assert eh <= 1.5 * ew and eh < 1e-6, (N, cin, cout, H, W, eh, ew)
    torch.cuda.synchronize()
    assert ops.conv1x1_range_flag() == 0
    assert ops.ctx.lib.vido_conv3x3_h_supported(1, 24, 128, 16, 16) and ops.ctx.lib.vido_conv3x3_h_supported(1, 32, 64, 16, 16) and not ops.ctx.lib.vido_conv3x3_h_supported(1, 32, 96, 16, 16)
    xb = torch.randn(1, 16, 16, 16); xb[0, 3, 5, 5] = 1e5
    ops.conv3x3_h_bias_act(xb.cuda(), pack_conv3x3_h(torch.randn(128, 16, 3, 3)).cuda(), None, 128, 1.0); torch.cuda.synchronize()
    assert ops.conv1x1_range_flag() == 1


@pytest.mark.gpu
def test_split_fp16_fully_connected_layer_against_float64_and_the_library(vido, ctx):
    """csrc/fch.hip (the box head's fc6 / fc7 as split-fp16 GEMMs split over K) against float64 on the detector's two shapes, a ragged row count, a small layer and per-output
    weight scales e^N(0,1): rms error (each output relative to its column's mean magnitude before the activation) <= 1.5 x the library's fp32 GEMM on the same input; the K split
    of the host rule; shapes the kernel does not take answer 0; the range flag."""
    from vido_slam_amd.nets.ops import HipOps, pack_conv1x1
    ops = HipOps(ctx); lib = ops.ctx.lib
    F = torch.nn.functional
    assert lib.vido_fc_h_splitk(1000, 12544, 1024) == 4 and lib.vido_fc_h_splitk(1000, 1024, 1024) == 4 and lib.vido_fc_h_splitk(100, 12544, 1024) == 8 and lib.vido_fc_h_splitk(37, 64, 128) == 2
    assert lib.vido_fc_h_splitk(1000, 12544, 1000) == 0 and lib.vido_fc_h_splitk(1000, 48, 128) == 0 and lib.vido_fc_h_splitk(0, 64, 128) == 0
    for rows, k, outs, slope in ((1000, 12544, 1024, 0.0), (1000, 1024, 1024, 0.0), (333, 2048, 256, 0.1), (37, 64, 128, 1.0)):
        g = torch.Generator().manual_seed(k + rows)
        x = torch.relu(torch.randn(rows, k, generator=g)); w = torch.randn(outs, k, generator=g) / k ** 0.5 * torch.exp(torch.randn(outs, 1, generator=g)); b = torch.randn(outs, generator=g)
        pre = x.double() @ w.double().t() + b.double(); ref = F.leaky_relu(pre, slope); sc = pre.abs().mean(0, keepdim=True)
        yh = ops.fc_h(x.cuda(), pack_conv1x1(w.reshape(outs, k, 1, 1), 3).cuda(), b.cuda(), outs, slope)
        yl = F.leaky_relu(F.linear(x.cuda(), w.cuda(), b.cuda()), slope)
        eh, el = (float(((y.cpu().double() - ref) / sc).pow(2).mean().sqrt()) for y in (yh, yl))
        assert eh <= 1.5 * el and eh < 1e-6, (rows, k, outs, eh, el)
        lin = torch.nn.Linear(k, outs); lin.weight.data.copy_(w); lin.bias.data.copy_(b); lin = lin.cuda()
        assert torch.equal(ops.fc_h_linear(lin, x.cuda(), slope), yh)
    torch.cuda.synchronize()
    assert ops.conv1x1_range_flag() == 0
    xb = torch.randn(130, 64); xb[5, 7] = 1e5
    ops.fc_h(xb.cuda(), pack_conv1x1(torch.randn(128, 64, 1, 1), 3).cuda(), None, 128, 1.0); torch.cuda.synchronize()
    assert ops.conv1x1_range_flag() == 1


@pytest.mark.gpu
def test_mask_logit_select_equals_the_logits_layer_on_the_label_channel(vido, ctx):
    """vido_mask_logit_select (csrc/nets.hip): sigmoid(mask_fcn_logits(feat))[n, label[n]] for every detection, computed for that one channel only — against the float64
    evaluation of the same expression; labels at both ends of the class range; odd sizes."""
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    for n, c, H, W, classes in ((100, 256, 28, 28, 81), (7, 48, 5, 9, 3), (1, 300, 1, 1, 2)):
        g = torch.Generator().manual_seed(n + c)
        feat = torch.relu(torch.randn(n, c, H, W, generator=g)); conv = torch.nn.Conv2d(c, classes, 1)
        conv.weight.data = torch.randn(classes, c, 1, 1, generator=g) / c ** 0.5; conv.bias.data = torch.randn(classes, generator=g)
        labels = torch.randint(0, classes, (n,), generator=g); labels[0] = 0; labels[-1] = classes - 1
        ref = torch.sigmoid(torch.nn.functional.conv2d(feat.double(), conv.weight.data.double(), conv.bias.data.double()))[torch.arange(n), labels][:, None]
        got = ops.mask_logit_select(feat.cuda(), conv.cuda(), labels.cuda()).cpu()
        assert tuple(got.shape) == (n, 1, H, W) and float((got.double() - ref).abs().max()) < 2e-7


@pytest.mark.gpu
def test_roi_levels_equals_the_level_mapper_expression(vido, ctx):
    """vido_roi_levels (one launch) against the torch expression of maskrcnn_benchmark's LevelMapper it replaces, on 200 000 random boxes, boxes at the level boundaries
    (sides 112 / 224 / 448 / 896 -+ one ulp), degenerate and huge boxes: identical levels."""
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    g = torch.Generator().manual_seed(4)
    xy = torch.rand(200000, 2, generator=g) * 1000; wh = torch.exp(torch.rand(200000, 2, generator=g) * 9 - 1)
    boxes = torch.cat([xy, xy + wh], 1)
    edge = []
    for side in (56.0, 112.0, 224.0, 448.0, 896.0):
        for d in (-1e-3, -1e-4, 0.0, 1e-4, 1e-3):
            edge.append([10.0, 20.0, 10.0 + side - 1 + d, 20.0 + side - 1])
    boxes = torch.cat([boxes, torch.tensor(edge), torch.tensor([[0.0, 0.0, 0.0, 0.0], [5.0, 5.0, 4.0, 4.0], [0.0, 0.0, 1e5, 1e5], [0.0, 0.0, 0.5, 0.5]])]).cuda()
    for k_min, k_max in ((2.0, 5.0), (3.0, 4.0)):
        area = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
        ref = torch.floor(4 + torch.log2(torch.sqrt(area) / 224 + 1e-6)).clamp(min=k_min, max=k_max).to(torch.int64) - int(k_min)
        got = ops.roi_levels(boxes, k_min, k_max)
        assert got.dtype == torch.int32 and torch.equal(got.to(torch.int64), ref)



@pytest.mark.gpu
def test_det_order_equals_the_torch_tail_of_the_static_head(vido, ctx):
    """vido_det_order (one launch) against the torch expressions of analyse_image_static it replaces: confidence test, stable descending sort with ties, labels of the live slots,
    their count — random scores with ties, n_det below / at / above the slot count, cap 100 and 1."""
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    g = torch.Generator().manual_seed(8)
    for cap, nd in ((100, 37), (100, 100), (100, 0), (100, 250), (1, 1), (64, 63)):
        scores = (torch.randint(0, 40, (cap,), generator=g).float() / 40).cuda()      # many ties
        labels = torch.randint(1, 81, (cap,), generator=g).cuda(); n_det = torch.tensor([nd], dtype=torch.int32).cuda()
        for conf in (0.8, 0.0, 2.0):
            live = (scores > conf) & (torch.arange(cap, device="cuda") < n_det[0])
            order = torch.sort(torch.where(live, scores, scores.new_full((), -1.0)), descending=True, stable=True)[1]
            lab = torch.where(live, labels, torch.zeros_like(labels))[order]
            o2, l2, n2 = ops.det_order(scores, labels, n_det, conf)
            assert torch.equal(o2, order) and torch.equal(l2, lab) and int(n2) == int(live.sum())



@pytest.mark.gpu
def test_deconv2x2_as_split_fp16_gemm_against_float64_and_the_library(vido, ctx):
    """vido_deconv2x2_bias_act (csrc/conv1x1.hip, RES 3: the mask head's transposed convolution as one GEMM with a scatter epilogue) against float64 conv_transpose2d: rms error
    (per output channel's mean magnitude) <= 1.5 x the library's fp32 on the same input; the detector's shape, a small batch of odd maps, ReLU / none, no bias."""
    from vido_slam_amd.nets.ops import HipOps
    ops = HipOps(ctx)
    F = torch.nn.functional
    for n, cin, cout, H, W, slope, with_bias in ((100, 256, 256, 14, 14, 0.0, True), (3, 64, 128, 6, 10, 1.0, False), (1, 32, 128, 16, 8, 0.1, True)):
        g = torch.Generator().manual_seed(n + cin)
        conv = torch.nn.ConvTranspose2d(cin, cout, 2, 2, 0, bias=with_bias)
        conv.weight.data = torch.randn(cin, cout, 2, 2, generator=g) / cin ** 0.5 * torch.exp(torch.randn(1, cout, 1, 1, generator=g))
        if with_bias: conv.bias.data = torch.randn(cout, generator=g)
        x = torch.relu(torch.randn(n, cin, H, W, generator=g))
        pre = F.conv_transpose2d(x.double(), conv.weight.data.double(), conv.bias.data.double() if with_bias else None, 2); ref = F.leaky_relu(pre, slope)
        sc = pre.abs().mean((0, 2, 3), keepdim=True)
        convc = conv.cuda()
        got = ops.deconv2x2_conv(convc, x.cuda(), slope)
        assert got is not None and tuple(got.shape) == (n, cout, 2 * H, 2 * W)
        lib = F.leaky_relu(convc(x.cuda()), slope)
        eg, el = (float(((y.cpu().double() - ref) / sc).pow(2).mean().sqrt()) for y in (got, lib))
        assert eg <= 1.5 * el and eg < 1e-6, (n, cin, cout, H, W, eg, el)
    assert not ops.ctx.lib.vido_deconv2x2_supported(1, 32, 64, 16, 8) and not ops.ctx.lib.vido_deconv2x2_supported(1, 32, 128, 3, 3)
    torch.cuda.synchronize(); assert ops.conv1x1_range_flag() == 0
