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
        assert prop.shape == G["proposals"].shape and rel_err(prop.cpu().numpy(), G["proposals"]) < TOL and rel_err(obj.cpu().numpy(), G["objectness"]) < TOL
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
    assert out["proposals"].shape == G["proposals"].shape and rel_err(out["proposals"].cpu().numpy(), G["proposals"]) < 5e-3
    assert np.array_equal(out["labels"].cpu().numpy(), G["det_labels"])
    assert rel_err(out["masks"].cpu().numpy(), G["det_masks"]) < 5e-3


def test_full_size_graph_runs(ctx):
    """X-101-32x8d-FPN at the node's 800x1088 feed (random-init weights): shapes, dtypes, finiteness."""
    net = nets.fill_maskrcnn(nets.MaskRCNN(nets.HipOps(ctx)), 1).eval().cuda()
    bgr = (np.random.RandomState(0).rand(375, 1242, 3) * 255).astype(np.uint8)
    img, labels = nets.analyse_image(net, bgr)
    assert tuple(img.shape) == (375, 1242) and img.dtype == torch.uint8
    out = net(torch.rand(1, 3, 1088, 800, device="cuda"))
    assert out["proposals"].shape[1] == 4 and 0 < len(out["proposals"]) <= 1000 and bool(torch.isfinite(out["proposals"]).all())
    assert out["masks"].shape[1:] == (1, 28, 28)
