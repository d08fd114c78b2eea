"""Rows N1/N2 on CPU: our torch modules (cost volume = plain-PyTorch reference here, the HIP kernel in
test_nets_modules_gpu.py) against golden outputs of the REFERENCE modules (tools/gen_golden_nets.py)."""
import os
import numpy as np
import torch
import vido_slam_amd
from vido_slam_amd import nets

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "nets_kats.npz"))
TOL = 1e-4            # relative to the output's max magnitude (fp32, different summation orders)


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def test_liteflownet_matches_reference_module():
    net = nets.fill_deterministic(nets.LiteFlowNet(nets.correlation_torch_reference), int(G["lfn_seed"])).eval()
    assert list(net.state_dict().keys()) == [str(k) for k in G["lfn_keys"]]          # checkpoints load unchanged
    a = torch.from_numpy(G["lfn_first"].astype(np.float32) / 255.0)[None]; b = torch.from_numpy(G["lfn_second"].astype(np.float32) / 255.0)[None]
    flow = net(a, b).numpy()
    assert flow.shape == G["lfn_flow"].shape
    assert rel_err(flow, G["lfn_flow"]) < TOL
    assert np.abs(G["lfn_flow"]).max() > 0.1                                          # the fixture is not degenerate


def test_correlation_torch_reference_matches_oracle(oracle):
    rng = np.random.RandomState(3)
    for stride, (H, W) in ((1, (9, 13)), (2, (10, 14)), (2, (11, 15))):
        f1 = rng.randn(2, 5, H, W).astype(np.float32); f2 = rng.randn(2, 5, H, W).astype(np.float32)
        ref = oracle.correlation(f1, f2, stride)
        got = nets.correlation_torch_reference(torch.from_numpy(f1), torch.from_numpy(f2), stride).numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-5


def test_depth_decoder_matches_reference_module():
    dec = nets.fill_deterministic(nets.DepthDecoder(), int(G["md_seed"])).eval()
    assert list(dec.state_dict().keys()) == [str(k) for k in G["md_decoder_keys"]]
    rng = np.random.RandomState(int(G["md_feat_seed"]))
    shapes = [(1, 64, 32, 64), (1, 64, 16, 32), (1, 128, 8, 16), (1, 256, 4, 8), (1, 512, 2, 4)]
    feats = [torch.from_numpy(rng.uniform(0, 1.5, s).astype(np.float32)) for s in shapes]
    with torch.no_grad():
        out = dec(feats)
    for s in range(4):
        assert rel_err(out[("disp", s)].numpy(), G["md_disp%d" % s]) < TOL


def test_resnet18_encoder_layout():
    """torchvision is absent from the image, so the encoder is pinned by the published ResNet-18 layout: 11,689,512
    parameters, 122 state-dict entries, and the well-known key names/shapes."""
    enc = nets.ResnetEncoder18()
    sd = enc.state_dict()
    assert len(sd) == 122
    assert sum(p.numel() for p in enc.parameters()) == 11689512
    expect = {"encoder.conv1.weight": (64, 3, 7, 7), "encoder.bn1.running_var": (64,), "encoder.layer1.1.conv2.weight": (64, 64, 3, 3),
              "encoder.layer2.0.downsample.0.weight": (128, 64, 1, 1), "encoder.layer2.0.downsample.1.num_batches_tracked": (),
              "encoder.layer4.1.bn2.bias": (512,), "encoder.fc.weight": (1000, 512)}
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    assert "encoder.layer1.0.downsample.0.weight" not in sd
    nets.fill_deterministic(enc, 4).eval()
    with torch.no_grad():
        feats = enc(torch.rand(1, 3, 64, 128))
    assert [tuple(f.shape[1:]) for f in feats] == [(64, 32, 64), (64, 16, 32), (128, 8, 16), (256, 4, 8), (512, 2, 4)]


def test_analyse_wrappers_shapes():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 255, (70, 100, 3)).astype(np.uint8)
    lfn = nets.fill_deterministic(nets.LiteFlowNet(nets.correlation_torch_reference), 1).eval()
    flow = nets.analyse_flow(lfn, img, np.roll(img, 2, 1))
    assert tuple(flow.shape) == (70, 100, 2) and bool(torch.isfinite(flow).all())
    md = nets.fill_deterministic(nets.MonoDepth2(), 2).eval()
    d = nets.analyse_depth(md, img, feed=(64, 128))
    assert tuple(d.shape) == (70, 100) and int(d.min()) == 0 and int(d.max()) == 65535


def test_hip_ops_have_no_cpu_fallback():
    class FakeCtx: pass
    ops = nets.HipOps(FakeCtx())
    try:
        ops.correlation(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4), 1)
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("CPU tensors must be refused")
