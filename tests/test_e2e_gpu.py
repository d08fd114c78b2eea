"""GPU: network-node rewrites (nets/fuse.py: folded batch norms + fused epilogue, hipGraph replay) against the plain module graphs, and the
pipelined chain nets -> hand-over -> System::TrackRGBD (pipeline.NetNodes / EndToEnd)."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def test_folded_batchnorm_equals_plain_graph(vido):
    from vido_slam_amd import nets
    ctx = vido.Context(width=640, height=480, max_batch=1)
    ops = nets.HipOps(ctx)
    cfg = nets.MaskRCNNConfig(blocks=(3, 4, 6, 3), groups=4, width_per_group=4, res2_out=32, stem_out=16, fpn_out=16, mlp_dim=64, num_classes=7,
                              mask_layers=(16, 16, 16, 16), detections_per_img=20)
    net = nets.fill_maskrcnn(nets.MaskRCNN(ops, cfg), 3).eval().cuda()
    x = torch.rand(1, 3, 160, 224, device="cuda") * 255
    with torch.no_grad():
        ref = net.backbone(x)
        n = nets.fold_batchnorm(net, ops)
        assert n == 16 * 3 + 4 + 1                                   # 16 bottlenecks x 3 + 4 downsample branches + stem
        out = net.backbone(x)
    for a, b in zip(out, ref):
        assert rel_err(a, b) < 1e-4
    md = nets.fill_deterministic(nets.MonoDepth2(), 2).eval().cuda()
    img = torch.rand(1, 3, 192, 640, device="cuda")
    with torch.no_grad():
        ref = md(img)
        assert nets.fold_batchnorm(md, ops) == 20                    # ResNet-18: conv1 + 8 blocks x 2 + 3 downsample branches
        out = md(img)
    assert rel_err(out, ref) < 1e-4                     # folding moves the scale inside the fp32 accumulation: rounding-level differences through 20 layers


def test_graph_replay_equals_eager(vido):
    from vido_slam_amd import nets
    ctx = vido.Context(width=640, height=480, max_batch=1)
    ops = nets.HipOps(ctx)
    lfn = nets.fill_deterministic(nets.LiteFlowNet(ops.correlation, epilogue=ops.bias_act_), 1).eval().cuda()
    rng = np.random.RandomState(0)
    a = torch.as_tensor((rng.rand(128, 192, 3) * 255).astype(np.uint8), device="cuda"); b = torch.as_tensor((rng.rand(128, 192, 3) * 255).astype(np.uint8), device="cuda")
    fn = lambda p, q: nets.analyse_flow(lfn, p, q)
    ref = fn(a, b).clone()
    g = nets.Graphed(fn, [a, b])
    out = g(a, b)
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1e-6
    out2 = g(b, a).clone(); torch.cuda.synchronize()
    assert rel_err(out2, fn(b, a)) < 1e-6                          # new inputs flow through the static buffers


def test_pipelined_chain_tracks_and_hands_over(tmp_path, vido):
    """EndToEnd on a short clip, 160x... no: full 640x480 (the facade's ORB grid needs it), full-size nets."""
    from vido_slam_amd import pipeline, synth
    from vido_slam_amd.system import System
    from test_system_gpu import _settings
    n = 6
    scene = synth.convoy_scene(n + 1)
    net_ctx = vido.Context(width=640, height=480, max_batch=1)
    nodes = pipeline.NetNodes(net_ctx, 480, 640)
    assert nodes.folded > 100
    slam = System(); slam.Init(_settings(tmp_path, scene), System.RGBD)
    e2e = pipeline.EndToEnd(nodes, slam, n_image=10 ** 6, feed="given")
    for k in range(n):
        g, d, f, m = scene.frame(k)
        e2e.push(synth.gray_to_bgr(g), (np.ascontiguousarray(d, np.float32), np.ascontiguousarray(f, np.float32), np.ascontiguousarray(m, np.int32)))
    e2e.finish()
    assert len(e2e.poses) == n
    for k, T in enumerate(e2e.poses):
        E = T.astype(np.float64) @ np.linalg.inv(scene.Tcw(k))
        assert np.linalg.norm(E[:3, 3]) < 0.05, (k, E)
    assert e2e.stats[-1]["n_objects"] >= 4                          # the five convoy objects are tracked as dynamic
    # the network hand-over buffers were filled (flow of the last frame: finite numbers, mask u8 range, depth MONO16 range)
    hb = e2e.host[(n - 1) % e2e.RING]
    assert np.isfinite(hb["flow"].numpy()).all() and hb["depth"].numpy().max() <= 65535 and hb["mask"].numpy().min() >= 0
    e2e.close(); slam.close()
