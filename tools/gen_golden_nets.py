#!/usr/bin/env python3
"""Golden vectors for rows N1/N2: runs the REFERENCE torch modules, imported from /root/reference in
this container only, on small seeded inputs with deterministic weights and stores inputs + outputs in
tests/golden/nets_kats.npz.  Nothing of the reference travels: the fixture is data.

  * LiteFlowNet  — flow_net/src/layers.py Network.  Its cost volume is a cupy/CUDA kernel that cannot run here (no
    CUDA, no cupy): `cupy` is absent from the image and FunctionCorrelation is replaced by the C oracle restatement
    (oracle/nets_oracle.c vo_correlation, itself pinned by tests/test_oracle_cpu.py), everything else is the
    reference's own module graph.  `.cuda()` is patched to the identity and the checkpoint load is skipped.
  * MonoDepth2   — mono_depth2/src/networks/depth_decoder.py DepthDecoder, run as is.  The reference encoder is
    torchvision.models.resnet18, and torchvision is absent from this image: the encoder restatement is pinned by
    the published ResNet-18 state-dict layout instead (tests/test_nets_cpu.py) — "parity unpinned" for its values.
"""
import os, sys, types, importlib
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/thirdparty"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import vido_slam_amd                                         # noqa: E402
from vido_slam_amd.nets.weights import fill_deterministic    # noqa: E402
from vido_slam_amd.synth import make_canvas                  # noqa: E402
import pyoracle                                              # noqa: E402


def oracle_correlation(first, second, stride):
    out = pyoracle.correlation(first.numpy().astype(np.float32), second.numpy().astype(np.float32), int(stride))
    return torch.from_numpy(np.asarray(out))


def reference_liteflownet():
    cupy = types.ModuleType("cupy"); cupy.memoize = lambda **kw: (lambda f: f); sys.modules["cupy"] = cupy
    sys.path.insert(0, os.path.join(REF, "flow_net/src"))
    saved_load, saved_lsd, saved_cuda = torch.load, torch.nn.Module.load_state_dict, torch.Tensor.cuda
    torch.load = lambda *a, **k: {}
    torch.nn.Module.load_state_dict = lambda self, sd, *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        layers = importlib.import_module("layers")
        net = layers.Network("unused")
    finally:
        torch.load, torch.nn.Module.load_state_dict = saved_load, saved_lsd
    layers.correlation.FunctionCorrelation = lambda tenFirst, tenSecond, intStride: oracle_correlation(tenFirst, tenSecond, intStride)
    sys.path.pop(0); del sys.modules["layers"]
    return net.eval(), saved_cuda


def reference_depth_decoder():
    sys.path.insert(0, os.path.join(REF, "mono_depth2/src"))
    import layers as md_layers                               # noqa: F401  (depth_decoder does `from layers import *`)
    spec = importlib.util.spec_from_file_location("ref_depth_decoder", os.path.join(REF, "mono_depth2/src/networks/depth_decoder.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    sys.path.pop(0)
    return mod.DepthDecoder(num_ch_enc=np.array([64, 64, 128, 256, 512]), scales=range(4)).eval()


def main():
    torch.manual_seed(0); torch.set_grad_enabled(False)
    out = {}
    # ---- N1 --------------------------------------------------------------------------------------------------
    H, W = 64, 96
    canvas = make_canvas(H + 8, W + 8, seed=77, n_rect=30)
    img = lambda dx, dy: np.stack([canvas[dy:dy + H, dx:dx + W], np.roll(canvas, 3, 1)[dy:dy + H, dx:dx + W], np.roll(canvas, 5, 0)[dy:dy + H, dx:dx + W]], 0)
    first, second = img(4, 4), img(2, 3)
    net, saved_cuda = reference_liteflownet()
    fill_deterministic(net, seed=11)
    t1 = torch.from_numpy(first.astype(np.float32) / 255.0)[None]; t2 = torch.from_numpy(second.astype(np.float32) / 255.0)[None]
    flow = net(t1.clone(), t2.clone())
    torch.Tensor.cuda = saved_cuda
    out.update(lfn_first=first, lfn_second=second, lfn_flow=flow.numpy(), lfn_seed=np.int32(11))
    print("LiteFlowNet flow", flow.shape, float(flow.abs().mean()), float(flow.abs().max()))
    # ---- N2 decoder ------------------------------------------------------------------------------------------
    dec = reference_depth_decoder()
    fill_deterministic(dec, seed=12)
    rng = np.random.RandomState(5)
    shapes = [(1, 64, 32, 64), (1, 64, 16, 32), (1, 128, 8, 16), (1, 256, 4, 8), (1, 512, 2, 4)]
    feats = [torch.from_numpy(rng.uniform(0, 1.5, s).astype(np.float32)) for s in shapes]
    res = dec(feats)
    for s in range(4):
        out["md_disp%d" % s] = res[("disp", s)].numpy()
    out.update(md_seed=np.int32(12), md_feat_seed=np.int32(5))
    print("MonoDepth2 disp0", res[("disp", 0)].shape, float(res[("disp", 0)].mean()))
    keys = list(dec.state_dict().keys())
    out["md_decoder_keys"] = np.array(keys)
    out["lfn_keys"] = np.array(list(net.state_dict().keys()))
    np.savez_compressed(os.path.join(REPO, "tests/golden/nets_kats.npz"), **out)
    print("wrote tests/golden/nets_kats.npz")


if __name__ == "__main__":
    main()
