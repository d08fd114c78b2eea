"""Rows N1/N2 on CPU: our torch modules (cost volume = plain-PyTorch reference here, the HIP kernel in
test_nets_modules_gpu.py) against golden outputs of the REFERENCE modules (tools/gen_golden_nets.py)."""
import os
import numpy as np
import torch
import vido_slam_amd
from vido_slam_amd import nets

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "nets_kats.npz"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def _seeded_encoder():
    """the encoder tools/gen_golden_refimpl.py::encoder_case builds (same seed, same order of draws)"""
    from vido_slam_amd.nets.monodepth2 import ResnetEncoder18
    torch.manual_seed(5)
    enc = ResnetEncoder18().eval()
    with torch.no_grad():
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    img = torch.rand(3, 64, 96)
    return enc, img


def test_resnet18_encoder_matches_the_float64_forward():
    """ResnetEncoder18 (resnet_encoder.py:87-98 over torchvision's ResNet-18, which this image lacks) against tests/golden/refimpl_kats.npz: the five feature maps of a
    hand-written float64 numpy ResNet-18 forward (tests/refimpl/resnet18_f64.py: its own convolution, batch norm, max pool and block wiring) on a 64 x 96 image with
    randomised batch-norm statistics — pins the encoder's ARITHMETIC (strides, paddings, where the ReLUs and the shortcut sit), not only its parameter layout."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "refimpl_kats.npz"))
    enc, img = _seeded_encoder()
    assert np.array_equal(img.numpy(), G["enc_image"])                                  # the fixture's own input
    with torch.no_grad():
        feats = enc(img[None])
    assert [tuple(f.shape[1:]) for f in feats] == [(64, 32, 48), (64, 16, 24), (128, 8, 12), (256, 4, 6), (512, 2, 3)]
    for i, f in enumerate(feats):
        ref = G["enc_feat%d" % i]
        assert np.abs(f[0].numpy() - ref).max() <= 5e-6 * np.abs(ref).max(), i          # fp32 module against float64: observed 5e-7
    # and the float64 forward itself regenerates the fixture (the generator is deterministic)
    from tests.refimpl.resnet18_f64 import resnet18_encoder_f64
    again = resnet18_encoder_f64(enc.state_dict(), img.numpy())
    assert all(np.array_equal(a.astype(np.float32), G["enc_feat%d" % i]) for i, a in enumerate(again))


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


def test_pack_gconv3x3_is_the_operand_order_of_the_kernel():
    """nets/ops.py::pack_gconv3x3 (host side of csrc/gconv.hip): element (group g, output channel co, input channel ci, tap t) of the convolution weight sits where the kernel's
    A operand reads it — [g][co / 32][ci / 8][t][ci % 8][co % 32] for >= 32 channels per group, [g][ci / 8][t][ci % 8][co (16 slots, zero above cpg)] for 16 / 8."""
    from vido_slam_amd.nets.ops import pack_gconv3x3
    g = torch.Generator().manual_seed(3)
    for groups, cpg in ((2, 64), (3, 32), (4, 16), (5, 8)):
        w = torch.randn(groups * cpg, cpg, 3, 3, generator=g)
        p = pack_gconv3x3(w, groups)
        for _ in range(200):
            gi, co, ci, t = (int(torch.randint(0, n, (1,), generator=g)) for n in (groups, cpg, cpg, 9))
            ref = float(w[gi * cpg + co, ci, t // 3, t % 3])
            got = float(p[gi, co // 32, ci // 8, t, ci % 8, co % 32]) if cpg % 32 == 0 else float(p[gi, ci // 8, t, ci % 8, co])
            assert got == ref
        if cpg < 16:
            assert p.shape[-1] == 16 and float(p[..., cpg:].abs().max()) == 0.0
    assert pack_gconv3x3(torch.zeros(24, 12, 3, 3), 2) is None and pack_gconv3x3(torch.zeros(64, 32, 1, 1), 2) is None


def test_miopen_find_db_is_offered_only_to_its_own_miopen_build(monkeypatch, tmp_path):
    """pipeline._offer_miopen_db: the shipped find-db (file names carry the MIOpen version it was recorded with) is put into MIOPEN_USER_DB_PATH only when torch reports that
    version; an explicit setting and VIDO_NO_MIOPEN_DB win; a read-only package directory gets a private copy."""
    import os
    from vido_slam_amd import pipeline
    files = [f for f in os.listdir(pipeline._MIOPEN_DB) if ".HIP." in f]
    assert files
    tag = files[0].split(".HIP.")[1].split("_")[:3]
    ver = int(tag[0]) * 1000000 + int(tag[1]) * 1000 + int(tag[2])
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False); monkeypatch.delenv("VIDO_NO_MIOPEN_DB", raising=False)
    monkeypatch.setattr(torch.backends.cudnn, "version", lambda: ver)
    assert pipeline._offer_miopen_db() == pipeline._MIOPEN_DB and os.environ["MIOPEN_USER_DB_PATH"] == pipeline._MIOPEN_DB
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setattr(torch.backends.cudnn, "version", lambda: ver + 1000)          # another minor version: not offered
    assert pipeline._offer_miopen_db() is None and "MIOPEN_USER_DB_PATH" not in os.environ
    monkeypatch.setattr(torch.backends.cudnn, "version", lambda: ver)
    monkeypatch.setenv("VIDO_NO_MIOPEN_DB", "1")
    assert pipeline._offer_miopen_db() is None
    monkeypatch.delenv("VIDO_NO_MIOPEN_DB"); monkeypatch.setenv("MIOPEN_USER_DB_PATH", str(tmp_path))
    assert pipeline._offer_miopen_db() is None and os.environ["MIOPEN_USER_DB_PATH"] == str(tmp_path)
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setattr(os, "access", lambda p, m: False)                               # read-only installation
    got = pipeline._offer_miopen_db()
    assert got is not None and got != pipeline._MIOPEN_DB and sorted(os.listdir(got)) >= sorted(files)


def test_pack_conv1x1_is_the_operand_order_of_the_kernel():
    """nets/ops.py::pack_conv1x1 (host side of csrc/conv1x1.hip): element (co, k) of the 1x1 weight sits where lane 32 * (k & 1) + co % 32 of the wave that owns row block
    co / 32 reads its (k % 8) / 2-th operand of the group of four k-pairs k / 8 — v_mfma_f32_32x32x2's A operand: lane l supplies A[row l % 32][k l / 32]."""
    from vido_slam_amd.nets.ops import pack_conv1x1
    g = torch.Generator().manual_seed(5)
    for cout, cin in ((128, 32), (256, 64), (384, 96)):
        w = torch.randn(cout, cin, 1, 1, generator=g)
        p = pack_conv1x1(w)
        assert tuple(p.shape) == (cout // 32, cin // 8, 64, 4) and p.is_contiguous()
        for _ in range(300):
            co, k = (int(torch.randint(0, n, (1,), generator=g)) for n in (cout, cin))
            assert float(p[co // 32, k // 8, 32 * (k & 1) + co % 32, (k % 8) // 2]) == float(w[co, k, 0, 0])
    assert pack_conv1x1(torch.zeros(64, 32, 1, 1)) is None and pack_conv1x1(torch.zeros(128, 48, 1, 1)) is None and pack_conv1x1(torch.zeros(128, 32, 3, 3)) is None
    # layout 1 (128 x 112 tiles on v_mfma_f32_16x16x4: lane l supplies A[row l % 16][k l / 16]): element (co, k) is the (k % 16) / 4-th operand that lane 16 * (k & 3) + co % 16
    # reads for the group of four k-steps k / 16 of the 16-row fragment co / 16
    for cout, cin in ((128, 64), (256, 128)):
        w = torch.randn(cout, cin, 1, 1, generator=g)
        p = pack_conv1x1(w, 1)
        assert tuple(p.shape) == (cout // 16, cin // 16, 64, 4) and p.is_contiguous()
        for _ in range(300):
            co, k = (int(torch.randint(0, n, (1,), generator=g)) for n in (cout, cin))
            assert float(p[co // 16, k // 16, 16 * (k & 3) + co % 16, (k % 16) // 4]) == float(w[co, k, 0, 0])
    assert pack_conv1x1(torch.zeros(128, 96, 1, 1), 1) is None
    # layout 2 (the split-bf16 form, v_mfma_f32_32x32x16_bf16: lane l supplies A[row l % 32][k 8 (l / 32) .. + 7]): three bf16 planes whose sum is the weight EXACTLY
    from vido_slam_amd.nets.ops import split_bf16x3
    w = torch.randn(256, 64, 1, 1) * torch.logspace(-6, 3, 64)[None, :, None, None]
    p = pack_conv1x1(w, 2)
    assert tuple(p.shape) == (8, 4, 3, 64, 8) and p.dtype == torch.int16
    planes = split_bf16x3(w.reshape(256, 64))
    assert torch.equal(planes[0].double() + planes[1].double() + planes[2].double(), w.reshape(256, 64).double())
    for co, k in ((0, 0), (37, 9), (255, 63), (128, 16), (31, 8)):
        for pl in range(3):
            assert int(p[co // 32, k // 16, pl, 32 * ((k % 16) // 8) + co % 32, k % 8]) == int(planes[pl].view(torch.int16)[co, k])
    # layout 3 (the split-fp16 form, v_mfma_f32_32x32x16_f16, the default): two fp16 planes of the rows scaled by powers of two, the low one scaled by 2^11, + the inverse row
    # scales behind them; w == inv * (h + l / 2048) to within 2^-22 |w| (two roundings to 11 bits; 2^-24.5 rms) for everything down to 2^-26 of its row's maximum (smaller
    # entries: an absolute error below 2^-48 of the maximum), and no plane entry is infinite
    from vido_slam_amd.nets.ops import split_f16x2
    w2 = w.reshape(256, 64).clone(); w2[7] = 0.0; w2[9] *= 1e-20; w2[11] *= 1e15
    h, l, inv = split_f16x2(w2)
    assert torch.isfinite(h.float()).all() and torch.isfinite(l.float()).all() and float(h.float().abs().max()) <= 32768.0      # (a maximum just under 2^15 may round up to it)
    rec = inv.double()[:, None] * (h.double() + l.double() / 2048.0)
    big = w2.abs() >= w2.abs().amax(1, keepdim=True) * 2.0 ** -26
    assert float(((rec - w2.double()).abs() / w2.abs().double().clamp_min(1e-300))[big].max()) <= 2.0 ** -22
    assert float(((rec - w2.double()).abs() / w2.abs().amax(1, keepdim=True).double().clamp_min(1e-300)).max()) <= 2.0 ** -22 and torch.equal(rec[7], torch.zeros(64, dtype=torch.float64))
    relerr = ((rec - w2.double()).abs() / w2.abs().double().clamp_min(1e-300))[big]
    assert float(relerr.pow(2).mean().sqrt()) <= 2.0 ** -24
    lg = torch.log2(inv); assert torch.equal(lg, lg.round())                      # powers of two
    p3 = pack_conv1x1(w2.reshape(256, 64, 1, 1), 3)
    assert p3.dtype == torch.int16 and tuple(p3.shape) == (2 * 256 * (64 + 1),)
    pl3 = p3[:2 * 256 * 64].reshape(8, 4, 2, 64, 8)
    for co, k in ((0, 0), (37, 9), (255, 63), (128, 16), (31, 8)):
        for pl, src in enumerate((h, l)):
            assert int(pl3[co // 32, k // 16, pl, 32 * ((k % 16) // 8) + co % 32, k % 8]) == int(src.view(torch.int16)[co, k])
    assert torch.equal(p3[2 * 256 * 64:].view(torch.float32), inv)


def test_pack_conv3x3_h_is_the_operand_order_of_the_direct_kernel():
    """pack_conv3x3_h (csrc/conv3x3h.hip): plane p of element (co, ci, dy, dx) at [co / 32][ci / 16][dy][dx][p][32 ((ci % 16) / 8) + co % 32][ci % 8], the planes being
    split_f16x2 of the output channel's cin x 9 weights; the inverse channel scales follow; shapes the kernel does not take give None."""
    from vido_slam_amd.nets.ops import pack_conv3x3_h, split_f16x2
    g = torch.Generator().manual_seed(3)
    cout, cin = 256, 48
    w = torch.randn(cout, cin, 3, 3, generator=g) * torch.exp(torch.randn(cout, 1, 1, 1, generator=g))
    p = pack_conv3x3_h(w)
    assert p.dtype == torch.int16 and tuple(p.shape) == (2 * cout * (cin * 9 + 1),)
    h, l, inv = split_f16x2(w.reshape(cout, cin * 9))
    h = h.view(torch.int16).reshape(cout, cin, 3, 3); l = l.view(torch.int16).reshape(cout, cin, 3, 3)
    pl = p[:2 * cout * cin * 9].reshape(cout // 32, cin // 16, 3, 3, 2, 64, 8)
    for _ in range(400):
        co, ci, dy, dx = (int(torch.randint(0, n, (1,), generator=g)) for n in (cout, cin, 3, 3))
        lane = 32 * ((ci % 16) // 8) + co % 32
        assert int(pl[co // 32, ci // 16, dy, dx, 0, lane, ci % 8]) == int(h[co, ci, dy, dx]) and int(pl[co // 32, ci // 16, dy, dx, 1, lane, ci % 8]) == int(l[co, ci, dy, dx])
    assert torch.equal(p[2 * cout * cin * 9:].view(torch.float32), inv)
    rec = inv.double()[:, None] * (h.view(torch.float16).reshape(cout, -1).double() + l.view(torch.float16).reshape(cout, -1).double() / 2048.0)
    assert float(((rec - w.reshape(cout, -1).double()).abs() / w.reshape(cout, -1).abs().amax(1, keepdim=True).double()).max()) <= 2.0 ** -22
    assert pack_conv3x3_h(torch.zeros(96, 32, 3, 3)) is None and pack_conv3x3_h(torch.zeros(128, 32, 1, 1)) is None and tuple(pack_conv3x3_h(torch.zeros(64, 32, 3, 3)).shape) == (2 * 64 * (32 * 9 + 1),)
    p24 = pack_conv3x3_h(torch.ones(128, 24, 3, 3))                           # 24 input channels: padded to 32 with zero weights
    assert tuple(p24.shape) == (2 * 128 * (32 * 9 + 1),) and int((p24[:2 * 128 * 32 * 9].reshape(4, 2, 3, 3, 2, 64, 8)[:, 1, :, :, :, 32:, :] != 0).sum()) == 0


def test_conv1x1_tile_form_is_chosen_by_rounds_of_workgroups():
    """vido_conv1x1_layout (host side of csrc/conv1x1.hip).  Default: 128 x 128 tiles (the form that measured faster on the pipelined headline).  VIDO_CONV1X1_TN=0: 128 x 112
    tiles where they need fewer (rounds of 256 CUs) x (tile width) — every bottleneck shape of X-101-32x8d at the 800 x 1088 feed (850 tiles of 128 x 128 = 3.3 rounds -> 972
    of 128 x 112 = 3.8 rounds, each 7 / 8 of the work); 128 x 128 where the 112-wide form would add a round, or when the input channels are not a multiple of 64.  The switch
    is read once per process: the rule is asked in a child process."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); from vido_slam_amd.host import load_library; lib = load_library(); "
            "print([lib.vido_conv1x1_layout(*a) for a in ((256, 256, 200 * 272), (512, 512, 100 * 136), (1024, 1024, 50 * 68), (96, 128, 4096), (256, 256, 128 * 128 * 2))])" % ROOT)
    # (round 6: without any switch the split-fp16 form = layout 3 takes every shape, VIDO_CONV1X1_ARITH=bf16x3 the split-bf16 form = layout 2; VIDO_CONV1X1_ARITH=f32 or a
    # forced tile width bring the fp32-instruction forms back)
    for tn, want in (("0", [1, 1, 1, 0, 0]), ("128", [0, 0, 0, 0, 0]), ("112", [1, 1, 1, 0, 1]), (None, [3, 3, 3, 3, 3]), ("f32", [0, 0, 0, 0, 0]), ("bf16x3", [2, 2, 2, 2, 2])):
        env = {k: v for k, v in os.environ.items() if k not in ("VIDO_CONV1X1_TN", "VIDO_CONV1X1_ARITH")}
        if tn in ("f32", "bf16x3"):
            env["VIDO_CONV1X1_ARITH"] = tn
        elif tn is not None:
            env["VIDO_CONV1X1_TN"] = tn
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == str(want), (tn, out.stdout, out.stderr[-500:])


def test_fc_h_split_rule_answers_without_a_gpu():
    """vido_fc_h_splitk (host side of csrc/fch.hip): the smallest K split of 1, 2, 4, 8 that gives >= 256 workgroups of 128 x 128 outputs — bounded by divisibility (an even number
    of 16-feature steps per split) — and 0 for shapes the kernel does not take."""
    from vido_slam_amd.host import load_library
    lib = load_library()
    assert lib.vido_fc_h_splitk(1000, 12544, 1024) == 4 and lib.vido_fc_h_splitk(1000, 1024, 1024) == 4 and lib.vido_fc_h_splitk(4000, 1024, 1024) == 1
    assert lib.vido_fc_h_splitk(100, 12544, 1024) == 8 and lib.vido_fc_h_splitk(1000, 96, 128) == 1 and lib.vido_fc_h_splitk(1000, 64, 128) == 2
    assert lib.vido_fc_h_splitk(1000, 12544, 1000) == 0 and lib.vido_fc_h_splitk(1000, 40, 128) == 0 and lib.vido_fc_h_splitk(0, 64, 128) == 0 and lib.vido_fc_h_splitk(1000, 64, 64) == 0


def test_strided_grouped_conv_plan_answers_without_a_gpu():
    """vido_gconv3x3_s2_supported (host side of csrc/gconv.hip::k_gconv3x3_s2_m32 / _m16): the detector's three strided conv2 shapes have a kernel; a width that is not a
    multiple of 4, 24 channels per group, and a band that does not fit two LDS buffers are refused (the caller keeps the library convolution)."""
    from vido_slam_amd.host import load_library
    lib = load_library()
    ok = lambda H, W, ci, co: bool(lib.vido_gconv3x3_s2_supported(H, W, ci, co))
    assert ok(200, 272, 16, 16) and ok(100, 136, 32, 32) and ok(50, 68, 64, 64) and ok(2, 4, 32, 32) and ok(31, 40, 8, 8) and ok(100, 136, 32, 64)
    assert not ok(50, 70, 32, 32) and not ok(50, 68, 24, 24) and not ok(40, 300, 32, 32) and not ok(50, 68, 12, 32) and not ok(1, 8, 32, 32)


def test_strided_grouped_conv_band_addressing_reproduces_the_convolution():
    """csrc/gconv.hip::k_gconv3x3_s2_*: a numpy walk of the kernel's data path with the library's own geometry (vido_debug_gs2_plan).  Per position chunk the band is filled the way
    the copy instructions fill it — 16-byte slots in flattened (row, quad) order, even input rows first, then the odd ones, a zero quad in front of every row — and output position
    q = yo * Wop + xo reads tap (dy, dx) at 2 (q - r0 Wop) + dx + 3 in the even plane (dy = 1), the odd plane (dy = 0) or one row further down it (dy = 2).  Must be
    conv2d(stride 2, padding 1) for even and odd heights, a 2 x 4 map and the detector's widths."""
    import ctypes as C
    import torch.nn.functional as F
    from vido_slam_amd.host import load_library
    lib = load_library()
    g = torch.Generator().manual_seed(5)
    for H, W, cpg in ((10, 12, 32), (7, 8, 32), (2, 4, 32), (9, 136, 32), (6, 272, 16), (5, 20, 8)):
        out = (C.c_int * 9)()
        assert lib.vido_debug_gs2_plan(H, W, 8, cpg, out) == 1
        Ho, Wo, Wop, PL, NE, PS, gx, nj, P = list(out)
        assert Ho == (H + 1) // 2 and Wo == W // 2 and PL == W + 4 and Wop == PL // 2 and gx == -(-(Ho * Wop) // P) and 256 * nj <= PS and (2 * NE + 1) * PL + 4 <= PS
        x = torch.randn(1, 1, H, W, generator=g).double(); w = torch.randn(1, 1, 3, 3, generator=g).double()
        ref = F.conv2d(x, w, None, 2, 1)[0, 0].numpy(); xi = x[0, 0].numpy(); wk = w[0, 0].numpy()
        got = np.zeros((Ho, Wo))
        for chunk in range(gx):
            q0 = chunk * P; r0 = q0 // Wop
            plane = np.full(PS, np.nan)                                                   # (a read of something the copies did not write would poison the result)
            for s in range(64 * nj):                                                      # one 16-byte slot per lane and copy instruction
                rr, xq = divmod(s, PL // 4)
                row = 2 * (r0 + rr) if rr < NE else 2 * (r0 + rr - NE) - 1
                ok = rr < 2 * NE + 1 and xq >= 1 and 0 <= row < H
                plane[4 * s:4 * s + 4] = xi[row, 4 * (xq - 1):4 * (xq - 1) + 4] if ok else 0.0
            for q in range(q0, min(q0 + P, Ho * Wop)):
                yo, xo = divmod(q, Wop)
                if xo >= Wo:
                    continue
                base = 2 * (q - r0 * Wop) + 3
                acc = 0.0
                for dy in range(3):
                    off = (NE * PL, 0, NE * PL + PL)[dy]                                   # odd plane, even plane, odd plane one row down
                    for dx in range(3):
                        acc += wk[dy, dx] * plane[off + base + dx]
                got[yo, xo] = acc
        assert np.allclose(got, ref, rtol=0, atol=1e-12), (H, W, cpg, float(np.abs(got - ref).max()))


def test_grouped_conv_buffer_copy_band_addressing_reproduces_the_convolution():
    """csrc/gconv.hip::k_gconv3x3_m16d (8 / 16 channels per group, stride 1): the same walk — a channel plane is a run of 16-byte slots in flattened (row, quad) order with a zero
    quad in front of every row, filled by nj copy instructions of 64 slots; output position q = y * (W + 4) + x reads tap (dy, dx) at (q - r0 (W + 4)) + dy (W + 4) + dx + 3."""
    import ctypes as C
    import torch.nn.functional as F
    from vido_slam_amd.host import load_library
    lib = load_library()
    g = torch.Generator().manual_seed(6)
    for H, W, cpg in ((9, 12, 16), (3, 4, 8), (7, 272, 8), (12, 136, 16), (40, 8, 16)):
        out = (C.c_int * 6)()
        assert lib.vido_debug_gc16_plan(H, W, 8, cpg, out) == 1
        R, PS, nj, gx, P, Wpd = list(out)
        assert Wpd == W + 4 and gx == -(-(H * Wpd) // P) and 256 * nj <= PS and R * Wpd + 4 <= PS
        x = torch.randn(1, 1, H, W, generator=g).double(); w = torch.randn(1, 1, 3, 3, generator=g).double()
        ref = F.conv2d(x, w, None, 1, 1)[0, 0].numpy(); xi = x[0, 0].numpy(); wk = w[0, 0].numpy()
        got = np.zeros((H, W))
        for chunk in range(gx):
            q0 = chunk * P; r0 = q0 // Wpd
            plane = np.full(PS, np.nan)
            for s in range(64 * nj):
                rr, xq = divmod(s, Wpd // 4)
                row = r0 - 1 + rr
                ok = rr < R and xq >= 1 and 0 <= row < H
                plane[4 * s:4 * s + 4] = xi[row, 4 * (xq - 1):4 * (xq - 1) + 4] if ok else 0.0
            for q in range(q0, min(q0 + P, H * Wpd)):
                yy, xx = divmod(q, Wpd)
                if xx >= W:
                    continue
                base = (q - r0 * Wpd) + 3
                got[yy, xx] = sum(wk[dy, dx] * plane[base + dy * Wpd + dx] for dy in range(3) for dx in range(3))
        assert np.allclose(got, ref, rtol=0, atol=1e-12), (H, W, cpg, float(np.abs(got - ref).max()))
    assert lib.vido_debug_gc16_plan(9, 13, 8, 16, (C.c_int * 6)()) == 0                  # W not a multiple of 4: the round-3 kernel keeps the shape


def test_pack_wino3x3_operands_reproduce_the_convolution():
    """vido_wino3x3_pack (host side of csrc/wino.hip): U = G g G^T in the kernel's operand order.  A numpy walk of the kernel's own data path — V = B^T d B of the zero-padded
    4x4 windows, M_p[co][tile] = sum_c U_p[co][c] V_p[c][tile] with U_p[co][c] read from [co / 32][c / KC][p][32 * (c & 1) + co % 32][(c % KC) / 2], Y = A^T M A — must be
    the padded 3x3 cross-correlation (float64 reference), for both chunk sizes (KC = 8: 64-channel workgroups; KC = 4: 32-channel ones), ragged channel counts and odd maps."""
    import torch.nn.functional as F
    from vido_slam_amd.nets.ops import pack_wino3x3
    g = torch.Generator().manual_seed(11)
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    for cout, cin, H, W, form in ((64, 9, 6, 8, 0), (32, 13, 5, 7, 0), (96, 8, 4, 4, 0), (130, 17, 3, 6, 0), (64, 9, 6, 8, 1), (130, 17, 3, 6, 1)):      # form 1 (K-split launches): KC = 4 for every cout
        w = torch.randn(cout, cin, 3, 3, generator=g); x = torch.randn(1, cin, H, W, generator=g)
        up = pack_wino3x3(w, form)
        kc = 8 if (((cout + 63) // 64) * 64 - cout) < 32 and form == 0 else 4
        assert up.shape[1:] == (-(-cin // kc), 16, 64, kc // 2) and up.shape[0] * 32 >= cout and up.is_contiguous()
        U = np.zeros((16, up.shape[0] * 32, up.shape[1] * kc))
        for co in range(U.shape[1]):
            for c in range(U.shape[2]):
                U[:, co, c] = up[co // 32, c // kc, :, 32 * (c & 1) + co % 32, (c % kc) // 2].numpy()
        assert np.all(U[:, cout:] == 0) and np.all(U[:, :, cin:] == 0)                      # padded channels carry zero weights
        Ht, Wt = (H + 1) // 2, (W + 1) // 2
        xp = np.zeros((U.shape[2], 2 * Ht + 2, 2 * Wt + 2)); xp[:cin, 1:H + 1, 1:W + 1] = x[0].double().numpy()
        y = np.zeros((cout, 2 * Ht, 2 * Wt))
        for ty in range(Ht):
            for tx in range(Wt):
                d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
                V = np.einsum("ij,cjk,lk->cil", Bt, d, Bt).reshape(-1, 16)                   # [c][p]
                M = np.einsum("poc,cp->op", U, V)[:cout].reshape(cout, 4, 4)
                y[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ij,ojk,lk->oil", At, M, At)
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)[0].numpy()
        assert np.abs(y[:, :H, :W] - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())


def test_pack_conv_direct_operands_reproduce_the_convolution():
    """vido_conv_direct_pack (host side of csrc/convdirect.hip): element (co, c, tap) at [co / 32][tap][c / 2][32 * (c & 1) + co % 32], channel pairs of a tap padded to the
    kernel's step (2 pairs for 25 / 49 taps, 4 otherwise), channel blocks to an even count.  A numpy walk of the kernel's data path — per (tap, channel pair) a 32 x 2
    weight operand times the 2 x pixels patch rows, zero padding outside the image — must be conv2d (float64 reference), for strides, odd channel counts, ragged sizes."""
    import ctypes as C
    import torch.nn.functional as F
    from vido_slam_amd.nets.ops import pack_conv_direct
    from vido_slam_amd.host import load_library
    lib = load_library()
    lib.vido_conv_direct_packed_floats.restype = C.c_longlong
    g = torch.Generator().manual_seed(5)
    for cout, cin, H, W, k, s in ((32, 3, 9, 11, (7, 7), (1, 1)), (49, 32, 6, 8, (7, 1), (1, 1)), (70, 7, 7, 9, (3, 3), (2, 2)), (9, 32, 5, 6, (3, 3), (1, 1)), (1, 9, 4, 5, (1, 1), (1, 1)), (25, 25, 5, 7, (1, 5), (1, 1))):
        kh, kw = k; ph, pw = kh // 2, kw // 2; T = kh * kw
        w = torch.randn(cout, cin, kh, kw, generator=g); x = torch.randn(1, cin, H, W, generator=g)
        wp = pack_conv_direct(w).numpy()
        U = 2 if T >= 25 else 4
        cpr = (cin + 1) // 2; cpp = -(-cpr // U) * U; cb = -(-cout // 32); cbp = 1 if cb == 1 else -(-cb // 2) * 2
        assert wp.size == int(lib.vido_conv_direct_packed_floats(cin, cout, kh, kw)) == 64 * cbp * T * cpp
        wp = wp.reshape(cbp, T, cpp, 2, 32)                                                    # [block][tap][pair][channel parity][co % 32]
        assert lib.vido_conv_direct_supported(cin, cout, H, W, kh, kw, s[0], s[1], ph, pw) == 1
        Ho, Wo = (H + 2 * ph - kh) // s[0] + 1, (W + 2 * pw - kw) // s[1] + 1
        xp = np.zeros((2 * cpp, H + 2 * ph, W + 2 * pw)); xp[:cin, ph:ph + H, pw:pw + W] = x[0].double().numpy()
        y = np.zeros((cbp * 32, Ho, Wo))
        for t in range(T):
            ky, kx = divmod(t, kw)
            patch = xp[:, ky:ky + s[0] * Ho:s[0], kx:kx + s[1] * Wo:s[1]].reshape(cpp, 2, Ho, Wo)     # [pair][parity][oy][ox]
            y += np.einsum("bpqo,pqyx->boyx", wp[:, t].astype(np.float64), patch).reshape(cbp * 32, Ho, Wo)
        ref = F.conv2d(x.double(), w.double(), None, s, (ph, pw))[0].numpy()
        assert np.all(y[cout:] == 0) and np.abs(y[:cout] - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), (cout, cin, k, s)
    assert lib.vido_conv_direct_supported(8, 8, 8, 8, 3, 5, 1, 1, 1, 2) == 0 and lib.vido_conv_direct_packed_floats(8, 8, 3, 5) == 0
