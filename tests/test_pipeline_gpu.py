"""SURVEY.md 8f row 4 on the GPU: (1) the in-process network -> tracker hand-off gives the same front-end lists as the round trip through
host arrays; (2) a `.g2o` graph dumped by the facade's FullBatchOptimization is read back, solved through the C-ABI, and lands on the
optimised graph the facade dumped."""
import os, subprocess, sys
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_net_frontend_matches_host_round_trip(vido):
    from vido_slam_amd import nets, synth, pipeline
    from test_maskrcnn_gpu import TINY
    w, h = 640, 256
    ctx = vido.Context(width=w, height=h, max_batch=1, n_levels=5)       # a short frame like the demo's 640x192: the top pyramid levels would be smaller than one FAST cell row
    hops = nets.HipOps(ctx)
    flow_net = nets.fill_deterministic(nets.LiteFlowNet(hops.correlation, epilogue=hops.bias_act_), 5).eval().cuda()
    depth_net = nets.fill_deterministic(nets.MonoDepth2(), 9).eval().cuda()
    mask_net = nets.fill_maskrcnn(nets.MaskRCNN(hops, TINY), 3).eval().cuda()
    p = vido.track_params(dataset=2, depth_map_factor=256.0, bf=387.57, kaist_scale=1.2, th_depth_bg=80.0, th_depth_obj=60.0)
    fe = pipeline.NetFrontEnd(ctx, p, flow_net, depth_net, mask_net, mask_feed=(256, 640), confidence=0.05, keep_raw=True)
    seq = synth.Sequence(n_frames=4, w=w, h=h, seed=6, n_obj=2)
    frames = [seq.frame(k)[1] for k in range(3)]                  # BGR u8
    assert fe.push(frames[0]) is None
    for k in (1, 2):
        out = fe.push(frames[k])
        n_kp, n_stat, n_obj = int(out["n_kp"][0]), int(out["n_stat"][0]), int(out["n_obj"][0])
        got = {name: np.array(out[name][0]) for name in ("kps", "desc", "stat_idx", "stat_corr", "stat_flow", "stat_depth", "obj_keys", "obj_corr", "obj_depth", "obj_label", "obj_flow")}
        # the same network outputs through host arrays -> a second context's front end (copies, no aliasing).  (Re-running the networks
        # instead would not do: MIOpen's convolutions are not bit-reproducible from call to call.)
        flow, depth, mask = fe.raw
        assert flow.shape == (h, w, 2) and depth.dtype == torch.float32 and mask.dtype == torch.int32 and float(depth.max()) > 1000.0      # MONO16 range
        gray = pipeline.bgr_to_gray(torch.as_tensor(frames[k]).cuda()).cpu().numpy()
        b, g, r = (frames[k][..., c].astype(np.int64) for c in range(3))
        assert np.array_equal(gray, ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8))
        ctx2 = vido.Context(width=w, height=h, max_batch=1, n_levels=5)
        ff2 = vido.FrameFeatures(ctx2, p)
        ref = ff2.frontend_batch(0, gray[None], depth.cpu().numpy()[None].copy(), flow.cpu().numpy()[None], mask.cpu().numpy()[None])
        assert int(ref["n_kp"][0]) == n_kp and int(ref["n_stat"][0]) == n_stat and int(ref["n_obj"][0]) == n_obj and n_kp > 200
        for name, n in (("kps", n_kp), ("desc", n_kp), ("stat_idx", n_stat), ("stat_corr", n_stat), ("stat_flow", n_stat), ("stat_depth", n_stat),
                        ("obj_keys", n_obj), ("obj_corr", n_obj), ("obj_depth", n_obj), ("obj_label", n_obj), ("obj_flow", n_obj)):
            assert np.array_equal(got[name][:n], np.array(ref[name][0])[:n]), (k, name)
        # the slot the frame went to aliases the network outputs: the depth map read back is the pre-scaled MONO16 depth
        d_slot, f_slot, m_slot = fe.ff.read_maps(out["slot"])
        assert np.array_equal(f_slot, flow.cpu().numpy()) and np.array_equal(m_slot, mask.cpu().numpy())
        assert np.array_equal(d_slot, ff2.read_maps(0)[0])
        ctx2.close()
    ctx.close()


def test_g2o_import_reproduces_the_facade_batch(tmp_path, vido):
    sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import build
    from test_facade_gpu import write_clip
    from vido_slam_amd import g2o_io
    n = 10
    scene = vido.synth.Scene3D(n_frames=n, seed=3, objects=((-2.0, 0.2, 9.0, 0.25, 0.0, 0.05),))
    cfg = write_clip(str(tmp_path), scene, n, dataset=2, factor=256.0)
    r = subprocess.run([build.build_driver(), cfg, os.path.join(str(tmp_path), "poses.txt"), os.path.join(str(tmp_path), "res_")], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, VIDO_DUMP_G2O=str(tmp_path)))
    assert r.returncode == 0, r.stderr + r.stdout
    before, dy, ids = g2o_io.read_g2o(os.path.join(str(tmp_path), "dynamic_slam_graph_before_opt.g2o"))
    after, dy_after, ids_after = g2o_io.read_g2o(os.path.join(str(tmp_path), "dynamic_slam_graph_after_opt.g2o"))
    assert ids == ids_after and before["n_cam"] == n and dy is not None and dy["n_H"] > 0 and dy["n_tern"] > 100
    assert np.isclose(before["info_obs"], 1.0 / np.float32(80.0), rtol=1e-6) and np.isclose(dy["info_tern"], 1.0 / np.float32(100.0), rtol=1e-6)
    assert np.isclose(before["info_odo"], 1.0 / np.float32(0.0001), rtol=1e-6) and np.isclose(dy["info_smooth"], 1.0 / np.float32(0.001), rtol=1e-6)
    ctx = vido.Context()
    res = vido.ba_optimize(ctx, before, dynamic=dy)
    assert res["chi2_final"] < res["chi2_initial"]
    # same graph (9 significant digits of it), same solver: the imported solve ends where the facade's own batch ended
    assert np.abs(res["cam_T"].reshape(-1, 12) - np.asarray(after["cam_T"]).reshape(-1, 12)).max() < 2e-4
    assert np.abs(res["pt_xyz"] - np.asarray(after["pt_xyz"])).max() < 2e-3
    assert np.abs(res["H_T"].reshape(-1, 12) - np.asarray(dy_after["H_T"]).reshape(-1, 12)).max() < 2e-3
    ctx.close()
