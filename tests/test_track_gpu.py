"""GPU parity: tracking front-end data-parallel stages (C-ABI) vs the CPU oracle (oracle/track_oracle.c,
restating Tracking.cc:299-322, 369-421, 1582-1668, 3291-3357 and Frame.cc:72-211, 706-771). Bit-exact."""
import numpy as np
import pytest
import torch   # before the first Context: torch bundles its own HIP runtime and must be the one the process initialises

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(vido):
    from vido_slam_amd import synth
    ctx = vido.Context(width=640, height=480, max_batch=3)
    seq = synth.Sequence(n_frames=4, seed=2, depth_map_factor=1.0)
    frames = [seq.frame(k) for k in range(3)]
    return ctx, frames


@pytest.mark.parametrize("dataset", [0, 1, 2])
def test_upload_prescale_and_lists(vido, oracle, setup, dataset):
    ctx, frames = setup
    p = vido.track_params(dataset=dataset, depth_map_factor=5.0 if dataset else 1.0, bf=387.57, kaist_scale=1.2,
                          th_depth_bg=40.0, th_depth_obj=25.0)
    ff = vido.FrameFeatures(ctx, p)
    gray = np.stack([f[0] for f in frames]); flow = np.stack([f[3] for f in frames]); mask = np.stack([f[4] for f in frames])
    depth_raw = np.stack([f[2] for f in frames]).astype(np.float32)
    if dataset:          # disparity-like raw input so that bf/(d/f) lands in metric range
        depth_raw = (387.57 * 5.0 / np.maximum(depth_raw, 0.5)).astype(np.float32)
    depth_raw[0, 5, 5] = -3.0         # negative -> 0
    depth = depth_raw.copy()
    ff.upload(0, depth, flow, mask)
    ref_depth = np.stack([oracle.depth_prescale(d, dataset, p.depth_map_factor, p.bf, p.kaist_scale) for d in depth_raw])
    assert np.array_equal(depth, ref_depth)           # caller's buffer mutated in place, bit-exact
    assert np.array_equal(ff.read_maps(1)[0], ref_depth[1])
    kps, desc, cnt = ctx.orb_extract_batch(gray, want_desc=False)
    out = ff.features(0, kps, cnt)
    for f in range(3):
        i, c, fl, dd = oracle.static_candidates(kps[f, :cnt[f]], ref_depth[f], flow[f], mask[f], p.th_depth_bg)
        n = out["n_stat"][f]
        assert n == len(i) and n > 50
        assert np.array_equal(out["stat_idx"][f, :n], i) and np.array_equal(out["stat_corr"][f, :n], c)
        assert np.array_equal(out["stat_flow"][f, :n], fl) and np.array_equal(out["stat_depth"][f, :n], dd)
        k, c, od, lab, fl = oracle.dense_object_samples(ref_depth[f], flow[f], mask[f], p.th_depth_obj)
        n = out["n_obj"][f]
        assert n == len(k) and n > 100
        assert np.array_equal(out["obj_keys"][f, :n], k) and np.array_equal(out["obj_corr"][f, :n], c)
        assert np.array_equal(out["obj_depth"][f, :n], od) and np.array_equal(out["obj_label"][f, :n], lab)
        assert np.array_equal(out["obj_flow"][f, :n], fl)


def test_gathers_update_mask_unproject_sceneflow(vido, oracle, setup):
    ctx, frames = setup
    p = vido.track_params(dataset=0, th_depth_bg=40.0, th_depth_obj=25.0, fx=520.0, fy=515.0, cx=319.5, cy=239.5)
    ff = vido.FrameFeatures(ctx, p)
    flow = np.stack([f[3] for f in frames]); mask = np.stack([f[4] for f in frames]); depth = np.stack([f[2] for f in frames]).astype(np.float32)
    mask_cur_lost = mask[1].copy(); mask_cur_lost[mask_cur_lost == 2] = 0          # Mask R-CNN "lost" object 2 in frame 1
    mask[1] = mask_cur_lost
    ff.upload(0, depth, flow, mask)
    k, c, od, lab, fl = oracle.dense_object_samples(depth[0], flow[0], mask[0], p.th_depth_obj)
    edge = np.array([[0.5, 10.0], [639.2, 5.0], [100.7, 479.9], [-3.0, 8.0]], np.float32)
    keys = np.concatenate([c, edge])
    assert np.array_equal(ff.gather_static_depth(1, keys), oracle.gather_static_depth(keys, depth[1]))
    gd, gl = ff.gather_object_depth_label(1, keys)
    rd, rl = oracle.gather_object_depth_label(keys, depth[1], mask[1], p.th_depth_obj)
    assert np.array_equal(gd, rd) and np.array_equal(gl, rl)
    # UpdateMask: label 2 must be recovered by scattering frame 0's mask through frame 0's flow
    rec = ff.update_mask(0, 1, lab, c)
    ref_mask, ref_rec = oracle.update_mask(lab, c, mask[0], flow[0], mask[1])
    assert list(rec) == list(ref_rec) == [2]
    assert np.array_equal(ff.read_maps(1)[2], ref_mask)
    # back-projection + scene flow
    th = 0.03
    Tcw = np.array([[np.cos(th), 0, np.sin(th), 0.3], [0, 1, 0, -0.1], [-np.sin(th), 0, np.cos(th), 1.5], [0, 0, 0, 1]], np.float32)
    z = od.copy(); z[::17] = -1.0
    xw = ff.unproject_world(k, z, Tcw)
    assert np.array_equal(xw, oracle.unproject_world(k, z, p.fx, p.fy, p.cx, p.cy, Tcw))
    xw2 = ff.unproject_world(c, od, np.eye(4, dtype=np.float32))
    sem_cur = lab.copy(); sem_cur[::5] = 0
    f3, ol = ff.scene_flow(xw, xw2, lab, sem_cur, np.full(len(lab), -2, np.int32))
    r3, rl = oracle.scene_flow(xw, xw2, lab, sem_cur, np.full(len(lab), -2, np.int32))
    assert np.array_equal(f3, r3) and np.array_equal(ol, rl)


def test_fused_frontend_equals_separate_calls(vido, setup):
    """vido_frontend_batch (ORB + pre-scale + lists in one stream, keypoints handed over on the device, results as a view
    of pinned memory) must reproduce the three separate entry points bit for bit — host inputs and device inputs."""
    ctx, frames = setup
    p = vido.track_params(dataset=0, depth_map_factor=1.0, th_depth_bg=40.0, th_depth_obj=25.0)
    ff = vido.FrameFeatures(ctx, p)
    gray = np.stack([f[0] for f in frames]); flow = np.stack([f[3] for f in frames]); mask = np.stack([f[4] for f in frames])
    depth0 = np.stack([f[2] for f in frames]).astype(np.float32)
    d1 = depth0.copy(); ff.upload(0, d1, flow, mask)
    kps, desc, cnt = ctx.orb_extract_batch(gray, want_desc=True)
    ref = {k: v.copy() for k, v in ff.features(0, kps, cnt).items()}
    def check(out, depth_after):
        assert np.array_equal(depth_after, d1)
        assert np.array_equal(out["n_kp"], cnt) and np.array_equal(out["n_stat"], ref["n_stat"]) and np.array_equal(out["n_obj"], ref["n_obj"])
        for f in range(3):
            n = cnt[f]
            assert np.array_equal(out["kps"][f, :n], kps[f, :n]) and np.array_equal(out["desc"][f, :n], desc[f, :n])
            ns, no = ref["n_stat"][f], ref["n_obj"][f]
            for k in ("stat_idx", "stat_corr", "stat_flow", "stat_depth"):
                assert np.array_equal(out[k][f, :ns], ref[k][f, :ns]), k
            for k in ("obj_keys", "obj_corr", "obj_depth", "obj_label", "obj_flow"):
                assert np.array_equal(out[k][f, :no], ref[k][f, :no]), k
    d2 = depth0.copy()
    check(ff.frontend_batch(0, gray, d2, flow, mask), d2)
    g = torch.from_numpy(gray).cuda(); dd = torch.from_numpy(depth0.copy()).cuda(); fl = torch.from_numpy(flow).cuda(); mk = torch.from_numpy(mask).cuda()
    out = ff.frontend_batch(0, (g.data_ptr(), 3, 480, 640, 480 * 640, 640), dd.data_ptr(), fl.data_ptr(), mk.data_ptr())
    torch.cuda.synchronize()
    check(out, dd.cpu().numpy())
    # zero-copy: the slots refer to the caller's device buffers; later slot readers (gathers) see them
    dd2 = torch.from_numpy(depth0.copy()).cuda()
    out = ff.frontend_batch(0, (g.data_ptr(), 3, 480, 640, 480 * 640, 640), dd2.data_ptr(), fl.data_ptr(), mk.data_ptr(), alias=True)
    torch.cuda.synchronize()
    check(out, dd2.cpu().numpy())
    assert np.array_equal(ff.read_maps(1)[0], d1[1])
    keys = np.array([[10.5, 20.5], [300.0, 200.0]], np.float32)
    assert np.array_equal(ff.gather_static_depth(2, keys), np.array([d1[2][20, 10], d1[2][200, 300]], np.float32))
