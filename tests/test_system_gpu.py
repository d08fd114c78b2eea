"""GPU: the C handle over VIDO_SLAM::System (vido_system_*, include/vido_c.h) through its ctypes mirror (vido_slam_amd/system.py), the
device-side cvtColor ingest (row A2, Tracking.cc:327-340) and the pipelined end-to-end chain (pipeline.EndToEnd: run_vido.cc:142-157 ->
:229-235)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _settings(tmp, scene, rgb=0):
    fx, fy, cx, cy = scene.K
    p = os.path.join(str(tmp), "settings.yaml")
    with open(p, "w") as fh:
        fh.write("%%YAML:1.0\nCamera.width: %d\nCamera.height: %d\n" % (scene.w, scene.h))
        fh.write("Camera.fx: %r\nCamera.fy: %r\nCamera.cx: %r\nCamera.cy: %r\nCamera.k1: 0.0\nCamera.k2: 0.0\nCamera.p1: 0.0\nCamera.p2: 0.0\nCamera.k3: 0.0\n" % (fx, fy, cx, cy))
        fh.write("Camera.bf: 387.57\nCamera.fps: 10.0\nCamera.RGB: %d\nChooseData: 1\nDepthMapFactor: 1.0\nThDepthBG: 40.0\nThDepthOBJ: 25.0\n" % rgb)
        fh.write("MaxTrackPointBG: 3000\nMaxTrackPointOBJ: 800\nSFMgThres: 0.12\nSFDsThres: 0.3\nWINDOW_SIZE: 20\nOVERLAP_SIZE: 4\nUseSampleFeature: 0\n")
        fh.write("ORBextractor.nFeatures: 2000\nORBextractor.scaleFactor: 1.2\nORBextractor.nLevels: 8\nORBextractor.iniThFAST: 20\nORBextractor.minThFAST: 7\n")
    return p


@pytest.mark.parametrize("cn,rgb", [(3, False), (3, True), (4, False), (4, True)])
def test_color_ingest_bit_exact_vs_oracle(vido, oracle, cn, rgb):
    """k_ingest_color == vo_bgr2gray (cvtColor's 14-bit fixed point) for BGR / RGB / BGRA / RGBA, odd widths included; the keypoints equal
    those of the gray path on the converted image."""
    for (w, h) in ((640, 480), (333, 250)):
        rng = np.random.RandomState(7 + cn + w)
        base = vido.synth.make_frame(w, h, seed=11)
        img = np.stack([np.roll(base, k, axis=k % 2) for k in range(cn)], -1).astype(np.uint8)
        img[..., 1] = (img[..., 1].astype(np.int32) * 3 // 4 + rng.randint(0, 40, (h, w))).astype(np.uint8)
        ctx = vido.Context(width=w, height=h, max_batch=1)
        gray, kps, desc = ctx.orb_extract_color(img, rgb_order=rgb)
        ref = oracle.bgr2gray(img, rgb_order=rgb)
        assert np.array_equal(gray, ref)
        assert np.array_equal(ctx.orb_level(0, 0), ref)
        k2, d2 = ctx.orb_extract(ref)
        assert len(kps) == len(k2) and len(kps) > 200
        for f in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(kps[f], k2[f]), f
        assert np.array_equal(desc, d2)
        ctx.close()


def test_system_c_handle_tracks_bgr_clip(tmp_path, vido):
    """System.Init / TrackRGBD through the C handle with BGR frames (cvtColor on the device): poses follow the renderer's ground truth, the depth
    buffer is pre-scaled in place, stats are filled."""
    from vido_slam_amd.system import System
    n = 8
    scene = vido.synth.Scene3D(n_frames=n, seed=3, objects=((-2.0, 0.2, 9.0, 0.25, 0.0, 0.05),))
    slam = System(); slam.Init(_settings(tmp_path, scene), System.RGBD)
    keep = []
    for k in range(n):
        g, d, f, m = scene.frame(k)
        bgr = vido.synth.gray_to_bgr(g)
        d = np.ascontiguousarray(d, np.float32); d0 = d.copy()
        T = slam.TrackRGBD(bgr, d, np.ascontiguousarray(f, np.float32), np.ascontiguousarray(m, np.int32), None, None, float(k), None, n)
        keep.append((bgr, d, f, m))
        assert np.array_equal(d, np.where(d0 < 0, 0, d0))                   # ChooseData 1 (OMD): d / 1.0, negatives -> 0 (Tracking.cc:299-322)
        E = T.astype(np.float64) @ np.linalg.inv(scene.Tcw(k))
        assert np.linalg.norm(E[:3, 3]) < 0.05, (k, E)
        st = slam.stats()
        assert st["frame_id"] == k and st["n_keypoints"] > 500
        if k >= 2:
            assert st["n_static_inliers"] > 300 and st["ms_total"] > 0 and st["ms_local_ba"] > 0
    assert st["n_objects"] == 1
    slam.SaveResultsIJRR2020(os.path.join(str(tmp_path), "res_"))
    assert np.loadtxt(os.path.join(str(tmp_path), "res_refined_rgbd_new.txt")).shape == (n, 17)
    slam.close()


def test_system_errors_come_back_as_codes(tmp_path, vido):
    from vido_slam_amd.system import System
    s = System()
    with pytest.raises(vido.VidoError):
        s.Init(os.path.join(str(tmp_path), "missing.yaml"))
    with pytest.raises(vido.VidoError):
        s.TrackRGBD(np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.float32), np.zeros((8, 8, 2), np.float32), np.zeros((8, 8), np.int32))
