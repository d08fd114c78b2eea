"""GPU end-to-end: the C++ facade (VIDO_SLAM::System::TrackRGBD -> Tracking -> Frame -> Optimizer over the C-ABI) driven by
the offline driver (tools/run_vido_slam.cpp, counterpart of vido_slam/demo/run_vido_slam.cc) on a geometrically consistent
synthetic RGB-D + flow + mask clip (BASELINE.json configs[0] plumbing).  Checks the estimated
world->camera poses against the generator's ground truth."""
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_clip(tmp, scene, n, dataset=1, factor=1.0, bf=387.57, sample_features=0):
    for sub in ("image_0", "flow_image", "depth_image", "mask_image"):
        os.makedirs(os.path.join(tmp, sub), exist_ok=True)
    for k in range(n):
        g, d, f, m = scene.frame(k)
        g.tofile(os.path.join(tmp, "image_0", "%06d.gray" % k))
        with open(os.path.join(tmp, "flow_image", "%06d.flo" % k), "wb") as fh:
            np.array([202021.25], np.float32).tofile(fh); np.array([scene.w, scene.h], np.int32).tofile(fh); f.astype(np.float32).tofile(fh)
        if dataset == 2:                  # KITTI convention: the file holds a scaled disparity, depth = bf / (value / factor)  (Tracking.cc:305-309)
            d = (bf * factor / d.astype(np.float64))
        d.astype(np.float32).tofile(os.path.join(tmp, "depth_image", "%06d.depth" % k))
        m.astype(np.int32).tofile(os.path.join(tmp, "mask_image", "%06d.mask" % k))
    fx, fy, cx, cy = scene.K
    cfg = os.path.join(tmp, "config.yaml")
    with open(cfg, "w") as fh:
        fh.write("%%YAML:1.0\nimage_path: %s\nn_frames: %d\nCamera.width: %d\nCamera.height: %d\n" % (os.path.join(tmp, "image_0"), n, scene.w, scene.h))
        fh.write("Camera.fx: %r\nCamera.fy: %r\nCamera.cx: %r\nCamera.cy: %r\nCamera.k1: 0.0\nCamera.k2: 0.0\nCamera.p1: 0.0\nCamera.p2: 0.0\nCamera.k3: 0.0\n" % (fx, fy, cx, cy))
        fh.write("Camera.bf: %r\nCamera.fps: 10.0\nCamera.RGB: 0\nChooseData: %d\nDepthMapFactor: %r\nThDepthBG: 40.0\nThDepthOBJ: 25.0\n" % (bf, dataset, factor))
        fh.write("MaxTrackPointBG: 3000\nMaxTrackPointOBJ: 800\nSFMgThres: 0.12\nSFDsThres: 0.3\nWINDOW_SIZE: 20\nOVERLAP_SIZE: 4\nUseSampleFeature: %d\n" % sample_features)
        fh.write("ORBextractor.nFeatures: 2000\nORBextractor.scaleFactor: 1.2\nORBextractor.nLevels: 8\nORBextractor.iniThFAST: 20\nORBextractor.minThFAST: 7\n")
    return cfg


def test_offline_clip_recovers_ground_truth_poses(tmp_path, vido):
    sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd"))
    import build
    driver = build.build_driver()
    n = 10
    scene = vido.synth.Scene3D(n_frames=n, seed=3, objects=((-2.0, 0.2, 9.0, 0.25, 0.0, 0.05),))   # scene flow 0.255 m/frame > SFMgThres
    cfg = write_clip(str(tmp_path), scene, n)
    out = os.path.join(str(tmp_path), "poses.txt")
    r = subprocess.run([driver, cfg, out, os.path.join(str(tmp_path), "res_")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "incremental_equals_rebuild 1" in r.stdout, r.stdout              # Map::UpdateTracklets == GetStaticTrack / GetDynamicTrackNew rebuilds
    P = np.loadtxt(out)
    assert P.shape == (n, 17)
    assert np.allclose(P[0, 1:].reshape(4, 4), np.eye(4))                     # first frame = identity (Initialization)
    t_err, r_err = [], []
    for k in range(1, n):
        T = P[k, 1:].reshape(4, 4); G = scene.Tcw(k)
        E = T @ np.linalg.inv(G)
        t_err.append(np.linalg.norm(E[:3, 3])); r_err.append(np.degrees(np.arccos(np.clip((np.trace(E[:3, :3]) - 1) / 2, -1, 1))))
    assert max(t_err) < 0.05 and max(r_err) < 0.2, (t_err, r_err)            # 0.25 m / 0.4 deg per frame motion; exact flow & depth
    # refined trajectory + object motion files were written (SaveResultsIJRR2020 layout)
    ref = np.loadtxt(os.path.join(str(tmp_path), "res_refined_rgbd_new.txt"))
    assert ref.shape == (n, 17)
    mot = np.loadtxt(os.path.join(str(tmp_path), "res_obj_mot_rgbd_new.txt"), ndmin=2)
    assert mot.shape[1] == 18 and len(mot) >= n - 3
    # the square translates by (0.25, 0, 0.05) m per frame in the world: the estimated object motions must say so
    for row in mot:
        H = row[2:14].reshape(3, 4)
        assert np.abs(H[:, :3] - np.eye(3)).max() < 0.02 and np.abs(H[:, 3] - np.array([0.25, 0.0, 0.05])).max() < 0.05, row
    assert set(mot[:, 1].astype(int)) == {1}                                # one consistently tracked object id


def test_kitti_mode_runs_the_full_batch_with_object_factors(tmp_path, vido):
    """ChooseData = KITTI: at the last frame Tracking::Track calls Optimizer::FullBatchOptimization (Tracking.cc:1489-1498) —
    static landmarks + object motion vertices + dynamic point chains — and the refined results land in *_RF."""
    sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd"))
    import build
    driver = build.build_driver()
    n = 10
    scene = vido.synth.Scene3D(n_frames=n, seed=3, objects=((-2.0, 0.2, 9.0, 0.25, 0.0, 0.05),))
    cfg = write_clip(str(tmp_path), scene, n, dataset=2, factor=256.0)
    out = os.path.join(str(tmp_path), "poses.txt")
    r = subprocess.run([driver, cfg, out, os.path.join(str(tmp_path), "res_")], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, VIDO_DUMP_G2O=str(tmp_path)))
    assert r.returncode == 0, r.stderr + r.stdout
    # the graph dumps the reference writes (Optimizer.cc:1937-1939), in the vendored g2o's text format
    for name in ("dynamic_slam_graph_before_opt.g2o", "dynamic_slam_graph_after_opt.g2o"):
        lines = open(os.path.join(str(tmp_path), name)).read().splitlines()
        tags = [l.split()[0] for l in lines]
        assert tags[0] == "PARAMS_SE3OFFSET" and tags.count("EDGE_SE3_PRIOR") == 1
        assert tags.count("VERTEX_SE3:QUAT") >= n and tags.count("VERTEX_TRACKXYZ") > 100 and tags.count("EDGE_SE3_TRACKXYZ") > 300
        assert tags.count("EDGE_SE3_MOTION") > 10 and tags.count("EDGE_SE3:QUAT") >= n - 1
        ids = [int(l.split()[1]) for l in lines if l.startswith("VERTEX")]
        assert ids == list(range(1, len(ids) + 1))                                   # one running id in creation order
    ref = np.loadtxt(os.path.join(str(tmp_path), "res_refined_rgbd_new.txt"))
    assert ref.shape == (n, 17)
    for k in range(1, n):
        T = np.linalg.inv(ref[k, 1:].reshape(4, 4)); G = scene.Tcw(k)             # vmCameraPose_RF holds camera-to-world poses
        E = T @ np.linalg.inv(G)
        assert np.linalg.norm(E[:3, 3]) < 0.06, (k, E)
    mot = np.loadtxt(os.path.join(str(tmp_path), "res_obj_mot_refined.txt"), ndmin=2)
    assert len(mot) >= n - 3
    moved = [np.abs(row[2:14].reshape(3, 4) - np.eye(3, 4)).max() for row in mot]
    assert max(moved) > 1e-3                                                     # the motion vertices start at identity and were optimised


def write_kaist_clip(tmp, scene, n, bf=387.57, factor=100.0, dist=(0.0, 0.0, 0.0, 0.0, 0.0)):
    """The reference's on-disk layout (vido_slam/demo/run_vido_slam.cc:47-65, 113-122): vTimestampsImage.txt (header + integer nanoseconds),
    <image_path>/<first 19 chars of to_string(stamp)>.png Bayer-RG u8, ../flow_image/*.flo, ../depth_image/*.png u16 (KAIST: depth = bf / (value / DepthMapFactor)),
    ../mask_image/*.png u8."""
    from PIL import Image
    for sub in ("image_0", "flow_image", "depth_image", "mask_image"):
        os.makedirs(os.path.join(tmp, sub), exist_ok=True)
    stamps = [1544590798700000000 + 100000000 * k for k in range(n)]
    with open(os.path.join(tmp, "vTimestampsImage.txt"), "w") as fh:
        fh.write("#timestamp [ns]\n" + "".join("%d\n" % s for s in stamps))
    for k, s in enumerate(stamps):
        name = ("%d" % s)[:19]
        g, d, f, m = scene.frame(k)
        Image.fromarray(g, mode="L").save(os.path.join(tmp, "image_0", name + ".png"))          # a gray scene: its Bayer mosaic is the gray image itself
        with open(os.path.join(tmp, "flow_image", name + ".flo"), "wb") as fh:
            np.array([202021.25], np.float32).tofile(fh); np.array([scene.w, scene.h], np.int32).tofile(fh); f.astype(np.float32).tofile(fh)
        raw = np.clip(np.rint(bf * factor / d.astype(np.float64)), 1, 65535).astype(np.uint16)
        Image.fromarray(raw, mode="I;16").save(os.path.join(tmp, "depth_image", name + ".png"))
        Image.fromarray(m.astype(np.uint8), mode="L").save(os.path.join(tmp, "mask_image", name + ".png"))
    fx, fy, cx, cy = scene.K
    cfg = os.path.join(tmp, "config.yaml")
    with open(cfg, "w") as fh:
        fh.write("%%YAML:1.0\nimage_path: %s\nstart_index: 0\nslam_mode: 0\nCamera.width: %d\nCamera.height: %d\n" % (os.path.join(tmp, "image_0"), scene.w, scene.h))
        fh.write("Camera.fx: %r\nCamera.fy: %r\nCamera.cx: %r\nCamera.cy: %r\nCamera.k1: %r\nCamera.k2: %r\nCamera.p1: %r\nCamera.p2: %r\nCamera.k3: %r\n" % ((fx, fy, cx, cy) + tuple(dist)))
        fh.write("Camera.bf: %r\nCamera.fps: 10.0\nCamera.RGB: 0\nChooseData: 3\nDepthMapFactor: %r\nThDepthBG: 40.0\nThDepthOBJ: 25.0\n" % (bf, factor))
        fh.write("MaxTrackPointBG: 3000\nMaxTrackPointOBJ: 800\nSFMgThres: 0.12\nSFDsThres: 0.3\nWINDOW_SIZE: 20\nOVERLAP_SIZE: 4\nUseSampleFeature: 0\n")
        fh.write("ORBextractor.nFeatures: 2000\nORBextractor.scaleFactor: 1.2\nORBextractor.nLevels: 8\nORBextractor.iniThFAST: 20\nORBextractor.minThFAST: 7\n")
    return cfg


def test_reference_disk_layout_50_frame_clip(tmp_path, vido):
    """BASELINE configs[0]: a 50-frame clip in the REFERENCE's on-disk layout (time-stamp file, Bayer-RG / 16-bit / 8-bit PNGs, .flo) through the offline driver:
    PNG decoding (zlib inflate + scan-line filters), Bayer demosaic, the KAIST depth convention, the cvtColor ingest on the device, 50 x TrackRGBD with a full 20-frame
    local-BA window for the last 30 frames.  Poses follow the renderer's ground truth (depth is quantised to 16 bits like the dataset's)."""
    sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd"))
    import build
    driver = build.build_driver()
    n = 50
    scene = vido.synth.Scene3D(n_frames=n, seed=3, step=0.2, yaw_deg=0.1, objects=((-2.0, 0.4, 9.0, 0.02, 0.0, 0.22),), wall_z=48.0)
    cfg = write_kaist_clip(str(tmp_path), scene, n)
    out = os.path.join(str(tmp_path), "poses.txt")
    r = subprocess.run([driver, cfg, out, os.path.join(str(tmp_path), "res_")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "frames 50" in r.stdout and "incremental_equals_rebuild 1" in r.stdout, r.stdout
    P = np.loadtxt(out)
    assert P.shape == (n, 17)
    err, rpe = [], []
    for k in range(1, n):
        E = P[k, 1:].reshape(4, 4) @ np.linalg.inv(scene.Tcw(k))
        err.append(np.linalg.norm(E[:3, 3]))
        D = P[k, 1:].reshape(4, 4) @ np.linalg.inv(P[k - 1, 1:].reshape(4, 4)); Dg = scene.Tcw(k) @ np.linalg.inv(scene.Tcw(k - 1))
        rpe.append(np.linalg.norm((D @ np.linalg.inv(Dg))[:3, 3]))
    # frame-to-frame odometry without loop closure: the per-frame relative error (what the reference prints per frame, Tracking.cc:659-700) stays at the centimetre,
    # the accumulated drift within 3 % of the 9.8 m path (u16 disparity depth, 48 m background)
    assert max(rpe) < 0.03 and max(err) < 0.3, (max(rpe), max(err), np.mean(err))
    ref = np.loadtxt(os.path.join(str(tmp_path), "res_refined_rgbd_new.txt"))
    assert ref.shape == (n, 17)


def test_device_resident_ba_window_equals_the_map_walk(tmp_path, vido):
    """SURVEY 8(f) row 2, second half: PartialBatchOptimization on the device-resident window (csrc/bawin.hip: per frame only the new frame's feature rows and the tracklet
    label changes go up; the graph is assembled on the device) against the Map walk of rounds 1-2 (facade.cpp batch_optimize: Optimizer.cc:56-94, 276-350) on a 50-frame
    clip — the window slides for 30 frames and the ring (24 slots) wraps twice.  VIDO_BA_RESIDENT_CHECK=1 solves EVERY window both ways from the same Map state and counts
    disagreements (graph size, poses to 1e-7); separately the whole clip is run with the walk only and the trajectories compared."""
    sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd"))
    import build
    driver = build.build_driver()
    n = 50
    scene = vido.synth.Scene3D(n_frames=n, seed=3, step=0.2, yaw_deg=0.1, objects=((-2.0, 0.4, 9.0, 0.02, 0.0, 0.22),), wall_z=48.0)
    cfg = write_kaist_clip(str(tmp_path), scene, n)
    runs = {}
    for tag, env in (("check", {"VIDO_BA_RESIDENT_CHECK": "1"}), ("walk", {"VIDO_BA_HOST_WALK": "1"}), ("resident", {})):
        out = os.path.join(str(tmp_path), "poses_%s.txt" % tag)
        r = subprocess.run([driver, cfg, out, os.path.join(str(tmp_path), "res_%s_" % tag)], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr + r.stdout
        line = [l for l in r.stdout.splitlines() if l.startswith("resident_window")][0].split()
        runs[tag] = (np.loadtxt(out), np.loadtxt(os.path.join(str(tmp_path), "res_%s_initial_rgbd_new.txt" % tag)), int(line[2]), int(line[4]), r.stderr)
    assert runs["check"][2] >= n - 3 and runs["check"][3] == 0, runs["check"][4][-2000:]        # every window of the clip was cross-checked, none disagreed
    assert runs["walk"][2] == 0 and runs["resident"][2] == 0
    # same trajectory (TrackRGBD's return values) and same refined window poses (vmCameraPose -> initial_rgbd_new.txt) whichever way the windows were assembled
    for tag in ("check", "resident"):
        assert np.abs(runs[tag][0] - runs["walk"][0]).max() < 1e-4 and np.abs(runs[tag][1] - runs["walk"][1]).max() < 1e-4, tag


def test_kaist_intrinsics_with_lens_distortion_1280x560(tmp_path, vido):
    """Row A11 inside the facade: the KAIST settings of the reference (src/config/kaist_config.yaml:24-33: 1280 x 560, fx 816.4, k1 = -0.05004 ...), ChooseData: 3, the
    reference's disk layout.  Frame::Frame runs UndistortKeyPoints (Frame.cc:603-633 -> vido_undistort_points, pinned bit-exactly against the oracle in
    tests/test_trackhost_cpu.py) on every frame: mvKeysUn differs from mvKeys by more than a pixel at the image corners, and — as in the reference, whose tracking lists are
    built from mvKeys, not mvKeysUn — the poses still follow the renderer's (pinhole) ground truth."""
    sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd"))
    import build
    driver = build.build_driver()
    n = 8
    K = (816.402, 817.38, 608.2658, 266.688); dist = (-0.05004, 0.120012, -0.0006259, -0.00118, -0.063505)
    scene = vido.synth.Scene3D(n_frames=n, w=1280, h=560, K=K, seed=5, step=0.2, yaw_deg=0.1, objects=((-2.0, 0.4, 9.0, 0.02, 0.0, 0.22),), wall_z=48.0)
    cfg = write_kaist_clip(str(tmp_path), scene, n, dist=dist)
    out = os.path.join(str(tmp_path), "poses.txt")
    r = subprocess.run([driver, cfg, out, os.path.join(str(tmp_path), "res_")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    line = [l for l in r.stdout.splitlines() if l.startswith("undistort keys")][0].split()
    assert int(line[2]) > 1000 and float(line[4]) > 1.0, line                 # the undistortion ran and moved keypoints (corner shift at these coefficients: several px)
    P = np.loadtxt(out)
    assert P.shape == (n, 17)
    for k in range(1, n):
        E = P[k, 1:].reshape(4, 4) @ np.linalg.inv(scene.Tcw(k))
        assert np.linalg.norm(E[:3, 3]) < 0.08, (k, E)
    # the same clip declared distortion-free: identical poses (nothing downstream of the tracking lists reads mvKeysUn — Tracking.cc:283-782), no displacement reported
    cfg0 = write_kaist_clip(str(tmp_path), scene, n)
    out0 = os.path.join(str(tmp_path), "poses0.txt")
    r0 = subprocess.run([driver, cfg0, out0, os.path.join(str(tmp_path), "res0_")], capture_output=True, text=True, timeout=600)
    assert r0.returncode == 0, r0.stderr + r0.stdout
    line0 = [l for l in r0.stdout.splitlines() if l.startswith("undistort keys")][0].split()
    assert float(line0[4]) == 0.0
    assert np.array_equal(np.loadtxt(out0), P)


def test_use_sample_feature_option(tmp_path, vido):
    """UseSampleFeature: 1 (Frame.cc:101-150, Tracking.cc:3013-3018): the static candidates are 3000 random grid samples instead of the ORB keypoints; same filter, same
    downstream pipeline.  (The reference seeds its generator with time(NULL); this build seeds a counter generator with the frame id, so the run is repeatable.)"""
    sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd"))
    import build
    driver = build.build_driver()
    n = 8
    scene = vido.synth.Scene3D(n_frames=n, seed=3, objects=((-2.0, 0.2, 9.0, 0.25, 0.0, 0.05),))
    cfg = write_clip(str(tmp_path), scene, n, sample_features=1)
    outs = []
    for run in range(2):
        out = os.path.join(str(tmp_path), "poses%d.txt" % run)
        r = subprocess.run([driver, cfg, out, os.path.join(str(tmp_path), "res%d_" % run)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        outs.append(np.loadtxt(out))
    assert np.array_equal(outs[0], outs[1])                                       # repeatable
    P = outs[0]
    for k in range(1, n):
        E = P[k, 1:].reshape(4, 4) @ np.linalg.inv(scene.Tcw(k))
        assert np.linalg.norm(E[:3, 3]) < 0.05, (k, E)


def test_window_solve_beside_the_next_frame_gives_the_same_result_files(tmp_path, vido):
    """VIDO_LBA_ASYNC=1 (facade.cpp start_local_ba / finish_local_ba): the local window solve of frame k on a helper thread and a context of its own, joined before the Map
    grows again.  Same clip through the offline driver in both orders: every pose the tracker returns, the refined trajectory and the object-motion files must be IDENTICAL
    (nothing the tracker reads depends on the solve; the solves see the same Map in the same order)."""
    sys.path.insert(0, os.path.join(ROOT, "vido-slam_amd"))
    import build
    driver = build.build_driver()
    n = 30                                                                    # past WINDOW_SIZE: full 20-frame windows and ring evictions are covered
    scene = vido.synth.Scene3D(n_frames=n, seed=5, objects=((-2.0, 0.2, 9.0, 0.25, 0.0, 0.05),))
    cfg = write_clip(str(tmp_path), scene, n)
    outs = {}
    for mode in ("sync", "async"):
        env = {k: v for k, v in os.environ.items() if k not in ("VIDO_LBA_ASYNC", "VIDO_LBA_SYNC")}
        if mode == "async":
            env["VIDO_LBA_ASYNC"] = "1"
        out = os.path.join(str(tmp_path), "poses_%s.txt" % mode); pre = os.path.join(str(tmp_path), "res_%s_" % mode)
        r = subprocess.run([driver, cfg, out, pre], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr + r.stdout
        outs[mode] = (np.loadtxt(out), np.loadtxt(pre + "refined_rgbd_new.txt"), np.loadtxt(pre + "initial_rgbd_new.txt"))
    for a, b in zip(outs["sync"], outs["async"]):
        assert a.shape == b.shape and np.array_equal(a, b)
