"""GPU parity: seeded P3P-RANSAC (vido_pnp_ransac) vs the CPU oracle (oracle/pnp_oracle.c) — same samples, same
sequential bookkeeping => same winning hypothesis; pose compared at 1e-6 (FP64 transcendental/rounding differences),
inlier masks identical except for points within 1e-6 px^2 of the threshold."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,outl,seed", [(3000, 0.3, 1), (800, 0.1, 2), (60, 0.0, 3), (200, 0.6, 4)])
def test_pnp_ransac_matches_oracle(vido, oracle, n, outl, seed):
    ctx = vido.Context()
    s = vido.problems.synth_pose_scene(n, seed=seed, noise_px=0.05, outlier_frac=outl)
    T, mask, cnt = vido.pnp_ransac(ctx, s["Xw"], s["uv_cur"], s["K"], seed=seed)
    Tr, maskr, cntr, Tm = oracle.pnp_ransac(s["Xw"], s["uv_cur"], s["K"], seed=seed, with_ransac_model=True)
    assert np.abs(T - Tr).max() < 1e-6
    # the returned pose is the all-inlier refit (cv::solvePnPRansac re-estimates on the inliers, Tracking.cc:1967-1970 gets that pose): a least-squares pose over hundreds of
    # points, closer to the truth than the 3-point model it started from, with a smaller reprojection RMS over the inliers
    def rms(Tx):
        Xc = s["Xw"][maskr] @ Tx[:3, :3].T + Tx[:3, 3]; fx, fy, cx, cy = s["K"]
        return np.sqrt(np.mean((fx * Xc[:, 0] / Xc[:, 2] + cx - s["uv_cur"][maskr, 0]) ** 2 + (fy * Xc[:, 1] / Xc[:, 2] + cy - s["uv_cur"][maskr, 1]) ** 2))
    assert rms(T) <= rms(Tm) + 1e-12
    if n >= 200:
        assert np.abs(T - s["T_cur"]).max() <= np.abs(Tm - s["T_cur"]).max()
    Tr = Tm                                                           # the inlier mask belongs to the RANSAC model (OpenCV does not re-evaluate it after the refit)
    # inlier masks: EXACT, except for points whose squared reprojection error under the oracle's pose lies within EPS of the 0.4 px threshold (the two poses
    # differ by FP64 rounding of the quartic's roots, so only such points can fall on different sides)
    EPS = 1e-6
    fx, fy, cx, cy = s["K"]
    Xc = s["Xw"] @ Tr[:3, :3].T + Tr[:3, 3]
    e2 = (fx * Xc[:, 0] / Xc[:, 2] + cx - s["uv_cur"][:, 0]) ** 2 + (fy * Xc[:, 1] / Xc[:, 2] + cy - s["uv_cur"][:, 1]) ** 2
    band = np.abs(e2 - 0.4 ** 2) <= EPS
    assert np.array_equal(mask[~band], maskr[~band])
    assert abs(cnt - cntr) <= int(band.sum())
    assert np.abs(T - s["T_cur"]).max() < 0.05 and cnt > 0.3 * n * (1 - outl)


def test_pnp_batch_equals_single_calls(vido):
    """vido_pnp_ransac_batch (all objects of a frame in one launch pair, one synchronisation) == one vido_pnp_ransac per problem, bit for bit."""
    import ctypes as C
    ctx = vido.Context()
    probs = [vido.problems.synth_pose_scene(n, seed=10 + i, noise_px=0.05, outlier_frac=0.2) for i, n in enumerate((800, 3, 450, 0, 1200))]
    K = probs[0]["K"]
    X = [np.ascontiguousarray(p["Xw"], np.float32) for p in probs]; x = [np.ascontiguousarray(p["uv_cur"], np.float32) for p in probs]
    ns = np.array([len(a) for a in X], np.int32); seeds = np.arange(5, dtype=np.uint64) + 77
    T = np.zeros((5, 16)); masks = [np.zeros(max(n, 1), np.uint8) for n in ns]; cnt = np.zeros(5, np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    p3 = (C.c_void_p * 5)(*[a.ctypes.data if len(a) else None for a in X]); p2 = (C.c_void_p * 5)(*[a.ctypes.data if len(a) else None for a in x]); pm = (C.c_void_p * 5)(*[m.ctypes.data for m in masks])
    ctx._check(ctx.lib.vido_pnp_ransac_batch(ctx.h, 5, p3, p2, vp(ns), C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]), C.c_double(K[3]), 500, C.c_double(0.4), C.c_double(0.98),
                                             vp(seeds), vp(T), pm, vp(cnt)))
    for i, p in enumerate(probs):
        Ti, mi, ci = vido.pnp_ransac(ctx, p["Xw"], p["uv_cur"], K, seed=int(seeds[i]))
        assert np.array_equal(T[i].reshape(4, 4), Ti) and ci == cnt[i] and np.array_equal(masks[i][:ns[i]].astype(bool), mi), i


def test_pnp_degenerate(vido):
    ctx = vido.Context()
    T, mask, cnt = vido.pnp_ransac(ctx, np.zeros((3, 3)), np.zeros((3, 2)), (500, 500, 320, 240))
    assert cnt == 0 and np.allclose(T, np.eye(4))
