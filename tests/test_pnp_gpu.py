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
    Tr, maskr, cntr = oracle.pnp_ransac(s["Xw"], s["uv_cur"], s["K"], seed=seed)
    assert np.abs(T - Tr).max() < 1e-6
    assert abs(cnt - cntr) <= 2 and (mask != maskr).sum() <= 2
    assert np.abs(T - s["T_cur"]).max() < 0.05 and cnt > 0.3 * n * (1 - outl)


def test_pnp_degenerate(vido):
    ctx = vido.Context()
    T, mask, cnt = vido.pnp_ransac(ctx, np.zeros((3, 3)), np.zeros((3, 2)), (500, 500, 320, 240))
    assert cnt == 0 and np.allclose(T, np.eye(4))
