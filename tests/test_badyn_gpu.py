"""Row A25 (object part of FullBatchOptimization) on the MI355X: the HIP path (static points by 3x3 Schur, dynamic
tracklets by block-tridiagonal chain elimination, reduced pose system over cameras + object motions) against the oracle,
which solves the full un-eliminated system densely like g2o does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4        # SE(3) / point parity, relative (north_star); observed ~1e-9


@pytest.fixture(scope="module")
def ctx(vido):
    c = vido.Context()
    yield c
    c.close()


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max()))


@pytest.mark.parametrize("n_cam,n_pt,n_obj,ppo,seed", [(6, 40, 2, 6, 4), (8, 60, 2, 8, 9), (14, 80, 3, 10, 5)])
def test_dynamic_ba_matches_oracle(vido, oracle, ctx, n_cam, n_pt, n_obj, ppo, seed):
    P = vido.problems
    base = P.synth_ba_problem(n_cam=n_cam, n_pt=n_pt, kind="global", track_len=5, seed=seed)
    dyn = P.synth_ba_dynamic(base, n_obj=n_obj, pts_per_obj=ppo, seed=seed + 1)
    base["max_iters"] = 25
    ref = oracle.badyn_optimize(base, dyn)
    got = vido.ba_optimize(ctx, base, dynamic=dyn)
    assert abs(got["chi2_initial"] - ref["chi2_initial"]) <= 1e-9 * ref["chi2_initial"]
    assert (got["iterations"], got["lm_trials"]) == (ref["iterations"], ref["lm_trials"])
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"]
    for key in ("cam_T", "pt_xyz", "H_T", "dyn_xyz"):
        assert rel(got[key], ref[key]) < TOL, key


@pytest.mark.parametrize("n_cam,max_len,seed", [(40, 5, 3), (48, 12, 6)])
def test_dynamic_ba_band_layout_matches_oracle(vido, oracle, ctx, n_cam, max_len, seed):
    """longer sequences with bounded tracklets: poses are interleaved by frame and the reduced system is banded (pose-block LDS-window
    Cholesky for the short band, scalar banded Cholesky in HBM for the wide one); the oracle still solves the dense full system"""
    P = vido.problems
    base = P.synth_ba_problem(n_cam=n_cam, n_pt=60, kind="global", track_len=4, seed=seed)
    dyn = P.synth_ba_dynamic(base, n_obj=2, pts_per_obj=5, seed=seed + 1, max_len=max_len)
    base["max_iters"] = 6
    ref = oracle.badyn_optimize(base, dyn)
    got = vido.ba_optimize(ctx, base, dynamic=dyn)
    assert (got["iterations"], got["lm_trials"]) == (ref["iterations"], ref["lm_trials"])
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"]
    for key in ("cam_T", "pt_xyz", "H_T", "dyn_xyz"):
        assert rel(got[key], ref[key]) < TOL, key


def test_configs3b_20kf_2k_landmarks_5_objects_x_100_points_matches_sparse_oracle(vido, oracle, ctx):
    """BASELINE configs[3] with its dynamic part at the defining size: 20 keyframes x 2 000 static landmarks (35.6 k observations) + 5 objects x 100
    points tracked through all 20 frames = 10 000 dynamic point vertices, 9 500 ternary edges, 95 object-motion vertices (36 690 unknowns).  The oracle solves the sparse
    un-eliminated system with SuperLU (vo_badyn_optimize_sparse, pinned to the dense form in tests/test_oracle_cpu.py); the HIP path eliminates static points (3x3 Schur)
    and dynamic chains (block tridiagonal) and solves the 690-unknown pose system."""
    import copy
    P = vido.problems
    base = P.synth_ba_problem(n_cam=20, n_pt=2000, kind="local", seed=7)
    dyn = P.synth_ba_dynamic(base, n_obj=5, pts_per_obj=100, seed=8, full_tracks=True)
    assert (dyn["n_H"], dyn["n_dyn"], dyn["n_tern"]) == (95, 10000, 9500) and len(base["obs_cam"]) > 30000
    base["max_iters"] = 20
    ref = oracle.badyn_optimize_sparse(copy.deepcopy(base), copy.deepcopy(dyn))
    got = vido.ba_optimize(ctx, base, dynamic=dyn)
    assert abs(got["chi2_initial"] - ref["chi2_initial"]) <= 1e-9 * ref["chi2_initial"]
    assert (got["iterations"], got["lm_trials"]) == (ref["iterations"], ref["lm_trials"])
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"]
    for key in ("cam_T", "pt_xyz", "H_T", "dyn_xyz"):
        assert rel(got[key], ref[key]) < TOL, key


def test_dynamic_ba_first_step_exact(vido, oracle, ctx):
    """one LM iteration: identical step => the elimination scheme is the same linear solve as the dense oracle"""
    P = vido.problems
    base = P.synth_ba_problem(n_cam=7, n_pt=50, kind="global", track_len=4, seed=12)
    dyn = P.synth_ba_dynamic(base, n_obj=2, pts_per_obj=7, seed=13)
    base["max_iters"] = 1
    ref = oracle.badyn_optimize(base, dyn); got = vido.ba_optimize(ctx, base, dynamic=dyn)
    assert got["lm_trials"] == ref["lm_trials"]
    for key in ("cam_T", "pt_xyz", "H_T", "dyn_xyz"):
        assert rel(got[key], ref[key]) < 1e-8, key


def test_dynamic_ba_rejects_non_chain_graphs(vido, ctx):
    P = vido.problems
    base = P.synth_ba_problem(n_cam=6, n_pt=40, kind="global", track_len=4, seed=3)
    dyn = P.synth_ba_dynamic(base, n_obj=1, pts_per_obj=4, seed=4)
    dyn["tern_cur"] = dyn["tern_cur"].copy(); dyn["tern_cur"][1] = dyn["tern_cur"][0]      # two predecessors
    with pytest.raises(vido.VidoError):
        vido.ba_optimize(ctx, base, dynamic=dyn)


def test_static_problem_unchanged_by_the_dynamic_entry(vido, oracle, ctx):
    P = vido.problems
    base = P.synth_ba_problem(n_cam=8, n_pt=100, kind="local", seed=2)
    a = vido.ba_optimize(ctx, base); b = oracle.ba_optimize(base)
    assert (a["iterations"], a["lm_trials"]) == (b["iterations"], b["lm_trials"]) and rel(a["cam_T"], b["cam_T"]) < 1e-8
