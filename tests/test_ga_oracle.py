"""oracle/ga_losses.py (dense global-alignment losses) against the vectors the reference's unmodified
refine_net.py produced (tests/golden/ga_small.npz, tests/golden/make_golden_ga.py)."""
import os

import numpy as np
import torch

from oracle import ga_losses as ga

D = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ga_small.npz")))
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))


def test_float64_matches_reference():
    assert rel(ga.trajs_scales(D["trajs_scales"], float(D["pw_break"])), D["f64.trajs_scales_exp"]) < 1e-14
    ms = ga.frame_scaled_depth(D)
    assert rel(ms, D["f64.mono_scaled"]) < 1e-13
    assert abs(ga.spatial_loss(D) / D["f64.loss_spatial"] - 1) < 1e-12
    assert abs(ga.inter_frame_loss(D) / D["f64.loss_rigid"] - 1) < 1e-12
    assert abs(ga.pts_3d_loss(D) / D["f64.loss_pts3d"] - 1) < 1e-6          # the reference casts the points to float32 (refine_net.py:312,326)
    assert abs(ga.forward(D, 0.5) / D["f64.total_alpha05"] - 1) < 1e-12


def test_float32_within_rounding():
    d32 = {k: (v.astype(np.float32) if v.dtype == np.float64 and v.ndim > 0 else v) for k, v in D.items()}
    assert rel(ga.frame_scaled_depth(d32), D["f64.mono_scaled"]) < 1e-6
    assert abs(ga.inter_frame_loss(d32) / D["f64.loss_rigid"] - 1) < 1e-5
    assert abs(ga.spatial_loss(d32) / D["f64.loss_spatial"] - 1) < 1e-5


def test_torch_statement_and_its_gradients_match_the_reference_autograd():
    """oracle/ga_torch.py: forward equal to the numpy oracle and to the reference, gradients of forward() equal to the ones
    the reference's own autograd produced (fixture keys *.grad_*)."""
    from oracle import ga_torch
    for alpha, key in ((0.0, "a00"), (0.5, "a05")):
        tot, sp, rg, g_ts, g_fs = ga_torch.total_and_grads(D, alpha)
        assert abs(sp / float(D["f64.loss_spatial"]) - 1) < 1e-12
        if alpha > 0:
            assert abs(rg / float(D["f64.loss_rigid"]) - 1) < 1e-12 and abs(tot / float(D["f64.total_alpha05"]) - 1) < 1e-12
        for got, name in ((g_ts, "grad_trajs_scales"), (g_fs, "grad_frame_scales")):
            ref = D[f"f64.{name}_{key}"]
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-30) + 1e-18, (name, key, np.abs(got - ref).max())


def test_full_total_and_every_gradient_match_the_reference_autograd():
    """oracle/ga_torch.py:full_total_and_grads against tests/golden/ga_total.npz — RefineNet.forward in both of its
    branches (the weights of run_global_refine.py:61-67 with refined intrinsics; loss_weight_dict=None at the constructor's
    defaults) and the gradient of the total w.r.t. trajs_scales, frame_scales_, pose and K by the reference's own autograd.
    The oracle derives the pose gradient from a first-order left perturbation, not from the stand-in's formulas.  1e-8: the
    reference casts the 3-D points to float32 inside pts_3d_loss (refine_net.py:322,337) even in its float64 run."""
    from oracle import ga_torch
    G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ga_total.npz")))
    d = {k: G[k] for k in G if "." not in k}
    for name, w, rk in (("A", list(G["weights"]), True), ("B", [1.0, 0.5, 0.0, 0.0, 0.1], False)):
        r = ga_torch.full_total_and_grads(d, w, "l1", refine_intrinsics=rk)
        assert abs(r["total"] / float(G[f"f64.{name}.total"]) - 1) < 1e-8
        assert abs(r["spatial"] / float(G[f"f64.{name}.spatial"]) - 1) < 1e-12
        if w[1]:
            assert abs(r["rigid"] / float(G[f"f64.{name}.rigid"]) - 1) < 1e-12
        if w[2]:
            assert abs(r["pts3d"] / float(G[f"f64.{name}.pts3d"]) - 1) < 1e-7
        if w[3]:
            assert abs(r["cam_smooth"] / float(G[f"f64.{name}.cam_smooth"]) - 1) < 1e-12
        assert abs(r["scale_smooth"] / float(G[f"f64.{name}.scale_smooth_l1"]) - 1) < 1e-12
        for k in ("grad_trajs_scales", "grad_frame_scales", "grad_pose", "grad_K"):
            ref = G[f"f64.{name}.{k}"]
            assert r[k].shape == ref.shape
            assert np.abs(r[k] - ref).max() <= 1e-7 * max(np.abs(ref).max(), 1e-30) + 1e-18, (name, k, np.abs(r[k] - ref).max())
        if not w[2]:
            assert np.all(G[f"f64.{name}.grad_pose"] == 0) and np.all(G[f"f64.{name}.grad_K"] == 0)
    for mode in ("l1", "l2", "huber"):
        got = float(ga_torch.scale_grid_smoothness_loss(torch.as_tensor(d["frame_scales_"], dtype=torch.float64), mode))
        assert abs(got / float(G[f"f64.A.scale_smooth_{mode}"]) - 1) < 1e-12
