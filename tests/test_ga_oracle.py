"""oracle/ga_losses.py (dense global-alignment losses) against the vectors the reference's unmodified
refine_net.py produced (tests/golden/ga_small.npz, tests/golden/make_golden_ga.py)."""
import os

import numpy as np

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
