"""Trajectory error as the reference reports it: absolute trajectory error (ATE) = RMSE of the
translation part after a Sim(3) (Umeyama) alignment of the estimate onto the reference
trajectory (/root/reference/main/utils.py:337-340: evo `main_ape.ape(..., pose_relation=
translation_part, align=True, correct_scale=True)`, statistic `rmse`).  `evo` is not available
offline; this restates the published closed form (Umeyama 1991) in float64 numpy.

Host-side measurement code (SURVEY.md §8d "ATE"): it consumes trajectories, it is not on the
BA hot path.
"""
import numpy as np


def umeyama(est, ref, with_scale=True):
    """Least-squares similarity (s, R, t) with ref ~ s * R @ est + t; est, ref: [n, 3]."""
    est = np.asarray(est, np.float64)
    ref = np.asarray(ref, np.float64)
    if est.shape != ref.shape or est.ndim != 2 or est.shape[1] != 3 or est.shape[0] < 3:
        raise ValueError("umeyama: two [n>=3, 3] point sets expected")
    mu_e, mu_r = est.mean(0), ref.mean(0)
    xe, xr = est - mu_e, ref - mu_r
    cov = xr.T @ xe / est.shape[0]
    U, d, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0.0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    var_e = (xe * xe).sum() / est.shape[0]
    s = float(np.trace(np.diag(d) @ S) / var_e) if with_scale else 1.0
    t = mu_r - s * R @ mu_e
    return s, R, t


def camera_centres(poses):
    """World-to-camera poses [n, 7] = (t, q xyzw) -> camera centres in the world, -R^T t
    (the reference saves `poses.inv()`, batrack.py:1086-1087)."""
    p = np.asarray(poses, np.float64)
    t, q = p[:, :3], p[:, 3:] / np.linalg.norm(p[:, 3:], axis=1, keepdims=True)
    qv, w = -q[:, :3], q[:, 3:4]                      # conjugate: rotate by R^T
    uv = 2.0 * np.cross(qv, t)
    return -(t + w * uv + np.cross(qv, uv))


def ate_rmse(est_xyz, ref_xyz, align=True, correct_scale=True):
    """APE-RMSE of the translation part (utils.py:337-340)."""
    est = np.asarray(est_xyz, np.float64)
    ref = np.asarray(ref_xyz, np.float64)
    if align:
        s, R, t = umeyama(est, ref, with_scale=correct_scale)
        est = s * est @ R.T + t
    return float(np.sqrt(((est - ref) ** 2).sum(1).mean()))
