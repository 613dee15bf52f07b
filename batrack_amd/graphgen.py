"""Deterministic synthetic factor graphs for the BA hot path (numpy only).

Shapes follow what the reference's caller hands to ``BA_rgbd_droid``
(/root/reference/main/batrack.py:856-875): a pose buffer [N_buf,7]
(tx ty tz qx qy qz qw), a patch buffer [P_tot,3] = (x, y, inverse depth) with
track id ``kk = frame*M + slot`` (batrack.py:403-404), per-frame intrinsics
[N_buf,4], targets stored as (u, v, disp) rows so that the 2-D target is a
stride-3 view (batrack.py:871), visibility weights [E,2] and the
motion-decoupled ``weights_pose`` (batrack.py:789-792).

The recipe is SURVEY.md §8(d): linear camera ramp, uniform tracks, K
observations per track with ``jj = clamp(ii + k - K/2)`` (self-edges and clamped
duplicates are legal), 0.5 px target noise, 30 % of tracks marked dynamic.

Everything is float64 here; callers cast.  This module is input synthesis only:
no part of the BA computation lives here.
"""
from __future__ import annotations

import dataclasses

import numpy as np

SINTEL = dict(wd=1024, ht=436, fx=500.0, fy=500.0, cx=512.0, cy=218.0)
SHIBUYA = dict(wd=640, ht=360, fx=772.548, fy=772.548, cx=320.0, cy=180.0)
SHIBUYA_CROP = dict(SHIBUYA, ht=352)      # as the pipeline sees it: frames cropped to multiples of 16 (stream.py:156-157, 270-271)
DAVIS = dict(wd=848, ht=480, fx=600.0, fy=600.0, cx=424.0, cy=240.0)

# name -> (N, M, K)
CONFIGS = {
    "C1": (8, 32, 8),      # 8 KF / 2,048 edges / 256 tracks
    "C3": (64, 256, 8),    # 64 KF / 131,072 edges / 16,384 tracks
}


@dataclasses.dataclass
class Graph:
    poses: np.ndarray          # [N_buf,7] initial (perturbed) poses
    poses_gt: np.ndarray       # [N_buf,7]
    patches: np.ndarray        # [P_tot,3]  (x, y, disp) initial
    disp_gt: np.ndarray        # [P_tot]
    mono_disp: np.ndarray      # [P_tot]    prior (SURVEY: = GT disparity)
    intrinsics: np.ndarray     # [N_buf,4]
    targets3: np.ndarray       # [E,3]      (u, v, disp) ; 2-D target = [:, :2]
    weights: np.ndarray        # [E,2]
    weights_pose: np.ndarray   # [E,2]
    ii: np.ndarray             # [E] int64 source frame
    jj: np.ndarray             # [E] int64 target frame
    kk: np.ndarray             # [E] int64 track / patch slot
    bounds: tuple              # (0, 0, wd, ht)
    n_frames: int
    M: int

    @property
    def E(self):
        return int(self.ii.shape[0])


# ---------------------------------------------------------------- SE3 helpers
def _quat_mul(a, b):
    ax, ay, az, aw = np.moveaxis(a, -1, 0)
    bx, by, bz, bw = np.moveaxis(b, -1, 0)
    return np.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def _quat_rot(q, p):
    qv = q[..., :3]
    uv = 2.0 * np.cross(qv, p)
    return p + q[..., 3:4] * uv + np.cross(qv, uv)


def se3_exp(xi):
    """Exp of (tau, phi) rows -> (t, q) rows; closed form, float64."""
    xi = np.asarray(xi, dtype=np.float64)
    tau, phi = xi[..., :3], xi[..., 3:]
    th2 = (phi * phi).sum(-1, keepdims=True)
    th = np.sqrt(th2)
    small = th < 1e-6
    ths = np.where(small, 1.0, th)
    imag = np.where(small, 0.5 - th2 / 48.0, np.sin(0.5 * ths) / ths)
    real = np.where(small, 1.0 - th2 / 8.0, np.cos(0.5 * ths))
    q = np.concatenate([imag * phi, real], -1)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    c1 = np.where(small, 0.5 - th2 / 24.0, (1.0 - np.cos(ths)) / (ths * ths))
    c2 = np.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - np.sin(ths)) / (ths ** 3))
    pxt = np.cross(phi, tau)
    t = tau + c1 * pxt + c2 * np.cross(phi, pxt)
    return np.concatenate([t, q], -1)


def se3_mul(a, b):
    q = _quat_mul(a[..., 3:], b[..., 3:])
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    t = a[..., :3] + _quat_rot(a[..., 3:], b[..., :3])
    return np.concatenate([t, q], -1)


def se3_inv(a):
    qi = a[..., 3:] * np.array([-1.0, -1.0, -1.0, 1.0])
    return np.concatenate([-_quat_rot(qi, a[..., :3]), qi], -1)


def reproject(poses, patches, intr, ii, jj, kk):
    """Pinhole reprojection of track kk from frame ii into frame jj -> (u, v, Z)."""
    Gij = se3_mul(poses[jj], se3_inv(poses[ii]))
    x, y, d = patches[kk, 0], patches[kk, 1], patches[kk, 2]
    fxi, fyi, cxi, cyi = intr[ii].T
    fxj, fyj, cxj, cyj = intr[jj].T
    X0 = np.stack([(x - cxi) / fxi, (y - cyi) / fyi, np.ones_like(x)], -1)
    P = _quat_rot(Gij[:, 3:], X0) + Gij[:, :3] * d[:, None]
    Z = np.maximum(P[:, 2], 1e-2)
    return fxj * P[:, 0] / Z + cxj, fyj * P[:, 1] / Z + cyj, P[:, 2]


# ------------------------------------------------------------------ generator
def make_graph(N, M, K, seed=0, cam=SINTEL, n_buf=None, shuffle=False,
               dyn_frac=0.3, px_noise=0.5, pose_noise=0.01, disp_noise=0.1):
    """SURVEY.md §8(d) ``make_graph(N, M, K, seed)``.

    n_buf   : pose-buffer length (>= N); extra slots hold the identity pose and
              zero-disparity patches, like the reference's BUFFER_SIZE slots.
    shuffle : randomly permute the edge list (the API accepts any order).
    """
    rng = np.random.default_rng(seed)
    n_buf = N if n_buf is None else int(n_buf)
    assert n_buf >= N
    wd, ht = cam["wd"], cam["ht"]

    ramp = np.linspace(0.0, 1.0, N)[:, None]
    xi_gt = ramp * np.array([0.5, 0.0, 1.0, 0.0, 0.1, 0.0])
    poses_gt = np.tile(np.array([0, 0, 0, 0, 0, 0, 1.0]), (n_buf, 1))
    poses_gt[:N] = se3_exp(xi_gt)
    pert = rng.normal(0.0, pose_noise, size=(N, 6))
    pert[0] = 0.0
    poses = poses_gt.copy()
    poses[:N] = se3_mul(se3_exp(pert), poses_gt[:N])

    P_tot = n_buf * M
    patches = np.zeros((P_tot, 3))
    disp_gt = np.zeros(P_tot)
    na = N * M
    patches[:na, 0] = rng.uniform(20.0, wd - 20.0, na)
    patches[:na, 1] = rng.uniform(20.0, ht - 20.0, na)
    disp_gt[:na] = rng.uniform(0.2, 1.0, na)
    patches[:na, 2] = disp_gt[:na] * (1.0 + rng.normal(0.0, disp_noise, na))
    mono = disp_gt.copy()

    intr = np.tile(np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"]]), (n_buf, 1))

    trk = np.arange(na, dtype=np.int64)
    kk = np.repeat(trk, K)
    ii = kk // M
    off = np.tile(np.arange(K, dtype=np.int64) - K // 2, na)
    jj = np.clip(ii + off, 0, N - 1)

    gt_patches = patches.copy()
    gt_patches[:, 2] = disp_gt
    u, v, _ = reproject(poses_gt, gt_patches, intr, ii, jj, kk)
    E = kk.shape[0]
    targets3 = np.zeros((E, 3))
    targets3[:, 0] = u + rng.normal(0.0, px_noise, E)
    targets3[:, 1] = v + rng.normal(0.0, px_noise, E)
    targets3[:, 2] = disp_gt[kk]

    weights = np.ones((E, 2))
    dynamic = rng.random(na) < dyn_frac
    weights_pose = weights * (~dynamic[kk])[:, None]

    if shuffle:
        p = rng.permutation(E)
        ii, jj, kk = ii[p], jj[p], kk[p]
        targets3, weights, weights_pose = targets3[p], weights[p], weights_pose[p]

    return Graph(poses=poses, poses_gt=poses_gt, patches=patches, disp_gt=disp_gt,
                 mono_disp=mono, intrinsics=intr, targets3=targets3,
                 weights=weights, weights_pose=np.ascontiguousarray(weights_pose),
                 ii=ii, jj=jj, kk=kk, bounds=(0.0, 0.0, float(wd), float(ht)),
                 n_frames=N, M=M)


def make_random_graph(N, M, seed=0, deg=(2, 6), far_frac=0.3, groups=1, cam=SINTEL, n_buf=None, shuffle=True):
    """Irregular co-visibility: every track of frame i is observed from a random number of frames, most of
    them near i, a fraction anywhere in the trajectory (loop-closure-like), self edges and repeated
    observations included.  Exercises elimination orders, fill and level schedules that the banded
    ``make_graph`` never produces.  ``groups`` > 1 splits the frames into that many sub-sequences with no
    edges between them (a forest of elimination trees).  Geometry, noise and weights as in ``make_graph``."""
    g = make_graph(N, M, 1, seed=seed, cam=cam, n_buf=n_buf)
    rng = np.random.default_rng(seed + 1000)
    na = N * M
    ii_l, jj_l, kk_l = [], [], []
    for k in range(na):
        i = k // M
        gsz = -(-N // groups)
        lo, hi = (i // gsz) * gsz, min(N, (i // gsz + 1) * gsz)
        nd = int(rng.integers(deg[0], deg[1] + 1))
        for _ in range(nd):
            if rng.random() < far_frac:
                j = int(rng.integers(lo, hi))
            else:
                j = int(np.clip(i + rng.integers(-3, 4), lo, hi - 1))
            ii_l.append(i); jj_l.append(j); kk_l.append(k)
    ii, jj, kk = (np.asarray(a, np.int64) for a in (ii_l, jj_l, kk_l))
    gt_patches = g.patches.copy()
    gt_patches[:, 2] = g.disp_gt
    u, v, _ = reproject(g.poses_gt, gt_patches, g.intrinsics, ii, jj, kk)
    E = kk.shape[0]
    targets3 = np.zeros((E, 3))
    targets3[:, 0] = u + rng.normal(0.0, 0.5, E)
    targets3[:, 1] = v + rng.normal(0.0, 0.5, E)
    targets3[:, 2] = g.disp_gt[kk]
    weights = rng.uniform(0.2, 1.0, (E, 2))
    dynamic = rng.random(na) < 0.3
    weights_pose = weights * (~dynamic[kk])[:, None]
    if shuffle:
        p = rng.permutation(E)
        ii, jj, kk, targets3, weights, weights_pose = ii[p], jj[p], kk[p], targets3[p], weights[p], weights_pose[p]
    return Graph(poses=g.poses, poses_gt=g.poses_gt, patches=g.patches, disp_gt=g.disp_gt, mono_disp=g.mono_disp,
                 intrinsics=g.intrinsics, targets3=targets3, weights=weights,
                 weights_pose=np.ascontiguousarray(weights_pose), ii=ii, jj=jj, kk=kk, bounds=g.bounds,
                 n_frames=N, M=M)


def make_config(name, seed=0, **kw):
    N, M, K = CONFIGS[name]
    return make_graph(N, M, K, seed=seed, **kw)


def make_window_graph(n_frames=50, M=256, window=12, kf_stride=2, opt_window=15,
                      removal=20, seed=0, cam=SINTEL, n_buf=None):
    """Real-shape sliding-window graph (SURVEY.md §8 'real Sintel steady state').

    Replays the reference's edge bookkeeping: every ``kf_stride`` frames all
    patches of the keyframes inside the last ``window`` frames are connected to
    every frame of that window (batrack.py:399-410), duplicates included, and
    edges whose source frame is older than ``removal`` frames are dropped
    (batrack.py:1020-1024).  Returns (graph, fixedp) at the final frame, with
    ``fixedp = n - opt_window`` (batrack.py:858-859).
    """
    n = n_frames
    g = make_graph(n, M, 1, seed=seed, cam=cam, n_buf=n_buf)
    rng = np.random.default_rng(seed + 1)
    kk_l, jj_l = [], []
    for cur in range(2, n + 1, kf_stride):
        lo = max(cur - window, 0)
        kfs = np.arange(lo, cur, kf_stride)
        pat = (kfs[:, None] * M + np.arange(M)[None, :]).reshape(-1)
        frames = np.arange(lo, cur)
        kk_l.append(np.repeat(pat, frames.size))
        jj_l.append(np.tile(frames, pat.size))
    kk = np.concatenate(kk_l).astype(np.int64)
    jj = np.concatenate(jj_l).astype(np.int64)
    ii = kk // M
    keep = ii >= n - removal
    ii, jj, kk = ii[keep], jj[keep], kk[keep]
    gt_patches = g.patches.copy()
    gt_patches[:, 2] = g.disp_gt
    u, v, _ = reproject(g.poses_gt, gt_patches, g.intrinsics, ii, jj, kk)
    E = kk.shape[0]
    t3 = np.zeros((E, 3))
    t3[:, 0] = u + rng.normal(0.0, 0.5, E)
    t3[:, 1] = v + rng.normal(0.0, 0.5, E)
    t3[:, 2] = g.disp_gt[kk]
    weights = np.ones((E, 2))
    dynamic = rng.random(n * M) < 0.3
    wp = np.ascontiguousarray(weights * (~dynamic[kk])[:, None])
    g2 = dataclasses.replace(g, targets3=t3, weights=weights, weights_pose=wp,
                             ii=ii, jj=jj, kk=kk)
    return g2, max(n - opt_window, 1)


def roughen(g: Graph, seed=1):
    """Return a copy of ``g`` that exercises the discontinuities the reference
    keeps (SURVEY.md §7 'Discontinuities are part of the contract'): zeroed
    visibility, far-off targets (>250 px), out-of-bounds and behind-camera
    points, prior-less tracks, per-frame intrinsics, shuffled edge order."""
    rng = np.random.default_rng(seed)
    E = g.E
    t3 = g.targets3.copy()
    w = g.weights.copy()
    wp = g.weights_pose.copy()
    far = rng.random(E) < 0.03
    t3[far, 0] += 400.0
    w[rng.random(E) < 0.05] = 0.0
    wp = wp * (w > 0)
    patches = g.patches.copy()
    na = g.n_frames * g.M
    edge = rng.random(na) < 0.04
    patches[:na, 0] = np.where(edge, rng.uniform(-30.0, 5.0, na), patches[:na, 0])
    tiny = rng.random(na) < 0.03
    patches[:na, 2] = np.where(tiny, 8.0, patches[:na, 2])   # very close points
    mono = g.mono_disp.copy()
    mono[:na] = np.where(rng.random(na) < 0.2, 0.0, mono[:na])
    intr = g.intrinsics.copy()
    intr[:, 0] *= 1.0 + 0.02 * rng.standard_normal(intr.shape[0])
    intr[:, 1] *= 1.0 + 0.02 * rng.standard_normal(intr.shape[0])
    intr[:, 2] += rng.normal(0, 3.0, intr.shape[0])
    p = rng.permutation(E)
    return dataclasses.replace(
        g, patches=patches, mono_disp=mono, intrinsics=intr,
        targets3=t3[p], weights=w[p], weights_pose=np.ascontiguousarray(wp[p]),
        ii=g.ii[p], jj=g.jj[p], kk=g.kk[p])
