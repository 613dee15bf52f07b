"""The BA's caller, replayed on a synthetic sequence: frame bookkeeping, edge list growth and
pruning, the 2*ITER call pattern and the write-back of `BATRACK.__call__` / `update()`
(/root/reference/main/batrack.py:937-1008, 856-895), with the tracker network replaced by a
synthetic observation model (ground-truth reprojection + noise).  SURVEY.md §8(a) rows 0 and 12,
§8(d) "ATE": the same loop is driven once with the HIP `BA_rgbd_droid` and once with the CPU
oracle (tests / tools only), and the two trajectories are compared by their ATE.

`WindowedBA` takes ANY function with the reference's `BA_rgbd_droid` signature (ba.py:217), so it
doubles as the check that batrack_amd.backend.ba is a drop-in at the call site.  State tensors
live on `device`; nothing here is specific to the GPU except the `ba` passed in.

Out of scope (not restated): image preprocessing, patch selection, the tracker network, the
motion-magnitude keyframe removal (`keyframe`, batrack.py:1026-1071 — `keyframe_simple` is used,
as in the reference when `use_keyframe` is off), point-cloud export, visualisers.
"""
import dataclasses
import time

import numpy as np
import torch

from . import graphgen
from .backend import projective_ops as pops
from .backend.lietorch import SE3


@dataclasses.dataclass
class SlamConfig:
    """The `slam:` block of the reference's configs/sintel.yaml (names kept)."""
    MOTION_DAMPING: float = 0.5
    REMOVAL_WINDOW: int = 20
    OPTIMIZATION_WINDOW: int = 15
    PATCHES_PER_FRAME: int = 256
    BUFFER_SIZE: int = 1024
    ITER: int = 4
    LOSS: str = "huber"
    USE_MAP_FILTERING: bool = True
    MAP_FILTERING_TH: float = 5.0
    MIN_TRACK_LEN: int = 3
    S_slam: int = 12
    kf_stride: int = 2
    num_init: int = 12
    init_updates: int = 12            # batrack.py:990-991


class SyntheticObservations:
    """Stand-in for everything upstream of the BA: a camera moving through a cloud of tracked
    points, `M` new tracks per frame.  Static tracks reproject exactly (plus pixel noise);
    a fraction `dyn_frac` drifts in the image (moving objects) and is labelled non-static, which
    is what the reference's motion-decoupling mask removes from `weights_pose`
    (batrack.py:789-792)."""

    def __init__(self, n_frames=50, M=256, seed=0, cam=graphgen.SINTEL, dyn_frac=0.3,
                 px_noise=0.5, disp_noise=0.1, prior_noise=0.05, drop_frac=0.03, dyn_speed=2.0):
        rng = np.random.default_rng(seed)
        self.n_frames, self.M, self.cam = int(n_frames), int(M), cam
        self.wd, self.ht = float(cam["wd"]), float(cam["ht"])
        s = np.arange(n_frames)[:, None] / 64.0                       # per-frame motion of workload C3
        xi = s * np.array([0.5, 0.0, 1.0, 0.0, 0.1, 0.0]) + \
            np.sin(2.0 * np.pi * s * 64.0 / 40.0) * np.array([0.0, 0.04, 0.0, 0.01, 0.0, 0.0])
        self.poses_gt = graphgen.se3_exp(xi)                           # world -> camera, frame 0 = identity
        na = n_frames * M
        self.xy = np.stack([rng.uniform(20.0, self.wd - 20.0, na), rng.uniform(20.0, self.ht - 20.0, na)], 1)
        self.disp_gt = rng.uniform(0.2, 1.0, na)
        self.disp_init = self.disp_gt * (1.0 + rng.normal(0.0, disp_noise, na))      # init_depth(mode='dmap')
        self.disp_prior = self.disp_gt * (1.0 + rng.normal(0.0, prior_noise, na))    # patches_monodisp
        self.dynamic = rng.random(na) < dyn_frac
        self.velocity = rng.normal(0.0, dyn_speed, (na, 2)) * self.dynamic[:, None]
        self.intrinsics = np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"]], np.float64)
        self.px_noise, self.drop_frac = px_noise, drop_frac
        self._rng = np.random.default_rng(seed + 12345)

    def frame_patches(self, f):
        sl = slice(f * self.M, (f + 1) * self.M)
        return np.concatenate([self.xy[sl], self.disp_init[sl, None]], 1), self.disp_prior[sl]

    def predict(self, kk, jj):
        """Targets and labels for edges (track kk -> frame jj): what `get_window_trajs` +
        `predict_target` (batrack.py:760-795) hand to the BA.  Returns targets3 [E,3],
        visible [E], static [E]."""
        ii = kk // self.M
        gt = np.concatenate([self.xy, self.disp_gt[:, None]], 1)
        intr = np.tile(self.intrinsics, (self.n_frames, 1))
        u, v, Z = graphgen.reproject(self.poses_gt, gt, intr, ii, jj, kk)
        E = kk.shape[0]
        dt = (jj - ii).astype(np.float64)
        t3 = np.stack([u + self.velocity[kk, 0] * dt + self._rng.normal(0.0, self.px_noise, E),
                       v + self.velocity[kk, 1] * dt + self._rng.normal(0.0, self.px_noise, E),
                       self.disp_prior[kk] / np.maximum(Z, 1e-2)], 1)          # (the tracker's depth: in the track's own frame, Z = 1, its prior)
        vis = (Z > 0.2) & (self._rng.random(E) >= self.drop_frac)
        return t3, vis, ~self.dynamic[kk]

    def depth_map(self, f):
        """Stand-in for the mono-depth network's map of frame f (`depth` of BATRACK.__call__, batrack.py:937; what
        `get_results(dmaps=...)` stores): a smooth positive field [ht, wd, 1], float32.  It is not the surface the tracks lie
        on — the hand-off to the global-alignment stage only samples it."""
        ht, wd = int(self.ht), int(self.wd)
        y, x = np.mgrid[0:ht, 0:wd].astype(np.float64)
        ph = 0.37 * f
        d = 2.5 + 1.2 * np.sin(2.0 * np.pi * x / wd + ph) * np.cos(2.0 * np.pi * y / ht - 0.5 * ph) + 0.4 * np.sin(6.0 * np.pi * (x + y) / (wd + ht) + ph)
        return d.astype(np.float32)[..., None]

    def centres_gt(self):
        from .evaluation import camera_centres
        return camera_centres(self.poses_gt)


class WindowedBA:
    """Sliding-window sparse SLAM back end around `ba` (same attribute names as the reference's
    BATRACK object where they exist: n, m, M, N, poses_, patches_, ii, jj, kk, targets_3d,
    weights, weights_pose)."""

    def __init__(self, obs, ba, cfg=None, device="cpu", sync=None, prefetch=None, se3=SE3):
        """se3: the pose class (default: the HIP-backed batrack_amd.backend.lietorch.SE3, GPU tensors only; the CPU-side
        tests of this loop pass the test oracle's torch formulas (SE3Ref) together with the oracle `ba`).
        prefetch: optional `batrack_amd.backend.ba.prefetch_plan`; it is called where the reference knows the
        edge list of the coming update() — after `append_factors` (before the tracker pass that `predict_target`
        stands for) and after `keyframe_simple` when the next frame appends nothing."""
        self.obs, self.ba, self.device = obs, ba, torch.device(device)
        self.prefetch = prefetch
        self.SE3 = se3
        self.cfg = cfg or SlamConfig(PATCHES_PER_FRAME=obs.M, BUFFER_SIZE=obs.n_frames + 1)
        c = self.cfg
        if c.PATCHES_PER_FRAME != obs.M or c.BUFFER_SIZE < obs.n_frames + 1:
            raise ValueError("config does not fit the sequence (patches per frame / buffer size)")
        self.N, self.M = c.BUFFER_SIZE, c.PATCHES_PER_FRAME
        self.wd, self.ht = obs.wd, obs.ht
        f32 = dict(dtype=torch.float32, device=self.device)
        i64 = dict(dtype=torch.int64, device=self.device)
        self.poses_ = torch.zeros(self.N, 7, **f32)
        self.poses_[:, 6] = 1.0                                                    # batrack.py:84
        self.patches_ = torch.zeros(self.N, self.M, 3, 1, 1, **f32)
        self.intrinsics_ = torch.as_tensor(obs.intrinsics, **f32).repeat(self.N, 1)
        # the per-track window buffers the tracker side keeps and results.pkl is cut from (batrack.py:66,78-93): a track's
        # targets / visibility / static label / weight in the S_local = 2 S_slam - 1 frames around its own frame
        self.S_local = 2 * c.S_slam - 1
        self.patches_local_ = torch.zeros(self.N, self.M, self.S_local, 3, **f32)
        self.patches_local_vis_ = torch.zeros(self.N, self.M, self.S_local, 1, **f32)
        self.patches_local_static_ = torch.ones(self.N, self.M, self.S_local, 1, **f32)
        self.patches_local_weights_ = torch.zeros(self.N, self.M, self.S_local, 1, **f32)
        self.patches_valid_ = torch.zeros(self.N, self.M, **f32)
        self.tstamps_ = torch.zeros(self.N, **i64)
        self.tlist, self.counter = [], 0
        self.ii, self.jj, self.kk = (torch.zeros(0, **i64) for _ in range(3))
        self.targets_3d = torch.zeros(1, 0, 3, **f32)
        self.weights = torch.zeros(1, 0, 2, **f32)
        self.weights_pose = torch.zeros(1, 0, 2, **f32)
        self.n = self.m = 0
        self.is_initialized = False
        self._sync = sync or (torch.cuda.synchronize if self.device.type == "cuda" else (lambda: None))
        self.stats = dict(updates=0, ba_calls=0, ba_seconds=0.0, edges_max=0, filtered=0)

    # views the reference exposes as properties (batrack.py:140-170)
    @property
    def poses(self):
        return self.poses_.view(1, self.N, 7)

    @property
    def patches(self):
        return self.patches_.view(1, self.N * self.M, 3, 1, 1)

    @property
    def intrinsics(self):
        return self.intrinsics_.view(1, self.N, 4)

    # ---- batrack.py:176-187
    def init_motion(self):
        if self.n > 1:
            SE3 = self.SE3
            P1 = SE3(self.poses_[self.n - 1][None])
            P2 = SE3(self.poses_[self.n - 2][None])
            xi = self.cfg.MOTION_DAMPING * (P1 * P2.inv()).log()
            self.poses_[self.n] = (SE3.exp(xi) * P1).data[0]
        elif self.n == 1:
            self.poses_[self.n] = self.poses_[self.n - 1]

    # ---- batrack.py:399-410: all patches of the window's keyframes x all frames of the window
    def _edges(self):
        r = self.cfg.S_slam
        lo = max(self.n - r, 0)
        idx = torch.arange(0, self.n * self.M, device=self.device).reshape(self.n, self.M)
        kf_idx = idx[lo:self.n:self.cfg.kf_stride].reshape(-1)
        frames = torch.arange(lo, self.n, device=self.device)
        return kf_idx.repeat_interleave(frames.numel()), frames.repeat(kf_idx.numel())

    # ---- batrack.py:189-204
    def append_factors(self, kk_new, jj_new):
        self.jj = torch.cat([self.jj, jj_new])
        self.kk = torch.cat([self.kk, kk_new])
        self.ii = torch.cat([self.ii, kk_new // self.M])
        self._kk_new, self._jj_new = kk_new, jj_new

    # ---- batrack.py:206-212
    def remove_factors(self, mask):
        keep = ~mask
        self.ii, self.jj, self.kk = self.ii[keep], self.jj[keep], self.kk[keep]
        self.targets_3d = self.targets_3d[:, keep]
        self.weights = self.weights[:, keep]
        self.weights_pose = self.weights_pose[:, keep]

    # ---- batrack.py:760-795 (labels from the synthetic observation model)
    def predict_target(self):
        kk, jj = self._kk_new.cpu().numpy(), self._jj_new.cpu().numpy()
        t3, vis, static = self.obs.predict(kk, jj)
        S = min(self.n, self.cfg.S_slam)
        w = np.ones((kk.shape[0], 2))
        w[~vis] = 0.0
        pad = 20
        inside = (t3[:, 0] >= pad) & (t3[:, 0] < self.wd - pad) & (t3[:, 1] >= pad) & (t3[:, 1] < self.ht - pad)
        w[~inside] = 0.0
        f32 = dict(dtype=torch.float32, device=self.device)
        if self.n >= self.cfg.MIN_TRACK_LEN:
            seen = (w > 0).any(1).reshape(-1, S).sum(1) >= self.cfg.MIN_TRACK_LEN     # per track of the window
            self.patches_valid_[self.n - S:self.n:self.cfg.kf_stride] = torch.as_tensor(seen.reshape(-1, self.M), **f32)   # batrack.py:783
            w[~np.repeat(seen, S)] = 0.0
        wp = w.copy()
        wp[~static] = 0.0
        t3_t, w_t = torch.as_tensor(t3, **f32)[None], torch.as_tensor(w, **f32)[None]
        self.targets_3d = torch.cat([self.targets_3d, t3_t], 1)
        self.weights = torch.cat([self.weights, w_t], 1)
        self.weights_pose = torch.cat([self.weights_pose, torch.as_tensor(wp, **f32)[None]], 1)
        self.update_local(t3_t, w_t, torch.as_tensor(vis, device=self.device)[None], torch.as_tensor(static, device=self.device)[None])

    # ---- batrack.py:632-663
    def update_local(self, target_3d, weights, vis_e, static_e):
        """The new edges' targets, visibility, static label and weight into the tracks' window buffers, slot
        (jj - ii) + (S_local + 1) // 2 - 1 of the track."""
        ii, jj, kk = self._kk_new // self.M, self._jj_new, self._kk_new
        local_id = (jj - ii) + (self.S_local + 1) // 2 - 1
        ok = (local_id >= 0) & (local_id < self.S_local)
        kv, lv = kk[ok], local_id[ok]
        self.patches_local_.view(self.N * self.M, self.S_local, 3)[kv, lv] = target_3d[0, ok]
        self.patches_local_vis_.view(self.N * self.M, self.S_local, 1)[kv, lv] = vis_e[0, ok][:, None].float()
        self.patches_local_static_.view(self.N * self.M, self.S_local, 1)[kv, lv] = static_e[0, ok][:, None].float()
        self.patches_local_weights_.view(self.N * self.M, self.S_local, 1)[kv, lv] = weights[0, ok][:, :1]

    # ---- batrack.py:327-338
    def map_point_filtering(self):
        coords = pops.transform(self.SE3(self.poses), self.patches, self.intrinsics, self.ii, self.jj, self.kk)
        err = torch.norm(coords[:, :, 0, 0] - self.targets_3d[..., :2], dim=-1)
        bad = ~(err < self.cfg.MAP_FILTERING_TH)
        self.stats["filtered"] += int((bad & (self.weights[..., 0] > 0)).sum())
        self.weights[bad] = 0
        self.weights_pose[bad] = 0

    # ---- batrack.py:856-895
    def update(self):
        c = self.cfg
        t0 = max(self.n - c.OPTIMIZATION_WINDOW if self.is_initialized else 1, 1)
        ep, lmbda = 10, 1e-4
        bounds = [0, 0, self.wd, self.ht]
        Gs = self.SE3(self.poses)
        patches = self.patches
        # the depth prior is the track's own slot of its window buffer, a strided view consumed in place (batrack.py:866)
        mono = self.patches_local_.view(1, self.N * self.M, self.S_local, 3)[:, :, (self.S_local + 1) // 2 - 1, 2:]
        self._sync()
        tic = time.perf_counter()
        for _ in range(c.ITER):
            Gs, patches = self.ba(Gs, patches, mono, self.intrinsics, self.targets_3d[..., :2], self.targets_3d[..., 2:],
                                  self.weights_pose, lmbda, self.ii, self.jj, self.kk, bounds, ep=ep, fixedp=t0,
                                  structure_only=False, loss=c.LOSS, alpha=0.05)
            Gs, patches = self.ba(Gs, patches, mono, self.intrinsics, self.targets_3d[..., :2], self.targets_3d[..., 2:],
                                  self.weights, lmbda, self.ii, self.jj, self.kk, bounds, ep=ep, fixedp=t0,
                                  structure_only=True, loss=c.LOSS, alpha=0.05)
        self._sync()
        self.stats["ba_seconds"] += time.perf_counter() - tic
        self.stats["ba_calls"] += 2 * c.ITER
        self.stats["updates"] += 1
        self.stats["edges_max"] = max(self.stats["edges_max"], int(self.ii.numel()))
        self.patches_[:] = patches.reshape(self.N, self.M, 3, 1, 1)
        self.poses_[:] = Gs.vec().reshape(self.N, 7)
        if c.USE_MAP_FILTERING:
            self.map_point_filtering()

    def _prefetch(self, n_at_update):
        """Hand the edge list of the update() that will run with `n_at_update` frames to the plan builder."""
        c = self.cfg
        if self.prefetch is None or self.ii.numel() == 0:
            return
        if not (self.is_initialized or n_at_update == c.num_init + 1):
            return                                              # no update() before the initialisation frame
        t0 = max(n_at_update - c.OPTIMIZATION_WINDOW, 1)        # update(): is_initialized is set by then
        self.prefetch(self.ii, self.jj, self.kk, self.N, self.N * self.M, t0, self.device)

    # ---- batrack.py:1020-1024
    def keyframe_simple(self):
        self.remove_factors(self.kk // self.M < self.n - self.cfg.REMOVAL_WINDOW)

    # ---- batrack.py:937-1008
    def __call__(self):
        if self.n + 1 >= self.N:
            raise RuntimeError("The buffer size is too small")
        c = self.cfg
        pat, prior = self.obs.frame_patches(self.n)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.patches_[self.n] = torch.as_tensor(pat, **f32).view(self.M, 3, 1, 1)
        if self.n % c.kf_stride == 0 and not self.is_initialized:
            self.patches_valid_[self.n] = 1                                       # batrack.py:968-969
        self.init_motion()
        self.tlist.append(float(self.n))
        self.tstamps_[self.n] = self.counter
        self.counter += 1
        self.n += 1
        self.m += self.M
        if (self.n - 1) % c.kf_stride == 0:
            self.append_factors(*self._edges())
            self._prefetch(self.n)
            self.predict_target()
        if self.n == c.num_init + 1 and not self.is_initialized:
            self.is_initialized = True
            for _ in range(c.init_updates):
                self.update()
        elif self.is_initialized:
            self.update()
            self.keyframe_simple()
            if self.n % c.kf_stride != 0:                       # the next frame appends nothing: its edge list is final now
                self._prefetch(self.n + 1)

    # ---- batrack.py:1080-1135: the hand-off to the global-alignment stage (results.pkl)
    def get_results(self, rgbs=None, dmaps=None, dmaps_gt=None, save_path=None):
        """The reference's results dictionary, all eleven keys with its shapes and dtypes (batrack.py:1113-1125):
        `cams_T_world` [T,4,4] float32 (inverse pose matrices), `intrinsics` [T,4] float32, `tstamps` [T] float64,
        `trajs_2d_disp` [T,M,S_local,3] float32 (a track's (u, v, disparity) targets in the S_local frames around its own),
        `trajs_valid` [T,M] bool (some weight > 0 in the window), `trajs_static`, `trajs_vis` [T,M,S_local] float32,
        `grid_query_frames` (the frames with a valid patch), `dmaps` / `rgbs` / `dmaps_gt` as float64 arrays of what the caller
        hands in (None stays None).  T = frames seen; no frame is ever dropped from the buffer by `keyframe_simple`, so the
        reference's per-time-stamp pose lookup (`get_pose`) is the buffer itself.  Pickled to `save_path` if given."""
        T = self.counter
        G = self.SE3(self.poses_[:T])
        pts_valid = self.patches_valid_[:T].detach().cpu().numpy()
        trajs_valid = self.patches_local_weights_[:T, ..., 0]
        results = {
            "cams_T_world": G.inv().matrix().detach().cpu().numpy(),
            "intrinsics": self.intrinsics_[:T].detach().cpu().numpy(),
            "tstamps": np.array(self.tlist, dtype=float),
            "trajs_2d_disp": self.patches_local_[:T].detach().cpu().numpy(),
            "trajs_valid": (trajs_valid.sum(dim=2) > 0).detach().cpu().numpy(),
            "trajs_static": self.patches_local_static_[:T, ..., 0].detach().cpu().numpy(),
            "trajs_vis": self.patches_local_vis_[:T, ..., 0].detach().cpu().numpy(),
            "grid_query_frames": np.arange(T)[pts_valid.sum(axis=1) > 0],
            "dmaps": None if dmaps is None else np.array(dmaps, dtype=float),
            "rgbs": None if rgbs is None else np.array(rgbs, dtype=float),
            "dmaps_gt": None if dmaps_gt is None else np.array(dmaps_gt, dtype=float),
        }
        if save_path is not None:
            import pickle
            with open(save_path, "wb+") as f:
                pickle.dump(results, f)
        return results

    def run(self, n_frames=None):
        for _ in range(self.obs.n_frames if n_frames is None else n_frames):
            self()
        return self.poses_[:self.n].detach().cpu().numpy().astype(np.float64)
