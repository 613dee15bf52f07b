"""Losses of the reference's dense global-alignment stage and their gradients on the gfx950 kernels.

`RefineLosses` holds what `RefineNet` (/root/reference/main/global_refine/model/refine_net.py) holds after its
`_init_from_ba` — same attribute names (`trajs_2d`, `trajs_disp`, `trajs_disp_mono`, `trajs_vis`, `trajs_static`, `jj`,
`intrinsics`, `grid_query_frames`, parameters `trajs_scales`, `frame_scales_`, `frame_shifts_`, `pose`, `K`; settings
`loss_weight_dict`, `alpha`, `scale_smoothness_weight`, `refine_intrinsics`, `K_scale`) — and evaluates
  get_frame_scaled_depth()          refine_net.py:148-174
  spatial_loss()                    the huber depth term of forward(), refine_net.py:252-268
  inter_frame_loss()                refine_net.py:199-225 (the O(Q S N^2) rigidity term)
  pts_3d_loss()                     refine_net.py:314-354
  cam_smooth_vec_loss()             refine_net.py:356-360
  scale_grid_smoothness_loss(mode)  refine_net.py:362-392 ('l1' | 'l2' | 'huber')
  forward()                         the total the reference optimises: the `loss_weight_dict` branch (refine_net.py:274-297;
                                    run_global_refine.py:61-67 always passes one) or, with `loss_weight_dict = None`,
                                    spatial + alpha * rigid + scale_smoothness_weight * smoothness('l1') (:299-301)
  backward()                        its gradients w.r.t. every parameter the reference's Adam loop steps (trainer.py:33-43):
                                    `trajs_scales`, `frame_scales_`, `pose` (pypose's left-perturbation convention, see
                                    include/batrack_ga.h) and `K`
  loss()                            forward() as a torch.autograd node, so that the reference's loop body
                                    `loss = net.loss(); loss.backward(); optimizer.step()` fills `.grad` of the parameters
through include/batrack_ga.h.  `half_disp=True` keeps the two disparity arrays in float16 and forms the depth residual in
float16 (BASELINE.json configs[4]).  GPU tensors only; there is no CPU fallback.
"""
import ctypes

import torch

from . import _lib

_SMOOTH = {"l1": 0, "l2": 1, "huber": 2}


def _align_depth_maps(depth_maps):
    """model/utils.py:268-312 (`align_depth_maps`, used when RefineNet is built with align_depth=True): every depth map is
    scaled so that its median over the overlap with its aligned predecessor matches the median of the previous (two) aligned
    maps.  A one-off pass over numpy arrays at construction, on the host as in the reference."""
    import numpy as np
    S = depth_maps.shape[0]
    out = np.zeros_like(depth_maps)
    out[0] = depth_maps[0]
    for i in range(1, S):
        prev, cur = out[i - 1, ..., 0], depth_maps[i, ..., 0]
        mask = (prev > 0) & (cur > 0)
        if mask.sum() < 100:                                   # min_overlap_threshold
            out[i, ..., 0] = cur
            continue
        if i == 1:
            med_prev = np.median(prev[mask])
        else:
            past = out[i - 2, ..., 0]
            med_prev = np.median(np.concatenate((past[(past > 0) & (prev > 0)], prev[mask])))
        out[i, ..., 0] = med_prev / np.median(cur[mask]) * cur
    return out


class RefineLosses:
    def __init__(self, trajs_2d, trajs_disp, trajs_disp_mono, trajs_vis, trajs_static, jj, intrinsics, grid_query_frames,
                 trajs_scales, frame_scales_, frame_shifts_, pose, H, W, pw_break=20.0, half_disp=False,
                 loss_weight_dict=None, alpha=0.5, scale_smoothness_weight=0.1, scale_smoothness_mode="l2",
                 refine_intrinsics=False, K=None, K_scale=20.0):
        dev = trajs_2d.device
        if dev.type != "cuda":
            raise RuntimeError("RefineLosses: tensors must be on the GPU (no CPU fallback in batrack_amd)")
        f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        self.T, self.N, self.S_local = trajs_disp.shape
        self.half_disp = bool(half_disp)
        dd = torch.float16 if half_disp else torch.float32
        self.trajs_2d = f(trajs_2d)
        self.trajs_disp = trajs_disp.to(device=dev, dtype=dd).contiguous()
        self.trajs_disp_mono = trajs_disp_mono.to(device=dev, dtype=dd).contiguous()
        self.trajs_vis, self.trajs_static = f(trajs_vis), f(trajs_static)
        self.jj = jj.to(device=dev, dtype=torch.int64).contiguous()
        self.intrinsics_raw = f(intrinsics)
        self.grid_query_frames = grid_query_frames.to(device=dev, dtype=torch.int64).contiguous()
        q = self.grid_query_frames
        if q.numel() == 0 or int(q.min()) < 0 or int(q.max()) >= self.T:
            raise ValueError("grid_query_frames must be frame numbers in [0, T)")
        # `loss[grid_query_frames].mean()` (refine_net.py:265,223) counts a frame that is listed twice twice.  The kernels take
        # lists of DISTINCT frames, so a list with repeats is split into layers — layer k = the frames listed more than k times —
        # and the query-dependent terms (spatial, inter-frame) are the layers' means weighted by Q_k / Q.  BATRACK.get_results
        # never produces a repeat (batrack.py:1095); one layer, weight 1, is the usual case.
        frames, counts = torch.unique(q, return_counts=True)
        self._query_layers = [(frames[counts > k].contiguous(), float((counts > k).sum()) / q.numel()) for k in range(int(counts.max()))]
        self.trajs_scales, self.frame_scales_, self.frame_shifts_, self.pose = f(trajs_scales), f(frame_scales_), f(frame_shifts_), f(pose)
        self.H, self.W, self.pw_break = int(H), int(W), float(pw_break)
        if tuple(self.trajs_2d.shape) != (self.T, self.N, self.S_local, 2) or tuple(self.jj.shape) != (self.T, self.S_local):
            raise ValueError("trajs_2d must be [T,N,S,2] and jj [T,S]")
        # what RefineNet.__init__ takes (refine_net.py:16-48)
        self.loss_weight_dict = None if loss_weight_dict is None else dict(loss_weight_dict)
        self.alpha, self.scale_smoothness_weight, self.scale_smoothness_mode = float(alpha), float(scale_smoothness_weight), scale_smoothness_mode
        self.refine_intrinsics, self.K_scale = bool(refine_intrinsics), float(K_scale)
        self.K = f(K) if K is not None else (torch.median(self.intrinsics_raw, dim=0)[0] / self.K_scale).clone()   # K_init, refine_net.py:77
        self._lib = _lib.lib()
        self._intr = torch.empty(self.T, 4, device=dev, dtype=torch.float32)
        self._mono_scaled = torch.empty(self.T, self.N, self.S_local, device=dev, dtype=torch.float32)
        self._losses = torch.zeros(5, device=dev, dtype=torch.float64)
        self._g_ms = None

    # ------------------------------------------------------------------ the hand-off from the sparse-SLAM stage
    @classmethod
    def from_results(cls, results, device="cuda:0", grid_size=4, pw_break=20.0, align_depth=False, loss_weight_dict=None,
                     refine_intrinsics=False, alpha=0.5, scale_smoothness_weight=0.1, scale_smoothness_mode="l2", half_disp=False,
                     K_scale=20.0):
        """What `RefineNet.__init__` + `_init_from_ba` do with a results.pkl (refine_net.py:16-121), on the GPU.  `results`: the
        dictionary `BATRACK.get_results` / `batrack_amd.sequence.WindowedBA.get_results` returns, or the path of its pickle.
          cams_T_world [T,4,4] -> `pose` by bt_ga_mat_to_se3 (`pp.mat2SE3`, refine_net.py:61)
          intrinsics [T,4] -> `intrinsics_raw`; `K` = their (lower) median / K_scale (refine_net.py:77)
          trajs_2d_disp [T,N,S,3] -> `trajs_2d`, `trajs_disp`; trajs_static, trajs_vis, grid_query_frames as they are
          jj = t + s - S // 2 (refine_net.py:92-97)
          dmaps [T,H,W,1] -> `trajs_disp_mono` [T,N,S] = 1 / max(bilinear sample of frame clamp(jj) at the track, 1e-2) by
                             bt_ga_sample_disp_mono (refine_net.py:99-110); `align_depth=True` first rescales the maps by
                             their running medians as model/utils.py:268-312 does (on the host, where the reference does it).
                             The maps are sampled in FLOAT32 (the reference keeps results['dmaps'] float64 and samples and
                             inverts in float64): `trajs_disp_mono` is within ~1e-7 relative of the reference's, tested at
                             1e-5 of its largest entry — the losses it enters are float32 anyway
          `pose`: as a rotation and a translation it is cams_T_world's; the quaternion's SIGN is the kernel's convention (w >= 0
                             from the trace branch, i.e. for every rotation below 180 degrees) — pypose is not in this image,
                             so its convention is unpinned; the camera-smoothness term reads the stored numbers and is
                             invariant under a common sign, not under the sign of one pose
          parameters at the reference's initial values: trajs_scales = 1, frame_scales_ = 1 [T,gh,gw], frame_shifts_ = 0.
        `trajs_valid` is kept as an attribute (the reference reads it into `self.trajs_valid` and never uses it in a loss)."""
        import numpy as np
        if isinstance(results, (str, bytes)) or hasattr(results, "__fspath__"):
            import pickle
            with open(results, "rb") as f:
                results = pickle.load(f)
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("RefineLosses.from_results: the target device must be a GPU (no CPU fallback in batrack_amd)")
        if results.get("dmaps") is None:
            raise ValueError("results['dmaps'] is required: the mono-depth maps the tracks' prior disparities are sampled from")
        L = _lib.lib()
        st = torch.cuda.current_stream(dev).cuda_stream
        dm = np.asarray(results["dmaps"])
        if dm.ndim != 4 or dm.shape[-1] < 1:
            raise ValueError("results['dmaps'] must be [T,H,W,C]")
        if align_depth:
            dm = _align_depth_maps(dm)
        t2d = torch.as_tensor(np.ascontiguousarray(results["trajs_2d_disp"]), dtype=torch.float32, device=dev)
        if t2d.dim() != 4 or t2d.shape[-1] != 3:
            raise ValueError("results['trajs_2d_disp'] must be [T,N,S_local,3]")
        T, N, S, _ = t2d.shape
        if dm.shape[0] != T:
            raise ValueError("results['dmaps'] and results['trajs_2d_disp'] disagree about the number of frames")
        H, W = int(dm.shape[1]), int(dm.shape[2])
        dmaps = torch.as_tensor(np.ascontiguousarray(dm[..., 0]), dtype=torch.float32, device=dev)
        trajs_2d = t2d[..., :2].contiguous()
        mono = torch.empty(T, N, S, device=dev, dtype=torch.float32)
        _lib.check(L.bt_ga_sample_disp_mono(dmaps.data_ptr(), trajs_2d.data_ptr(), mono.data_ptr(), T, N, S, H, W, st), "bt_ga_sample_disp_mono")
        cams = torch.as_tensor(np.ascontiguousarray(results["cams_T_world"]), dtype=torch.float32, device=dev).reshape(-1, 16)
        if cams.shape[0] != T:
            raise ValueError("results['cams_T_world'] must be [T,4,4]")
        pose = torch.empty(T, 7, device=dev, dtype=torch.float32)
        _lib.check(L.bt_ga_mat_to_se3(cams.data_ptr(), pose.data_ptr(), T, st), "bt_ga_mat_to_se3")
        gh, gw = (grid_size if isinstance(grid_size, (list, tuple)) else (grid_size, grid_size))
        jj = torch.arange(T, device=dev)[:, None] + torch.arange(S, device=dev)[None] - S // 2
        f = lambda k: torch.as_tensor(np.ascontiguousarray(results[k]), device=dev)
        net = cls(trajs_2d, t2d[..., 2].contiguous(), mono, f("trajs_vis"), f("trajs_static"), jj, f("intrinsics"),
                  f("grid_query_frames"), torch.ones(T, N, S, device=dev), torch.ones(T, int(gh), int(gw), device=dev),
                  torch.zeros(T, device=dev), pose, H, W, pw_break=pw_break, half_disp=half_disp, loss_weight_dict=loss_weight_dict,
                  alpha=alpha, scale_smoothness_weight=scale_smoothness_weight, scale_smoothness_mode=scale_smoothness_mode,
                  refine_intrinsics=refine_intrinsics, K_scale=K_scale)
        net.trajs_valid = f("trajs_valid")
        net.results = results
        return net

    @property
    def intrinsics(self):
        """refine_net.py:131-136: K * K_scale for every frame when the intrinsics are refined, else the per-frame input."""
        if self.refine_intrinsics:
            return (self.K.detach() * self.K_scale).expand(self.T, 4)
        return self.intrinsics_raw

    def _args(self, layer=0):
        a = _lib.GaArgs()
        a.T, a.N, a.S = self.T, self.N, self.S_local
        a.gh, a.gw = self.frame_scales_.shape[1:]
        query = self._query_layers[layer][0]
        a.H, a.W, a.Q = self.H, self.W, query.numel()
        self._intr.copy_(self.intrinsics)
        for n, t in (("trajs_2d", self.trajs_2d), ("trajs_disp", self.trajs_disp), ("trajs_disp_mono", self.trajs_disp_mono),
                     ("trajs_vis", self.trajs_vis), ("trajs_static", self.trajs_static), ("jj", self.jj), ("intrinsics", self._intr),
                     ("pose", self.pose), ("query", query), ("trajs_scales", self.trajs_scales),
                     ("frame_scales", self.frame_scales_), ("frame_shifts", self.frame_shifts_)):
            if not (t.is_cuda and t.is_contiguous()):
                raise RuntimeError(f"RefineLosses: `{n}` must be a contiguous GPU tensor")
            setattr(a, n, t.data_ptr())
        a.pw_break, a.half_disp = self.pw_break, 1 if self.half_disp else 0
        return a

    # ------------------------------------------------------------------ the total and its weights
    def weights(self, alpha=None):
        """(spatial, rigid, pts3d, cam_smooth, scale_smooth) of forward()'s total and the mode of the smoothness term — 'l1'
        in both branches of forward(), whatever `scale_smoothness_mode` says (refine_net.py:289,299)."""
        if self.loss_weight_dict is not None:
            d = self.loss_weight_dict
            w = [d.get("spatial_loss", 0.0), d.get("inter_frame_loss", 0.0), d.get("pts_3d_loss", 0.0),
                 d.get("cam_smooth_vec_loss", 0.0), d.get("scale_smoothness_loss", 0.0)]
            if self.alpha <= 0:                       # (loss_rigid is the constant 0.0 then, refine_net.py:270-273)
                w[1] = 0.0
        else:
            al = self.alpha if alpha is None else float(alpha)
            w = [1.0, al if al > 0 else 0.0, 0.0, 0.0, self.scale_smoothness_weight if self.scale_smoothness_weight > 0 else 0.0]
        return [float(x) for x in w], "l1"

    def _which(self, w, mode):
        return 1 | (2 if w[1] else 0) | (4 if w[2] else 0) | ((8 | (_SMOOTH[mode] << 8)) if (w[3] or w[4]) else 0)

    def forward(self, alpha=None):
        """RefineNet.forward(): the weighted total as a float64 scalar tensor.  `alpha` overrides `self.alpha` in the
        `loss_weight_dict = None` branch (kept from round 2's interface)."""
        w, mode = self.weights(alpha)
        l = self._run(self._which(w, mode))
        return sum(wi * l[i] for i, wi in enumerate(w) if wi)

    def backward(self, alpha=None, want=("trajs_scales", "frame_scales_", "pose", "K")):
        """Gradients of forward() as a dict of new float32 tensors: `trajs_scales` [T,N,S], `frame_scales_` [T,gh,gw],
        `pose` [T,7] (pypose's convention: left-perturbation gradient in the first six numbers, plus the plain derivative of
        the camera-smoothness term), `intrinsics` [T,4] per frame and `K` [4] = K_scale * their sum (refine_net.py:131-136)."""
        w, mode = self.weights(alpha)
        self._run(1)                                                      # mono_scaled for the current parameters
        if self._g_ms is None:
            self._g_ms = torch.empty_like(self._mono_scaled)
        dev = self.trajs_2d.device
        g_ts, g_fs = torch.empty_like(self._mono_scaled), torch.empty_like(self.frame_scales_, dtype=torch.float32)
        need_pose = (w[2] or w[3]) and "pose" in want
        need_k = (w[1] or w[2]) and ("K" in want or "intrinsics" in want)
        g_pose = torch.empty(self.T, 7, device=dev, dtype=torch.float32) if need_pose else None
        g_intr = torch.empty(self.T, 4, device=dev, dtype=torch.float32) if need_k else None
        st = torch.cuda.current_stream(dev).cuda_stream
        for k, (_, frac) in enumerate(self._query_layers):
            # (a list with repeated frames: the query-dependent terms once per layer at Q_k / Q of their weight, the rest with layer 0)
            wk = [w[0] * frac, w[1] * frac] + (list(w[2:]) if k == 0 else [0.0, 0.0, 0.0])
            tgt = (g_ts, g_fs, g_pose, g_intr) if k == 0 else tuple(None if t is None else torch.empty_like(t) for t in (g_ts, g_fs, g_pose, g_intr))
            a = self._args(k)
            gw = _lib.GaWeights(*wk, _SMOOTH[mode])
            _lib.check(self._lib.bt_ga_backward_total(ctypes.byref(a), self._mono_scaled.data_ptr(), ctypes.byref(gw), self._g_ms.data_ptr(),
                                                      tgt[0].data_ptr(), tgt[1].data_ptr(), tgt[2].data_ptr() if need_pose else None,
                                                      tgt[3].data_ptr() if need_k else None, st), "bt_ga_backward_total")
            if k > 0:
                for acc, t in zip((g_ts, g_fs, g_pose, g_intr), tgt):
                    if acc is not None:
                        acc += t
        out = {"trajs_scales": g_ts, "frame_scales_": g_fs}
        out["pose"] = g_pose if need_pose else torch.zeros(self.T, 7, device=dev)
        out["intrinsics"] = g_intr if need_k else torch.zeros(self.T, 4, device=dev)
        out["K"] = out["intrinsics"].sum(0) * self.K_scale if self.refine_intrinsics else torch.zeros(4, device=dev)
        return out

    def loss(self, alpha=None):
        """forward() as a float32 scalar attached to autograd: `.backward()` fills `.grad` of `self.trajs_scales`,
        `self.frame_scales_`, `self.pose` and `self.K` where they are leaves that require grad (the reference's
        nn.Parameters / pp.Parameter, refine_net.py:42-48)."""
        return _TotalLoss.apply(self.trajs_scales, self.frame_scales_, self.pose, self.K, self, alpha)

    # ------------------------------------------------------------------ the terms
    def _run(self, which):
        st = torch.cuda.current_stream(self.trajs_2d.device).cuda_stream
        total = None
        for k, (_, frac) in enumerate(self._query_layers):
            a = self._args(k)
            _lib.check(self._lib.bt_ga_forward(ctypes.byref(a), self._mono_scaled.data_ptr(), self._losses.data_ptr(),
                                               int(which if k == 0 else which & 3), st), "bt_ga_forward")
            if len(self._query_layers) == 1:
                return self._losses
            if k == 0:
                total = self._losses.clone()
                total[:2] *= frac
            else:
                total[:2] += frac * self._losses[:2]
        return total

    def get_frame_scaled_depth(self):
        self._run(1)
        return self._mono_scaled

    def spatial_loss(self):
        return self._run(1)[0].clone()

    def inter_frame_loss(self):
        return self._run(3)[1].clone()

    def pts_3d_loss(self):
        return self._run(5)[2].clone()

    def cam_smooth_vec_loss(self):
        return self._run(9)[3].clone()

    def scale_grid_smoothness_loss(self, mode="l2"):
        if mode not in _SMOOTH:
            raise ValueError(f"Unknown smoothness loss mode: {mode}")        # refine_net.py:387
        return self._run(9 | (_SMOOTH[mode] << 8))[4].clone()

    def losses(self, smooth_mode="l1"):
        """(spatial, inter-frame, 3-D points, camera smoothness, scale-grid smoothness) in one pass, as a float64 tensor of 5."""
        return self._run(15 | (_SMOOTH[smooth_mode] << 8)).clone()


class _TotalLoss(torch.autograd.Function):
    """forward()'s weighted total as one autograd node over bt_ga_forward / bt_ga_backward_total."""

    @staticmethod
    def forward(ctx, trajs_scales, frame_scales_, pose, K, net, alpha):
        ctx.net, ctx.alpha = net, alpha
        return net.forward(alpha).to(torch.float32)

    @staticmethod
    def backward(ctx, gout):
        g = ctx.net.backward(ctx.alpha)
        need = ctx.needs_input_grad
        return (g["trajs_scales"] * gout if need[0] else None, g["frame_scales_"] * gout if need[1] else None,
                g["pose"] * gout if need[2] else None, g["K"] * gout if need[3] else None, None, None)
