"""Losses of the reference's dense global-alignment stage and their gradients on the gfx950 kernels.

`RefineLosses` holds what `RefineNet` (/root/reference/main/global_refine/model/refine_net.py) holds after its
`_init_from_ba` — same attribute names (`trajs_2d`, `trajs_disp`, `trajs_disp_mono`, `trajs_vis`, `trajs_static`, `jj`,
`intrinsics`, `grid_query_frames`, parameters `trajs_scales`, `frame_scales_`, `frame_shifts_`, `pose`) — and evaluates
  get_frame_scaled_depth()   refine_net.py:148-174
  spatial_loss()             the huber depth term of forward(), refine_net.py:252-268
  inter_frame_loss()         refine_net.py:199-225 (the O(Q S N^2) rigidity term)
  pts_3d_loss()              refine_net.py:300-345
  forward(alpha)             total of refine_net.py:291-293 (loss_weight_dict = None, no scale-grid smoothness)
  backward(alpha) / loss(alpha)   gradients of forward(alpha) w.r.t. `trajs_scales` and `frame_scales_` (what the reference
                             gets from autograd and steps with Adam, trainer.py:23-77): explicit, or as a torch.autograd
                             node so that `net.loss(alpha).backward()` fills `.grad` of the two parameters
through include/batrack_ga.h.  pts_3d_loss (poses, intrinsics) is forward only.
`half_disp=True` keeps the two disparity arrays in float16 and forms the depth residual in float16 (BASELINE.json
configs[4]).  GPU tensors only; there is no CPU fallback.
"""
import ctypes

import torch

from . import _lib


class RefineLosses:
    def __init__(self, trajs_2d, trajs_disp, trajs_disp_mono, trajs_vis, trajs_static, jj, intrinsics, grid_query_frames,
                 trajs_scales, frame_scales_, frame_shifts_, pose, H, W, pw_break=20.0, half_disp=False):
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
        self.intrinsics = f(intrinsics)
        self.grid_query_frames = grid_query_frames.to(device=dev, dtype=torch.int64).contiguous()
        self.trajs_scales, self.frame_scales_, self.frame_shifts_, self.pose = f(trajs_scales), f(frame_scales_), f(frame_shifts_), f(pose)
        self.H, self.W, self.pw_break = int(H), int(W), float(pw_break)
        if tuple(self.trajs_2d.shape) != (self.T, self.N, self.S_local, 2) or tuple(self.jj.shape) != (self.T, self.S_local):
            raise ValueError("trajs_2d must be [T,N,S,2] and jj [T,S]")
        self._lib = _lib.lib()
        self._mono_scaled = torch.empty(self.T, self.N, self.S_local, device=dev, dtype=torch.float32)
        self._losses = torch.zeros(3, device=dev, dtype=torch.float64)
        self._g_ms = None

    def _args(self):
        a = _lib.GaArgs()
        a.T, a.N, a.S = self.T, self.N, self.S_local
        a.gh, a.gw = self.frame_scales_.shape[1:]
        a.H, a.W, a.Q = self.H, self.W, self.grid_query_frames.numel()
        for n, t in (("trajs_2d", self.trajs_2d), ("trajs_disp", self.trajs_disp), ("trajs_disp_mono", self.trajs_disp_mono),
                     ("trajs_vis", self.trajs_vis), ("trajs_static", self.trajs_static), ("jj", self.jj), ("intrinsics", self.intrinsics),
                     ("pose", self.pose), ("query", self.grid_query_frames), ("trajs_scales", self.trajs_scales),
                     ("frame_scales", self.frame_scales_), ("frame_shifts", self.frame_shifts_)):
            if not (t.is_cuda and t.is_contiguous()):
                raise RuntimeError(f"RefineLosses: `{n}` must be a contiguous GPU tensor")
            setattr(a, n, t.data_ptr())
        a.pw_break, a.half_disp = self.pw_break, 1 if self.half_disp else 0
        return a

    def backward(self, alpha=0.5):
        """(d forward(alpha) / d trajs_scales [T,N,S], d forward(alpha) / d frame_scales_ [T,gh,gw]) as new float32 tensors."""
        self._run(1)                                                      # mono_scaled for the current parameters
        if self._g_ms is None:
            self._g_ms = torch.empty_like(self._mono_scaled)
        g_ts, g_fs = torch.empty_like(self._mono_scaled), torch.empty_like(self.frame_scales_, dtype=torch.float32)
        a = self._args()
        st = torch.cuda.current_stream(self.trajs_2d.device).cuda_stream
        _lib.check(self._lib.bt_ga_backward(ctypes.byref(a), self._mono_scaled.data_ptr(), 1.0, float(alpha), self._g_ms.data_ptr(),
                                            g_ts.data_ptr(), g_fs.data_ptr(), st), "bt_ga_backward")
        return g_ts, g_fs

    def loss(self, alpha=0.5):
        """forward(alpha) as a float32 scalar attached to autograd: `.backward()` fills `.grad` of `self.trajs_scales` and
        `self.frame_scales_` when they are leaves that require grad (the reference's nn.Parameters)."""
        return _TotalLoss.apply(self.trajs_scales, self.frame_scales_, self, float(alpha))

    def _run(self, which):
        a = self._args()
        st = torch.cuda.current_stream(self.trajs_2d.device).cuda_stream
        _lib.check(self._lib.bt_ga_forward(ctypes.byref(a), self._mono_scaled.data_ptr(), self._losses.data_ptr(), int(which), st),
                   "bt_ga_forward")
        return self._losses

    def get_frame_scaled_depth(self):
        self._run(1)
        return self._mono_scaled

    def spatial_loss(self):
        return self._run(1)[0].clone()

    def inter_frame_loss(self):
        return self._run(3)[1].clone()

    def pts_3d_loss(self):
        return self._run(5)[2].clone()

    def losses(self):
        """(spatial, inter-frame, 3-D points) in one pass, as a float64 tensor of 3."""
        return self._run(7).clone()

    def forward(self, alpha=0.5):
        l = self._run(3 if alpha > 0 else 1)
        return l[0] + alpha * l[1] if alpha > 0 else l[0].clone()


class _TotalLoss(torch.autograd.Function):
    """spatial + alpha * inter-frame as one autograd node over bt_ga_forward / bt_ga_backward."""

    @staticmethod
    def forward(ctx, trajs_scales, frame_scales_, net, alpha):
        ctx.net, ctx.alpha = net, alpha
        return net.forward(alpha).to(torch.float32)

    @staticmethod
    def backward(ctx, gout):
        g_ts, g_fs = ctx.net.backward(ctx.alpha)
        return g_ts * gout, g_fs * gout, None, None
