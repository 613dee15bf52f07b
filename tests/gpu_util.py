"""Helpers for the -m gpu tests: run the HIP path (through the C ABI) on numpy inputs."""
import os

import numpy as np
import torch

from batrack_amd.backend.ba import BA_rgbd_droid
from batrack_amd.backend.lietorch import SE3
from batrack_amd.plan import Plan, Stepper

DEV = "cuda:0"

# Precision of the per-edge maths of the process under test: graphs that take k_tile — every graph a real
# window produces, every fixture — run float64 per edge; BT_FORCE prec=f32 or a forced k_stream / k_edge2 (the mixed-precision
# kernels of >= 2048-tile graphs) run float32 per edge like the reference's own float32 run.
import force as _force  # noqa: E402  (tests/force.py)
F32_EDGE = _force.f32_edges()
# relative tolerances against the reference's float64 result: state (poses', disparities'), reduced system (S, y), camera
# update dX, and the UPDATE itself over the touched poses / tracks (north_star: <= 1e-5).
#   float64 per edge, measured on the fixtures: S, y <= 7e-14; dX <= 2.6e-8 (it is stored as float32); update poses
#   <= 1.3e-6, disparities <= 2.8e-7 (the float32 rounding of the state they are written to); state <= 2e-8
#   float32 per edge (round 2): S, y <= 2.6e-6, dX <= 1.6e-4, update <= 1.6e-4 / 5.2e-5, state <= 2.7e-6
#   (forced onto the ill-conditioned 8-frame fixtures the mixed kernels measure 5.2e-6 on the state since round 6's tile layout — one
#    source frame per tile, other float32 summation chains —, 4.9e-6 before: the gate is for these forced runs only)
TOL = (dict(state=8e-6, sys=4e-6, dx=3e-4, upd_pose=3e-4, upd_disp=1e-4) if F32_EDGE else
       dict(state=2e-7, sys=1e-10, dx=1e-5, upd_pose=1e-5, upd_disp=1e-5))


def t32(a):
    return torch.as_tensor(np.asarray(a, np.float32), device=DEV)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def update_err(out, ref_out, inp, rows=None):
    """Relative error of the UPDATE (out - inp against ref_out - inp) over the rows the step touched: the
    north-star tolerance is on the pose / depth update, and a state norm dominated by unit quaternions and O(1)
    translations would hide two orders of magnitude of it."""
    out, ref_out, inp = (np.asarray(a, np.float64) for a in (out, ref_out, inp))
    if rows is not None:
        out, ref_out, inp = out[rows], ref_out[rows], inp[rows]
    du, dr = out - inp, ref_out - inp
    return np.linalg.norm(du - dr) / max(np.linalg.norm(dr), 1e-300)


class HipProblem:
    """Device copies of a golden/generated input dict, laid out as the caller holds them."""

    def __init__(self, d):
        self.poses = t32(d["poses"])[None]                       # [1,N,7]
        self.patches = t32(d["patches"])[None, :, :, None, None]  # [1,P,3,1,1]
        self.mono = t32(d["mono"])[None, :, None]                 # [1,P,1]
        self.intr = t32(d["intrinsics"])[None]
        self.t3 = t32(d["targets3"])[None]                        # [1,E,3]: 2-D target is a stride-3 view
        self.w = {k: t32(d[k])[None] for k in ("weights", "weights_pose")}
        self.ii, self.jj, self.kk = (torch.as_tensor(d[k], device=DEV) for k in ("ii", "jj", "kk"))
        self.bounds = [float(v) for v in d["bounds"]]

    def api_step(self, wkey, fixedp, so, loss="huber", poses=None, patches=None, lmbda=1e-4, ep=10.0, alpha=0.05):
        """Through BA_rgbd_droid exactly as batrack.py:871-875 calls it."""
        Gs = SE3(self.poses) if poses is None else poses
        pat = self.patches if patches is None else patches
        return BA_rgbd_droid(Gs, pat, self.mono, self.intr, self.t3[..., :2], self.t3[..., 2:], self.w[wkey],
                             lmbda, self.ii, self.jj, self.kk, self.bounds, ep=ep, fixedp=fixedp,
                             structure_only=so, loss=loss, alpha=alpha)

    def raw_step(self, wkey, fixedp, so=False, loss="huber", lmbda=1e-4, ep=10.0, alpha=0.05):
        """Through Plan/Stepper, returning the reduced system as well."""
        plan = Plan(self.ii, self.jj, self.kk, self.poses.shape[1], self.patches.shape[1], fixedp)
        st = Stepper(plan, DEV)
        P = self.poses[0].contiguous()
        pat = self.patches.reshape(-1, 3).contiguous()
        so = so or plan.n == 0
        pout = torch.empty_like(pat)
        Pout = P if so else torch.empty_like(P)
        tg = self.t3[0]
        args = (P, pat, self.mono.reshape(-1), self.intr[0], tg, tg.stride(0), self.w[wkey][0].contiguous(),
                Pout, pout, self.bounds, lmbda, ep, alpha, loss, so)
        st.step(*args, phase="reduce")                 # the step, split where multi-GPU all-reduces,
        torch.cuda.synchronize()                       # so the reduced system can be inspected
        sysv = st.system.cpu().numpy().copy()
        st.step(*args, phase="solve_update")
        torch.cuda.synchronize()
        assert not so and float(st.system.abs().max()) == 0.0 or so   # accumulators left clear
        out = dict(poses_out=Pout.cpu().numpy(), patches_out=pout.cpu().numpy(), plan=plan, stepper=st)
        if not so:
            D = 6 * plan.n
            out["S_lower"] = sysv[:D * D].reshape(D, D)
            out["y"] = sysv[D * D:]
            out["dX"] = st.dx.cpu().numpy()
            out["status"] = st.status()
        return out
