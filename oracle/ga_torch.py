"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

torch (float64, CPU) statement of the forward losses of oracle/ga_losses.py, so that torch.autograd gives the gradients of
    total(alpha) = spatial_loss + alpha * inter_frame_loss         (RefineNet.forward, refine_net.py:252-293)
with respect to the two parameters it reaches, `trajs_scales` [T,N,S] and `frame_scales_` [T,gh,gw] — the checker of
bt_ga_backward (include/batrack_ga.h) at sizes other than the fixture's — and (`full_total_and_grads`) of the weighted total
with every term of refine_net.py:274-392 w.r.t. `trajs_scales`, `frame_scales_`, `pose` and `K`: the checker of
bt_ga_backward_total (pinned by tests/golden/ga_total.npz, the reference's own autograd through the pypose stand-in).  Pinned twice: the forward equals the numpy
oracle, and the gradients equal the ones the reference's own autograd produced (tests/golden/ga_small.npz,
`*.grad_*`; tests/test_ga_oracle.py)."""
import numpy as np
import torch


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


def frame_scaled_depth(d, frame_scales_):
    T, N, S = d["trajs_disp_mono"].shape
    g = torch.exp(frame_scales_ / 10.0)
    gh, gw = g.shape[1:]
    jc = torch.as_tensor(np.clip(d["jj"], 0, T - 1))
    xy = _t(d["trajs_2d"])
    x = xy[..., 0] / (int(d["W"]) - 1) * (gw - 1)
    y = xy[..., 1] / (int(d["H"]) - 1) * (gh - 1)
    x0, y0 = torch.floor(x).long(), torch.floor(y).long()
    fx, fy = x - x0, y - y0
    fr = jc[:, None, :].expand(T, N, S)
    out = torch.zeros(T, N, S, dtype=torch.float64)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            yy, xx = y0 + dy, x0 + dx
            ok = (yy >= 0) & (yy < gh) & (xx >= 0) & (xx < gw)
            out = out + torch.where(ok, g[fr, yy.clamp(0, gh - 1), xx.clamp(0, gw - 1)], torch.zeros(())) * wy * wx
    return _t(d["trajs_disp_mono"]) * out + _t(d["frame_shifts"])[jc][:, None, :]


def spatial_loss(d, trajs_scales, ms):
    T, N, S = d["trajs_disp"].shape
    sexp = torch.exp((trajs_scales - trajs_scales.mean(dim=1, keepdim=True)) / float(d["pw_break"]))
    aligned = sexp * _t(d["trajs_disp"])
    mask = _t((d["trajs_vis"] > 0.9) & ((d["jj"] >= 0) & (d["jj"] < T))[:, None, :] &
              (np.linalg.norm(d["trajs_2d"], axis=-1) > 5) & (d["trajs_disp"] > 1e-2))
    e = (ms - aligned).abs()
    h = torch.where(e < 1.0, 0.5 * e * e, e - 0.5) * mask
    return h[torch.as_tensor(d["grid_query_frames"])].mean()


def _iproj(xy, disp, K):
    depth = 1.0 / disp.clamp(min=1e-2)
    return torch.stack([(xy[..., 0] - K[..., 2]) / K[..., 0] * depth, (xy[..., 1] - K[..., 3]) / K[..., 1] * depth, depth], -1)


def inter_frame_loss(d, ms, K=None):
    T, N, S = d["trajs_disp"].shape
    mid = S // 2
    okm = d["trajs_disp_mono"] > 1e-2
    acc = torch.zeros(S, N, N, dtype=torch.float64)
    xy, K = _t(d["trajs_2d"]), (_t(d["intrinsics"]) if K is None else K)
    for i in d["grid_query_frames"]:
        jj = d["jj"][i]
        Ki = K[torch.as_tensor(np.clip(jj, 0, T - 1))]
        P = _iproj(xy[i].permute(1, 0, 2), ms[i].T, Ki[:, None, :])                     # [S, N, 3]
        diff = P[:, :, None, :] - P[:, None, :, :]
        sq = (diff * diff).sum(-1)
        pd = torch.where(sq > 0, sq.clamp(min=1e-300).sqrt(), torch.zeros(()))          # norm with the zero subgradient at 0 (as torch.norm)
        vis, sta, okd = d["trajs_vis"][i].T, d["trajs_static"][i].T, okm[i].T
        mask = _t(((jj >= 0) & (jj < T))[:, None, None] & (vis[:, :, None] * vis[:, None, :] > 0.5) &
                  (sta[:, :, None] * sta[:, None, :] > 0.5) & (okd[:, :, None] & okd[:, None, :]))
        acc = acc + mask * (pd - pd[mid]).abs()
    return (acc / len(d["grid_query_frames"])).mean()


def total_and_grads(d, alpha, trajs_scales=None, frame_scales_=None):
    """(total, spatial, rigid, d total / d trajs_scales, d total / d frame_scales_) as float64 numpy."""
    ts = _t(d["trajs_scales"] if trajs_scales is None else trajs_scales).requires_grad_(True)
    fs = _t(d["frame_scales_"] if frame_scales_ is None else frame_scales_).requires_grad_(True)
    ms = frame_scaled_depth(d, fs)
    sp = spatial_loss(d, ts, ms)
    rg = inter_frame_loss(d, ms) if alpha > 0 else torch.zeros((), dtype=torch.float64)
    tot = sp + alpha * rg
    tot.backward()
    return float(tot.detach()), float(sp.detach()), float(rg.detach()), ts.grad.numpy(), fs.grad.numpy()


# ------------------------------------------------------------------ the rest of RefineNet.forward's total (refine_net.py:274-392)
def _qrot(q, p):
    qv, w = q[..., :3], q[..., 3:]
    uv = 2.0 * torch.linalg.cross(qv, p)
    return p + w * uv + torch.linalg.cross(qv, uv)


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)


def pts_3d_loss(d, ms, pose, K):
    """refine_net.py:314-354 with pose [T,7] (tx ty tz qx qy qz qw) and per-frame intrinsics K [T,4]."""
    T, N, S = d["trajs_disp"].shape
    mid = S // 2
    xy = _t(d["trajs_2d"])
    jc = torch.as_tensor(np.clip(d["jj"], 0, T - 1))                                  # [T,S]
    src = _iproj(xy[:, :, mid], ms[:, :, mid], K[:, None, :])                          # [T,N,3]
    pj, pt = pose[jc], pose[:, None, :].expand(T, S, 7)                                # pose[jj]^-1 * pose[t]
    qji = torch.cat([-pj[..., 3:6], pj[..., 6:]], -1)
    rel = _qrot(pt[..., 3:][:, None], src[:, :, None, :]) + (pt[..., :3] - pj[..., :3])[:, None]        # [T,N,S,3]
    from_src = _qrot(qji[:, None], rel)
    trg = _iproj(xy, ms, K[jc][:, None])                                               # [T,N,S,3]
    diff = from_src - trg
    sq = (diff * diff).sum(-1)
    dist = torch.where(sq > 0, sq.clamp(min=1e-300).sqrt(), torch.zeros(()))
    mask = _t((d["trajs_vis"] > 0.9) & ((d["jj"] >= 0) & (d["jj"] < T))[:, None, :] & (d["trajs_disp"] > 1e-2) & (d["trajs_static"] > 0.3))
    return (dist * mask).mean()


def cam_smooth_vec_loss(pose):
    def nrm(v):
        sq = (v * v).sum(-1)
        return torch.where(sq > 0, sq.clamp(min=1e-300).sqrt(), torch.zeros(()))
    return nrm(pose[:-1, :3] - pose[1:, :3]).mean() + 0.3 * nrm(pose[:-1, 3:] - pose[1:, 3:]).mean()


def scale_grid_smoothness_loss(frame_scales_, mode="l2"):
    s = torch.exp(frame_scales_ / 10.0)
    dh, dv = s[:, :, :-1] - s[:, :, 1:], s[:, :-1, :] - s[:, 1:, :]
    f = {"l2": lambda x: x * x, "l1": lambda x: x.abs(),
         "huber": lambda x: torch.where(x.abs() < 1.0, 0.5 * x * x, x.abs() - 0.5)}[mode]
    return f(dh).mean() + f(dv).mean()


def full_total_and_grads(d, weights, smooth_mode="l1", refine_intrinsics=False, K=None, K_scale=20.0,
                         trajs_scales=None, frame_scales_=None, pose=None):
    """The weighted total of RefineNet.forward (weights = spatial, rigid, pts3d, cam_smooth, scale_smooth) and its gradients
    w.r.t. trajs_scales, frame_scales_, pose and K, float64.  The pose gradient is stated in pypose's convention WITHOUT
    using its formulas: the poses that enter pts_3d_loss are Exp(delta_t) * pose_t written to first order in delta = (tau,
    phi) — t + tau + phi x t, (phi / 2, 1) * q — and autograd differentiates with respect to delta at 0 (the left perturbation,
    first six numbers); the term that reads the stored numbers directly (cam_smooth_vec_loss) is differentiated with
    respect to them.  Returns a dict."""
    w = [float(x) for x in weights]
    ts = _t(d["trajs_scales"] if trajs_scales is None else trajs_scales).requires_grad_(True)
    fs = _t(d["frame_scales_"] if frame_scales_ is None else frame_scales_).requires_grad_(True)
    raw = _t(d["pose"] if pose is None else pose).requires_grad_(True)
    T = raw.shape[0]
    delta = torch.zeros(T, 6, dtype=torch.float64, requires_grad=True)
    tau, phi = delta[:, :3], delta[:, 3:]
    t0, q0 = raw[:, :3].detach(), raw[:, 3:].detach()
    pert = torch.cat([t0 + tau + torch.linalg.cross(phi, t0), _qmul(torch.cat([0.5 * phi, torch.ones(T, 1, dtype=torch.float64)], -1), q0)], -1)
    if refine_intrinsics:
        Kp = _t(np.sort(d["intrinsics"], axis=0)[(T - 1) // 2] / K_scale if K is None else K).requires_grad_(True)   # torch.median: the lower middle value (refine_net.py:77)
        Kt = (Kp * K_scale).expand(T, 4)
    else:
        Kp, Kt = None, _t(d["intrinsics"])
    ms = frame_scaled_depth(d, fs)
    terms = dict(spatial=spatial_loss(d, ts, ms))
    terms["rigid"] = inter_frame_loss(d, ms, Kt) if w[1] else torch.zeros((), dtype=torch.float64)
    terms["pts3d"] = pts_3d_loss(d, ms, pert, Kt) if w[2] else torch.zeros((), dtype=torch.float64)
    terms["cam_smooth"] = cam_smooth_vec_loss(raw) if w[3] else torch.zeros((), dtype=torch.float64)
    terms["scale_smooth"] = scale_grid_smoothness_loss(fs, smooth_mode) if w[4] else torch.zeros((), dtype=torch.float64)
    tot = sum(wi * terms[k] for wi, k in zip(w, ("spatial", "rigid", "pts3d", "cam_smooth", "scale_smooth")))
    tot.backward()
    z = lambda g, like: np.zeros(like.shape) if g is None else g.numpy()
    g_pose = np.concatenate([z(delta.grad, delta), np.zeros((T, 1))], 1) + z(raw.grad, raw)
    return dict(total=float(tot.detach()), **{k: float(v.detach()) for k, v in terms.items()},
                grad_trajs_scales=z(ts.grad, ts), grad_frame_scales=z(fs.grad, fs), grad_pose=g_pose,
                grad_K=np.zeros(4) if Kp is None else z(Kp.grad, Kp))
