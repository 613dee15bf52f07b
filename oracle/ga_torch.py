"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

torch (float64, CPU) statement of the forward losses of oracle/ga_losses.py, so that torch.autograd gives the gradients of
    total(alpha) = spatial_loss + alpha * inter_frame_loss         (RefineNet.forward, refine_net.py:252-293)
with respect to the two parameters it reaches, `trajs_scales` [T,N,S] and `frame_scales_` [T,gh,gw] — the checker of
bt_ga_backward (include/batrack_ga.h) at sizes other than the fixture's.  Pinned twice: the forward equals the numpy
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


def inter_frame_loss(d, ms):
    T, N, S = d["trajs_disp"].shape
    mid = S // 2
    okm = d["trajs_disp_mono"] > 1e-2
    acc = torch.zeros(S, N, N, dtype=torch.float64)
    xy, K = _t(d["trajs_2d"]), _t(d["intrinsics"])
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
