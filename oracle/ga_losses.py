"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU restatement (numpy) of the forward losses of the reference's dense global-alignment stage
(/root/reference/main/global_refine/model/refine_net.py):
    trajs_scales()        refine_net.py:123-127   exp((s - mean_n s) / pw_break)
    frame_scaled_depth()  refine_net.py:148-174   per-frame scale grid exp(g / 10), bilinear (align_corners) at the track
                                                  position, times the mono disparity, plus the frame shift
    spatial_loss()        refine_net.py:252-268   masked smooth-L1 between the scaled mono disparity and the scaled track
                                                  disparity, mean over the query frames
    inter_frame_loss()    refine_net.py:199-225   O(Q S N^2) rigidity: |pairwise 3-D distance at slot s - at the centre slot|
    pts_3d_loss()         refine_net.py:300-345   3-D point consistency through the relative camera poses
Pinned by tests/golden/ga_small.npz (tests/golden/make_golden_ga.py runs the unmodified reference with a stand-in for the
absent pypose package: SE3 inverse / composition / action are our restatement of the published formulas).
"""
import numpy as np


def trajs_scales(p, pw_break=20.0):
    s = p - p.mean(axis=1, keepdims=True)
    return np.exp(s / pw_break)


def _qrot(q, v):
    qv, w = q[..., :3], q[..., 3:]
    uv = 2.0 * np.cross(qv, v)
    return v + w * uv + np.cross(qv, uv)


def _se3_rel(pj, pt):
    """pose_j^-1 * pose_t as (t, q), rows tx ty tz qx qy qz qw."""
    tj, qj, tt, qt = pj[..., :3], pj[..., 3:], pt[..., :3], pt[..., 3:]
    qji = np.concatenate([-qj[..., :3], qj[..., 3:]], -1)
    ax, ay, az, aw = np.moveaxis(qji, -1, 0)
    bx, by, bz, bw = np.moveaxis(np.broadcast_to(qt, qji.shape), -1, 0)
    q = np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                  aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)
    return _qrot(qji, tt - tj), q


def frame_scaled_depth(d):
    T, N, S = d["trajs_disp_mono"].shape
    g = np.exp(d["frame_scales_"] / 10.0)                                            # :139-140
    gh, gw = g.shape[1:]
    jc = np.clip(d["jj"], 0, T - 1)                                                  # [T, S]
    x = d["trajs_2d"][..., 0] / (int(d["W"]) - 1) * (gw - 1)                         # align_corners=True: [-1, 1] -> [0, size - 1]
    y = d["trajs_2d"][..., 1] / (int(d["H"]) - 1) * (gh - 1)
    x0, y0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    fx, fy = x - x0, y - y0
    fr = np.broadcast_to(jc[:, None, :], (T, N, S))
    out = np.zeros((T, N, S), g.dtype)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            yy, xx = y0 + dy, x0 + dx
            ok = (yy >= 0) & (yy < gh) & (xx >= 0) & (xx < gw)                       # padding_mode='zeros'
            out += np.where(ok, g[fr, np.clip(yy, 0, gh - 1), np.clip(xx, 0, gw - 1)], 0.0) * wy * wx
    return d["trajs_disp_mono"] * out + d["frame_shifts"][jc][:, None, :]


def spatial_loss(d, mono_scaled=None):
    T, N, S = d["trajs_disp"].shape
    ms = frame_scaled_depth(d) if mono_scaled is None else mono_scaled
    aligned = trajs_scales(d["trajs_scales"], float(d["pw_break"])) * d["trajs_disp"]
    mask = ((d["trajs_vis"] > 0.9) & ((d["jj"] >= 0) & (d["jj"] < T))[:, None, :] &
            (np.linalg.norm(d["trajs_2d"], axis=-1) > 5) & (d["trajs_disp"] > 1e-2)).astype(ms.dtype)
    e = np.abs(ms - aligned)
    h = np.where(e < 1.0, 0.5 * e * e, e - 0.5) * mask                               # F.smooth_l1_loss, beta = 1
    return h[d["grid_query_frames"]].mean()


def _iproj(xy, disp, K):
    depth = 1.0 / np.clip(disp, 1e-2, None)                                          # geomeotry.py:3-18
    return np.stack([(xy[..., 0] - K[..., 2]) / K[..., 0] * depth, (xy[..., 1] - K[..., 3]) / K[..., 1] * depth, depth], -1)


def inter_frame_loss(d, mono_scaled=None):
    T, N, S = d["trajs_disp"].shape
    ms = frame_scaled_depth(d) if mono_scaled is None else mono_scaled
    mid = S // 2
    okm = (d["trajs_disp_mono"] > 1e-2).astype(ms.dtype)
    acc = np.zeros((S, N, N), ms.dtype)
    for i in d["grid_query_frames"]:
        jj = d["jj"][i]
        K = d["intrinsics"][np.clip(jj, 0, T - 1)]                                   # [S, 4]
        P = _iproj(np.moveaxis(d["trajs_2d"][i], 1, 0), ms[i].T, K[:, None, :])      # [S, N, 3]
        pd = np.linalg.norm(P[:, :, None, :] - P[:, None, :, :], axis=-1)
        diff = np.abs(pd - pd[mid])
        vis, sta, okd = d["trajs_vis"][i].T, d["trajs_static"][i].T, okm[i].T        # [S, N]
        mask = (((jj >= 0) & (jj < T))[:, None, None] & (vis[:, :, None] * vis[:, None, :] > 0.5) &
                (sta[:, :, None] * sta[:, None, :] > 0.5) & (okd[:, :, None] * okd[:, None, :] > 0.5))
        acc += mask * diff
    return (acc / len(d["grid_query_frames"])).mean()


def pts_3d_loss(d, mono_scaled=None):
    T, N, S = d["trajs_disp"].shape
    ms = frame_scaled_depth(d) if mono_scaled is None else mono_scaled
    mid = S // 2
    jc = np.clip(d["jj"], 0, T - 1)
    src = _iproj(d["trajs_2d"][:, :, mid], ms[:, :, mid], d["intrinsics"][:, None, :])            # [T, N, 3]
    t_rel, q_rel = _se3_rel(d["pose"][jc], d["pose"][:, None, :])                                   # [T, S, .]
    from_src = _qrot(q_rel[:, None], src[:, :, None, :]) + t_rel[:, None]                           # [T, N, S, 3]
    trg = _iproj(d["trajs_2d"], ms, d["intrinsics"][jc][:, None, :, :])
    dist = np.linalg.norm(from_src - trg, axis=-1)
    mask = ((d["trajs_vis"] > 0.9) & ((d["jj"] >= 0) & (d["jj"] < T))[:, None, :] & (d["trajs_disp"] > 1e-2) &
            (d["trajs_static"] > 0.3)).astype(ms.dtype)
    return (dist * mask).mean()


def forward(d, alpha=0.5):
    """RefineNet.forward with loss_weight_dict = None and scale_smoothness_weight = 0 (refine_net.py:291-293)."""
    ms = frame_scaled_depth(d)
    return spatial_loss(d, ms) + (alpha * inter_frame_loss(d, ms) if alpha > 0 else 0.0)
