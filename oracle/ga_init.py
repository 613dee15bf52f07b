"""CPU restatement of the hand-off from the sparse-SLAM stage to the dense global alignment (TEST INFRASTRUCTURE: only
tests/ import it; the product path is batrack_amd.global_refine.RefineLosses.from_results over include/batrack_ga.h).

Follows /root/reference/main/global_refine/model/refine_net.py:53-121 (`RefineNet._init_from_ba`) and what it calls:
model/utils.py:6-102 (`bilinear_sample2d`), :268-312 (`align_depth_maps`).  Pinned by tests/golden/ga_init.npz, which
tests/golden/make_golden_ga_init.py produces by running the reference's UNMODIFIED RefineNet.__init__ on a synthetic results
dictionary (the one un-pinned piece: `pp.mat2SE3` — pypose is not in the image, tests/golden/refstubs/pypose stands in —
so poses are compared as rotations, q and -q alike).
"""
import numpy as np


def mat_to_se3(m):
    """pp.mat2SE3 (refine_net.py:61), as restated in tests/golden/refstubs/pypose: (t, q_xyzw), q normalised."""
    m = np.asarray(m, np.float64)
    out = np.zeros(m.shape[:-2] + (7,))
    for idx in np.ndindex(*m.shape[:-2]):
        R = m[idx][:3, :3]
        tr = np.trace(R)
        if tr > 0:
            sq = np.sqrt(tr + 1.0) * 2.0
            q = [(R[2, 1] - R[1, 2]) / sq, (R[0, 2] - R[2, 0]) / sq, (R[1, 0] - R[0, 1]) / sq, 0.25 * sq]
        elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
            sq = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2.0
            q = [0.25 * sq, (R[0, 1] + R[1, 0]) / sq, (R[0, 2] + R[2, 0]) / sq, (R[2, 1] - R[1, 2]) / sq]
        elif R[1, 1] > R[2, 2]:
            sq = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2.0
            q = [(R[0, 1] + R[1, 0]) / sq, 0.25 * sq, (R[1, 2] + R[2, 1]) / sq, (R[0, 2] - R[2, 0]) / sq]
        else:
            sq = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2.0
            q = [(R[0, 2] + R[2, 0]) / sq, (R[1, 2] + R[2, 1]) / sq, 0.25 * sq, (R[1, 0] - R[0, 1]) / sq]
        q = np.asarray(q)
        out[idx] = np.concatenate([m[idx][:3, 3], q / np.linalg.norm(q)])
    return out


def bilinear_sample2d(im, x, y):
    """model/utils.py:6-91 for im [B,C,H,W], x, y [B,N] -> [B,C,N]: coordinates and weights in float32 (`x.float()`), corner
    indices clamped to the image, weights from the unclamped corners, the blend in the image's dtype."""
    B, C, H, W = im.shape
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    x0 = np.floor(x).astype(np.int32); x1 = x0 + 1
    y0 = np.floor(y).astype(np.int32); y1 = y0 + 1
    x0c, x1c = np.clip(x0, 0, W - 1), np.clip(x1, 0, W - 1)
    y0c, y1c = np.clip(y0, 0, H - 1), np.clip(y1, 0, H - 1)
    b = np.arange(B)[:, None]
    px = lambda yy, xx: np.transpose(im[b, :, yy, xx], (0, 2, 1))          # [B,C,N]
    x0f, x1f, y0f, y1f = (a.astype(np.float32) for a in (x0, x1, y0, y1))
    w00, w01 = ((x1f - x) * (y1f - y))[:, None], ((x - x0f) * (y1f - y))[:, None]
    w10, w11 = ((x1f - x) * (y - y0f))[:, None], ((x - x0f) * (y - y0f))[:, None]
    return w00 * px(y0c, x0c) + w01 * px(y0c, x1c) + w10 * px(y1c, x0c) + w11 * px(y1c, x1c)


def align_depth_maps(depth_maps):
    """model/utils.py:268-312: every map scaled so that its median over the overlap matches the median of the previous
    (two) aligned maps."""
    S = depth_maps.shape[0]
    out = np.zeros_like(depth_maps)
    out[0] = depth_maps[0]
    for i in range(1, S):
        prev, cur = out[i - 1, ..., 0], depth_maps[i, ..., 0]
        mask = (prev > 0) & (cur > 0)
        if mask.sum() < 100:
            out[i, ..., 0] = cur
            continue
        if i == 1:
            med_prev = np.median(prev[mask])
        else:
            past = out[i - 2, ..., 0]
            med_prev = np.median(np.concatenate((past[(past > 0) & (prev > 0)], prev[mask])))
        out[i, ..., 0] = med_prev / np.median(cur[mask]) * cur
    return out


def init_from_ba(results, K_scale=20, align_depth=False):
    """refine_net.py:53-121.  Returns the attributes `_init_from_ba` leaves on the module (same names)."""
    dm = align_depth_maps(results["dmaps"]) if align_depth else np.asarray(results["dmaps"])
    dmaps = np.transpose(dm, (0, 3, 1, 2))                                   # 't h w c -> t c h w'
    t2d = np.asarray(results["trajs_2d_disp"])
    T, N, S, _ = t2d.shape
    intr = np.asarray(results["intrinsics"])
    out = dict(pose_init=mat_to_se3(results["cams_T_world"]), trajs_2d=t2d[..., :2], trajs_disp=t2d[..., 2],
               trajs_valid=np.asarray(results["trajs_valid"]), trajs_static=np.asarray(results["trajs_static"]),
               trajs_vis=np.asarray(results["trajs_vis"]), grid_query_frames=np.asarray(results["grid_query_frames"]),
               intrinsics_raw=intr, K_init=np.sort(intr, axis=0)[(T - 1) // 2] / K_scale,     # torch.median: the LOWER middle value
               T=T, N=N, S_local=S, H=dmaps.shape[-2], W=dmaps.shape[-1])
    ii = np.arange(T)
    jj = ii[:, None] + np.arange(S)[None] - S // 2
    out["ii"], out["jj"] = np.repeat(ii[:, None], S, 1), jj
    mono = np.zeros((T, N, S), dmaps.dtype)
    for t in range(T):
        f = np.clip(jj[t], 0, T - 1)
        xy = np.transpose(out["trajs_2d"][t], (1, 0, 2))                      # 'n s c -> s n c'
        depth = bilinear_sample2d(dmaps[f], xy[..., 0], xy[..., 1])           # [S,1,N]
        mono[t] = (1.0 / np.maximum(depth, 1e-2))[:, 0].T
    out["trajs_disp_mono"] = mono
    return out
