#!/usr/bin/env python3
"""Golden vectors for the hand-off from the sparse-SLAM stage to the dense global alignment: the
reference's UNMODIFIED /root/reference/main/global_refine/model/refine_net.py — `RefineNet.__init__` -> `_init_from_ba`
(refine_net.py:16-121) — run on a synthetic results.pkl, in this container only.  What it derives from the dictionary is
written next to the dictionary itself: `pose_init`, `K_init`, `jj`, `trajs_disp_mono` (the depth maps sampled at the tracks),
`trajs_2d`, `trajs_disp`, and — with the module's parameters at their initial values (ones / zeros) plus a seeded perturbation —
the total of forward() with run_global_refine.py:61-67's weights and its gradients, so that a test can go
dictionary -> from_results -> forward()/backward() and land on the reference's numbers.  `align_depth=True` once as well.
`pypose` is absent: tests/golden/refstubs/pypose stands in (mat2SE3 and the SE3 operations — our restatement, unpinned).
Only inputs we generated and numeric outputs are written (tests/golden/ga_init.npz).

    python tests/golden/make_golden_ga_init.py
"""
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/main/global_refine"
sys.path[:0] = [os.path.join(HERE, "refstubs"), REF]

from model.refine_net import RefineNet    # noqa: E402  (reference, unmodified)

WEIGHTS = {"spatial_loss": 5.0, "inter_frame_loss": 0.3, "pts_3d_loss": 1.0, "cam_smooth_vec_loss": 1.0,
           "scale_smoothness_loss": 0.3}                       # run_global_refine.py:61-67


def make_results(T=9, N=14, S=5, H=40, W=56, seed=2):
    """A results.pkl as BATRACK.get_results writes it (batrack.py:1113-1125), small."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((T, 4)) * 0.08 + np.array([0, 0, 0, 1.0])
    q[3] = [0.9, 0.1, -0.2, 0.05]                                       # a large rotation: trace < 0 branch of mat2SE3
    q[5] = [0.1, 0.95, 0.1, -0.1]
    q[6] = [0.05, -0.1, 0.97, 0.1]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                  np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                  np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
    cams = np.tile(np.eye(4), (T, 1, 1))
    cams[:, :3, :3] = R
    cams[:, :3, 3] = rng.standard_normal((T, 3)) * 0.3
    t2d = np.stack([rng.uniform(-1.5, W + 0.5, (T, N, S)), rng.uniform(-1.5, H + 0.5, (T, N, S)), rng.uniform(0.05, 1.5, (T, N, S))], -1)
    t2d[0, 0, 0, :2] = [W - 1.0, H - 1.0]                                # the last pixel exactly
    t2d[1, 1, 1, :2] = [0.0, 0.0]
    t2d[2, 2, S // 2, :2] = [6.5, 6.25]                                  # inside the patch of frame 2 whose depth is below the clamp
    t2d[3, 4, S // 2 - 1, :2] = [7.0, 6.0]                               # the same patch seen from frame 3's slot for frame 2
    yy, xx = np.mgrid[0:H, 0:W]
    dm = np.stack([1.5 + np.sin(0.2 * xx + 0.3 * f) * np.cos(0.15 * yy) + 0.3 * rng.random((H, W)) for f in range(T)])[..., None]
    dm[2, 5:9, 5:9, 0] = 0.004                                          # below the 1e-2 clamp of the depth
    vis = rng.uniform(0.3, 1.0, (T, N, S))
    static = (rng.random((T, N, S)) > 0.2).astype(np.float64)
    valid = rng.random((T, N)) > 0.2
    K = np.tile(np.array([60.0, 58.0, W / 2, H / 2]), (T, 1)) * (1 + 0.02 * rng.standard_normal((T, 4)))
    return {"cams_T_world": cams.astype(np.float32), "intrinsics": K.astype(np.float32), "tstamps": np.arange(T, dtype=float),
            "trajs_2d_disp": t2d.astype(np.float32), "trajs_valid": valid, "trajs_static": static.astype(np.float32),
            "trajs_vis": vis.astype(np.float32), "grid_query_frames": np.array([0, 1, 2, 4, 5, 7, 8]),
            "dmaps": dm.astype(float), "rgbs": None, "dmaps_gt": None}


def main():
    res = make_results()
    out = {"in." + k: v for k, v in res.items() if v is not None}
    rng = np.random.default_rng(5)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "results.pkl")
        with open(path, "wb") as f:
            pickle.dump(res, f)
        for tag, align in (("plain", False), ("aligned", True)):
            net = RefineNet("cpu", path, grid_size=4, pw_break=20, verbose=False, align_depth=align, loss_weight_dict=dict(WEIGHTS),
                            refine_intrinsics=True)
            for k in ("pose_init", "K_init", "jj", "ii", "trajs_disp_mono", "trajs_2d", "trajs_disp"):
                v = getattr(net, k)
                out[f"{tag}.{k}"] = (v.tensor() if hasattr(v, "tensor") else v).detach().numpy().copy()
            out[f"{tag}.T_N_S_H_W"] = np.array([net.T, net.N, net.S_local, net.H, net.W])
            if align:
                continue
            # the parameters right after __init__ ...
            out["init.trajs_scales"] = net.trajs_scales.detach().numpy().copy()
            out["init.frame_scales_"] = net.frame_scales_.detach().numpy().copy()
            out["init.total"] = np.float64(net.forward().item())
            # ... and perturbed (ones and zeros hide errors), with the gradients of the total by the reference's autograd
            pts, pfs = rng.standard_normal(tuple(net.trajs_scales.shape)) * 0.3, rng.standard_normal(tuple(net.frame_scales_.shape)) * 2.0
            with torch.no_grad():
                net.trajs_scales += torch.as_tensor(pts, dtype=torch.float32)
                net.frame_scales_ += torch.as_tensor(pfs, dtype=torch.float32)
            out["pert.trajs_scales"], out["pert.frame_scales_"] = net.trajs_scales.detach().numpy().copy(), net.frame_scales_.detach().numpy().copy()
            total = net.forward()
            total.backward()
            out["pert.total"] = np.float64(total.item())
            out["pert.grad_trajs_scales"] = net.trajs_scales.grad.numpy()
            out["pert.grad_frame_scales"] = net.frame_scales_.grad.numpy()
            out["pert.grad_pose"] = net.pose.grad.numpy() if net.pose.grad is not None else np.zeros((net.T, 7), np.float32)
            out["pert.grad_K"] = net.K.grad.numpy()
    out["weights"] = np.array([WEIGHTS[k] for k in ("spatial_loss", "inter_frame_loss", "pts_3d_loss", "cam_smooth_vec_loss", "scale_smoothness_loss")])
    np.savez_compressed(os.path.join(HERE, "ga_init.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items() if not k.startswith("in.")})
    print("totals", float(out["init.total"]), float(out["pert.total"]))


if __name__ == "__main__":
    main()
