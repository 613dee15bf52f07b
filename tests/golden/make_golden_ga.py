#!/usr/bin/env python3
"""Golden vectors for the dense global-alignment losses: runs the reference's UNMODIFIED
/root/reference/main/global_refine/model/refine_net.py (RefineNet.get_frame_scaled_depth, the spatial huber term of
forward(), inter_frame_loss, pts_3d_loss, and the gradients of forward() by the reference's autograd) on synthetic tracks,
in this container only.  `pypose` is absent: the
stand-in of tests/golden/refstubs/pypose is used (SE3 compose / inverse / action — our restatement; what these vectors
pin is refine_net.py).  A RefineNet object is built without its __init__ (which reads a results.pkl): the attributes
forward() reads are set directly.  Only inputs we generated and numeric outputs are written (tests/golden/ga_small.npz).

    python tests/golden/make_golden_ga.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/main/global_refine"
sys.path[:0] = [os.path.join(HERE, "refstubs"), REF]

import pypose as pp                       # noqa: E402  (stand-in)
from model.refine_net import RefineNet    # noqa: E402  (reference, unmodified)


def make_inputs(T=10, N=24, S=5, gh=4, gw=4, H=96, W=128, seed=0):
    rng = np.random.default_rng(seed)
    mid = S // 2
    ii = np.arange(T)
    jj = ii[:, None] + np.arange(S)[None] - mid                                     # refine_net.py:92-97 (unclamped: masks use it)
    trajs_2d = np.stack([rng.uniform(0, W - 1, (T, N, S)), rng.uniform(0, H - 1, (T, N, S))], -1)
    trajs_2d[0, 0, 0] = [1.0, 2.0]                                                  # |xy| < 5: the flow mask of forward()
    disp = rng.uniform(0.05, 1.5, (T, N, S))
    disp[1, 2, 3] = 0.005                                                           # below the 1e-2 mask
    mono = disp * rng.uniform(0.7, 1.4, (T, 1, 1)) * (1 + 0.05 * rng.standard_normal((T, N, S)))
    mono[2, 3, 1] = 0.004
    vis = rng.uniform(0.3, 1.0, (T, N, S))
    static = rng.uniform(0.0, 1.0, (T, N, S))
    K = np.tile(np.array([110.0, 105.0, W / 2, H / 2]), (T, 1)) * (1 + 0.01 * rng.standard_normal((T, 4)))
    q = rng.standard_normal((T, 4)) * 0.05 + np.array([0, 0, 0, 1.0])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    pose = np.concatenate([rng.standard_normal((T, 3)) * 0.2, q], 1)
    return dict(trajs_2d=trajs_2d, trajs_disp=disp, trajs_disp_mono=mono, trajs_vis=vis, trajs_static=static,
                jj=jj.astype(np.int64), intrinsics=K, pose=pose, grid_query_frames=np.array([0, 2, 3, 7, 9], np.int64),
                trajs_scales=rng.standard_normal((T, N, S)) * 0.3, frame_scales_=rng.standard_normal((T, gh, gw)) * 2.0,
                frame_shifts=np.zeros(T), H=np.int64(H), W=np.int64(W), pw_break=np.float64(20.0))


def build(d, dtype):
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    net = object.__new__(RefineNet)
    torch.nn.Module.__init__(net)
    T, N, S = d["trajs_disp"].shape
    net.T, net.N, net.S_local, net.H, net.W = T, N, S, int(d["H"]), int(d["W"])
    net.trajs_2d, net.trajs_disp, net.trajs_disp_mono = t(d["trajs_2d"]), t(d["trajs_disp"]), t(d["trajs_disp_mono"])
    net.trajs_vis, net.trajs_static = t(d["trajs_vis"]), t(d["trajs_static"])
    net.jj = torch.as_tensor(d["jj"])
    net.ii = torch.arange(T)[:, None].repeat(1, S)
    net.intrinsics_raw = t(d["intrinsics"])
    net.refine_intrinsics = False
    net.grid_query_frames = torch.as_tensor(d["grid_query_frames"])
    net.trajs_scales = torch.nn.Parameter(t(d["trajs_scales"]))
    net.frame_scales_ = torch.nn.Parameter(t(d["frame_scales_"]))
    net.frame_shifts_ = t(d["frame_shifts"])
    net.scale_mode, net.norm_pw_scale, net.pw_break = "exp", True, float(d["pw_break"])
    net.pose = pp.SE3(t(d["pose"]))
    # the masks __init__ precomputes (refine_net.py:113-121)
    from einops import rearrange
    for name, src in (("trajs_static_mat", net.trajs_static), ("trajs_vis_mat", net.trajs_vis),
                      ("trajs_disp_mono_mask_mat", (net.trajs_disp_mono > 1e-2).to(dtype))):
        m = rearrange(src, "t n s -> t s n")
        setattr(net, name, m.unsqueeze(3) @ m.unsqueeze(2))
    net.loss_weight_dict, net.verbose, net.scale_smoothness_weight, net.scale_smoothness_mode = None, False, 0.0, "l2"
    return net


def main():
    d = make_inputs()
    out = {k: np.asarray(v) for k, v in d.items()}
    with torch.no_grad():
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            net = build(d, dtype)
            out[f"{tag}.mono_scaled"] = net.get_frame_scaled_depth().numpy()
            out[f"{tag}.trajs_scales_exp"] = net.get_trajs_scales().numpy()
            net.alpha = 0.0
            out[f"{tag}.loss_spatial"] = np.float64(net.forward().item())           # forward() with alpha = 0: the spatial huber term alone
            out[f"{tag}.loss_rigid"] = np.float64(net.inter_frame_loss().item())
            out[f"{tag}.loss_pts3d"] = np.float64(net.pts_3d_loss().item())
            net.alpha = 0.5
            out[f"{tag}.total_alpha05"] = np.float64(net.forward().item())
    # gradients of forward() (spatial + alpha * rigid) w.r.t. the two parameters the default loss reaches, by the reference's
    # own autograd
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        for alpha in (0.0, 0.5):
            net = build(d, dtype)
            net.alpha = alpha
            net.forward().backward()
            key = "a00" if alpha == 0.0 else "a05"
            out[f"{tag}.grad_trajs_scales_{key}"] = net.trajs_scales.grad.numpy()
            out[f"{tag}.grad_frame_scales_{key}"] = net.frame_scales_.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "ga_small.npz"), **out)
    print({k: float(v) for k, v in out.items() if np.ndim(v) == 0 and k[0] == "f"})


if __name__ == "__main__":
    main()
