#!/usr/bin/env python3
"""Golden vectors for the TOTAL the reference's global-alignment stage optimises: the reference's
UNMODIFIED /root/reference/main/global_refine/model/refine_net.py — RefineNet.forward in both branches, with
  (A) the `loss_weight_dict` run_global_refine.py:61-67 always passes, intrinsics refined (`fixed_K` defaults to False,
      run_global_refine.py:56-57 -> refine_intrinsics=True) and the poses free, and
  (B) `loss_weight_dict=None` at the constructor's real defaults (alpha 0.5, scale_smoothness_weight 0.1),
every term of the total, and the gradient of the total with respect to every parameter trainer.py:33-43 hands to Adam
(trajs_scales, frame_scales_, pose, K) by the reference's own autograd.  `pypose` is absent: tests/golden/refstubs/pypose
stands in (SE3 compose / inverse / action and pypose's left-perturbation gradient convention — our restatement, unpinned).
Only inputs we generated and numeric outputs are written (tests/golden/ga_total.npz).

    python tests/golden/make_golden_ga_total.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ga as base              # noqa: E402  (inputs, the __init__-free construction; puts the stubs and the reference on sys.path)
import pypose as pp                        # noqa: E402  (stand-in)

WEIGHTS = {"spatial_loss": 5.0, "inter_frame_loss": 0.3, "pts_3d_loss": 1.0, "cam_smooth_vec_loss": 1.0,
           "scale_smoothness_loss": 0.3}                       # run_global_refine.py:61-67


def build(d, dtype, weights, refine_intrinsics):
    net = base.build(d, dtype)
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    net.K_scale = 20
    net.K_init = torch.median(net.intrinsics_raw, dim=0)[0] / net.K_scale            # refine_net.py:77
    net.K = torch.nn.Parameter(net.K_init.clone())
    net.refine_intrinsics = refine_intrinsics
    leaf = t(d["pose"]).requires_grad_(True)                                          # pp.Parameter(pose_init), refine_net.py:45
    net.pose = pp.SE3(leaf)
    net.loss_weight_dict = weights
    net.alpha, net.scale_smoothness_weight, net.scale_smoothness_mode = 0.5, 0.1, "l2"   # the constructor's defaults, refine_net.py:16
    return net, leaf


def main():
    d = base.make_inputs(T=10, N=24, S=5, seed=1)
    out = {k: np.asarray(v) for k, v in d.items()}
    out["weights"] = np.array([WEIGHTS[k] for k in ("spatial_loss", "inter_frame_loss", "pts_3d_loss", "cam_smooth_vec_loss", "scale_smoothness_loss")])
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        for name, weights, refine_k in (("A", WEIGHTS, True), ("B", None, False)):
            net, leaf = build(d, dtype, weights, refine_k)
            with torch.no_grad():
                out[f"{tag}.{name}.K_init"] = net.K_init.numpy()
                out[f"{tag}.{name}.cam_smooth"] = np.float64(net.cam_smooth_vec_loss().item())
                for mode in ("l1", "l2", "huber"):
                    out[f"{tag}.{name}.scale_smooth_{mode}"] = np.float64(net.scale_grid_smoothness_loss(mode=mode).item())
                out[f"{tag}.{name}.pts3d"] = np.float64(net.pts_3d_loss().item())
                out[f"{tag}.{name}.rigid"] = np.float64(net.inter_frame_loss().item())
                net.alpha = 0.0
                sm = net.scale_smoothness_weight
                net.scale_smoothness_weight, keep = 0.0, net.loss_weight_dict
                net.loss_weight_dict = None
                out[f"{tag}.{name}.spatial"] = np.float64(net.forward().item())
                net.alpha, net.scale_smoothness_weight, net.loss_weight_dict = 0.5, sm, keep
            total = net.forward()
            total.backward()
            out[f"{tag}.{name}.total"] = np.float64(total.item())
            out[f"{tag}.{name}.grad_trajs_scales"] = net.trajs_scales.grad.numpy()
            out[f"{tag}.{name}.grad_frame_scales"] = net.frame_scales_.grad.numpy()
            out[f"{tag}.{name}.grad_pose"] = leaf.grad.numpy() if leaf.grad is not None else np.zeros((net.T, 7))
            out[f"{tag}.{name}.grad_K"] = net.K.grad.numpy() if net.K.grad is not None else np.zeros(4)
    np.savez_compressed(os.path.join(HERE, "ga_total.npz"), **out)
    print({k: float(v) for k, v in out.items() if np.ndim(v) == 0 and k[0] == "f"})
    print({k: float(np.abs(v).max()) for k, v in out.items() if ".grad_" in k and k.startswith("f64")})


if __name__ == "__main__":
    main()
