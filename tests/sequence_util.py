"""Test-side helpers for the caller-loop replay (batrack_amd/sequence.py): the CPU oracle wrapped
in the reference's `BA_rgbd_droid` signature, so the same `WindowedBA` loop can be driven by the
oracle (here) and by the HIP step (GPU tests).  Test infrastructure only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import oracle  # noqa: E402


def oracle_BA_rgbd_droid(poses, patches, patches_monodisp, intrinsics, targets_2d, targets_disp, weights, lmbda,
                         ii, jj, kk, bounds, ep=100.0, PRINT=False, fixedp=1, structure_only=False,
                         loss='trivial', alpha=0.5, dtype=np.float64):
    """ba.py:217 signature over oracle.ba_step (float64 arithmetic, float32 state like the caller's buffers)."""
    n_buf, p_tot = poses.data.shape[1], patches.shape[1]
    out = oracle.ba_step(poses.data[0].cpu().numpy(), patches[0, :, :, 0, 0].cpu().numpy(),
                         patches_monodisp.reshape(-1).cpu().numpy(), intrinsics[0].cpu().numpy(),
                         targets_2d[0].cpu().numpy(), weights[0].cpu().numpy(),
                         ii.cpu().numpy(), jj.cpu().numpy(), kk.cpu().numpy(), bounds,
                         lmbda=float(lmbda), ep=float(ep), alpha=float(alpha), fixedp=int(fixedp),
                         structure_only=bool(structure_only), loss=loss, dtype=dtype)
    dev = patches.device
    pat = torch.as_tensor(out["patches_out"], dtype=torch.float32, device=dev).view(1, p_tot, 3, 1, 1)
    if structure_only:
        return poses, pat
    return type(poses)(torch.as_tensor(out["poses_out"], dtype=torch.float32, device=dev).view(1, n_buf, 7)), pat
