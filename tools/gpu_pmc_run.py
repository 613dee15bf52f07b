#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc passes: a few BA steps on a workload (M tracks per frame).
BT_PMC_COLD=1: a 512 MB sweep between the steps (L2 and MALL hold nothing of the graph when the Jacobian kernel starts)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
g = graphgen.make_graph(64, M, 8, seed=0)
f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights_pose)
ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
plan = Plan(ii, jj, kk, poses.shape[0], patches.shape[0], 1)
st = Stepper(plan, dev)
Po, Xo = torch.empty_like(poses), torch.empty_like(patches)
cold = os.environ.get("BT_PMC_COLD", "0") != "0"
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if cold else None
for _ in range(reps):
    if cold:
        flush.add_(1)
    st.step(poses, patches, mono, intr, t3, 3, w, Po, Xo, list(g.bounds), 1e-4, 10.0, 0.05, "huber", False)
torch.cuda.synchronize()
print(f"E={plan.E} m={plan.m} n_all={plan.n_all} algorithmic_bytes_k_tile={40*plan.E + 20*plan.m + 72*plan.n_all}")
