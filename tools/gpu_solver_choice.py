#!/usr/bin/env python3
"""Systems whose block-sparse factor does not fit LDS as double: the dense solver in double (the plan's choice since round 6, `wide`)
against the block-sparse float32 factor with refinement (BT_FORCE=solver=lds keeps the old choice).  Banded graphs (the benchmark
generator: every track seen from 8 consecutive frames) of N frames, and — argument `dense` — the generator's graph with a third of
the targets anywhere on the trajectory (a nearly dense reduced system).  Run once per setting:
    python tools/gpu_solver_choice.py ; BT_FORCE=solver=lds python tools/gpu_solver_choice.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oracle
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper

dev = "cuda:0"
print("BT_FORCE =", os.environ.get("BT_FORCE", ""))
CASES = ((64, 0.0), (96, 0.0), (128, 0.0), (192, 0.0), (256, 0.0), (48, 0.3), (96, 0.3), (160, 0.3), (256, 0.3))
if len(sys.argv) > 1 and sys.argv[1] == "dense":                  # (for a kernel trace of the dense solver alone)
    CASES = ((96, 0.3), (256, 0.3))
for N, far in CASES:
    g = graphgen.make_graph(N, 64, 8, seed=N)
    ii, jj, kk = g.ii.copy(), g.jj.copy(), g.kk.copy()
    t3 = np.asarray(g.targets3, np.float64).copy()
    if far > 0:                                                  # long-range edges: targets re-drawn on any frame, reprojected from the ground truth
        rng = np.random.default_rng(N)
        sel = rng.random(ii.size) < far
        jj[sel] = rng.integers(0, N, int(sel.sum()))
        gt = g.patches.copy(); gt[:, 2] = g.disp_gt
        u, v, _ = graphgen.reproject(g.poses_gt, gt, g.intrinsics, ii, jj, kk)
        t3[:, 0], t3[:, 1] = u, v
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    poses, patches, mono, intr, tg, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(t3), f32(g.weights_pose)
    I, J, K = (torch.as_tensor(a, device=dev) for a in (ii, jj, kk))
    plan = Plan(I, J, K, poses.shape[0], patches.shape[0], 1)
    st = Stepper(plan, dev)
    Po, Xo = torch.empty_like(poses), torch.empty_like(patches)
    args = (poses, patches, mono, intr, tg, 3, w, Po, Xo, list(g.bounds), 1e-4, 10.0, 0.05, "huber", False)
    for _ in range(3): st.step(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): st.step(*args)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
    f64 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    ref = oracle.ba_step(f64(g.poses), f64(g.patches), f64(g.mono_disp), f64(g.intrinsics), f64(t3), f64(g.weights_pose), ii, jj, kk, g.bounds,
                         fixedp=1, want_system=True, lmbda=float(np.float32(1e-4)), alpha=float(np.float32(0.05)))
    dx = st.dx.cpu().numpy().astype(np.float64).reshape(-1)
    xr = ref["dX"].reshape(-1)
    n = plan.n
    print(f"N={N:4d} far={far:.1f} n={n:4d} factor blocks {plan.info['nnz_blocks']:6d} (dense {n * (n + 1) // 2:6d})  step {us:9.1f} us  "
          f"dX vs oracle {np.linalg.norm(dx - xr) / np.linalg.norm(xr):.2e}  status {st.status()}", flush=True)
