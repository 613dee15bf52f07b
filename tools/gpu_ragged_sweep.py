#!/usr/bin/env python3
"""The benchmark generator's graph with a random 15 % of its observations dropped (tracks of different lengths: what a real large graph looks like),
4096 and 16384 tracks per frame: which Jacobian kernel the plan picks, and the kernels' times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper
dev = "cuda:0"
for M in (4096, 16384):
    g = graphgen.make_graph(64, M, 8, seed=0)
    keep = np.random.default_rng(11).random(np.asarray(g.kk).size) > 0.15
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    poses, patches, mono, intr = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics)
    t3, w = f32(np.asarray(g.targets3)[keep]), f32(np.asarray(g.weights_pose)[keep])
    ii, jj, kk = (torch.as_tensor(np.asarray(a)[keep], device=dev) for a in (g.ii, g.jj, g.kk))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan = Plan(ii, jj, kk, poses.shape[0], patches.shape[0], 1)
    plan_ms = (time.perf_counter() - t0) * 1e3
    pt = []
    for _ in range(5):
        t0 = time.perf_counter()
        p2 = Plan(ii, jj, kk, poses.shape[0], patches.shape[0], 1)
        pt.append((time.perf_counter() - t0) * 1e3)
        del p2
    st = Stepper(plan, dev)
    Po, Xo = torch.empty_like(poses), torch.empty_like(patches)
    scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
    acc = {}
    for k in range(13):
        ms = st.step_timed(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, False)
        if k >= 3:
            for n, v in ms.items(): acc.setdefault(n, []).append(v * 1e3)
    med = {n: float(np.median(v)) for n, v in acc.items()}
    alg = 40 * plan.E + 20 * plan.m + 72 * plan.n_all
    print(f"ragged E={plan.E} tiles={plan.tiles} {plan.jacobian_kernel}: " + " ".join(f"{n}={v:.1f}us" for n, v in med.items()) + f" | {alg/med['tile']/1e3/8000*100:.1f}% of 8 TB/s | slots {plan.slots} plan {plan_ms:.1f} ms first, {np.median(pt):.1f} ms median of 5", flush=True)
