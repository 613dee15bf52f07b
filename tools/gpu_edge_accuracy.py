#!/usr/bin/env python3
"""HIP step against the float64 oracle on generated graphs of the benchmark shape (slot-uniform tiles), through whichever
Jacobian kernel the environment selects (BT_FORCE=kernel=k_edge2 forces k_edge2, kernel=k_stream / k_tile pick the
others): reduced system, camera update and the state update.  GPU box:  python tools/gpu_edge_accuracy.py [M ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
from batrack_amd import graphgen  # noqa: E402
from gpu_util import HipProblem, rel, update_err  # noqa: E402

for M in [int(x) for x in (sys.argv[1:] or ["64", "256"])]:
    for frames, fixedp in ((16, 1), (64, 1), (64, 2)):
        g = graphgen.make_graph(frames, M, 8, seed=3)
        f = lambda a: np.asarray(a, np.float32).astype(np.float64)      # the oracle sees the float32 inputs the kernels see
        g.poses, g.patches, g.mono_disp, g.intrinsics, g.targets3, g.weights, g.weights_pose = (
            f(a) for a in (g.poses, g.patches, g.mono_disp, g.intrinsics, g.targets3, g.weights, g.weights_pose))
        d = dict(poses=g.poses, patches=g.patches, mono=g.mono_disp, intrinsics=g.intrinsics, targets3=g.targets3,
                 weights=g.weights, weights_pose=g.weights_pose, ii=g.ii, jj=g.jj, kk=g.kk, bounds=g.bounds)
        o = HipProblem(d).raw_step("weights_pose", fixedp, False, "huber")
        r = oracle.ba_step(g.poses, g.patches, g.mono_disp, g.intrinsics, g.targets3, g.weights_pose, g.ii, g.jj, g.kk, g.bounds,
                           lmbda=1e-4, ep=10.0, alpha=0.05, fixedp=fixedp, loss="huber", want_system=True)
        n = o["plan"].n
        free = np.arange(fixedp, fixedp + n)
        act = np.unique(g.kk)
        print(f"frames={frames} M={M} fixedp={fixedp} E={len(g.ii)} tiles={o['plan'].tiles}: S {rel(np.tril(o['S_lower']), np.tril(r['S'])):.2e} "
              f"y {rel(o['y'], r['y']):.2e} dX {rel(o['dX'].reshape(-1), r['dX'].reshape(-1)):.2e} | update pose "
              f"{update_err(o['poses_out'], r['poses_out'], g.poses.astype(np.float32), free):.2e} disp "
              f"{update_err(o['patches_out'][:, 2], r['patches_out'][:, 2], g.patches[:, 2].astype(np.float32), act):.2e} status {o['status']}", flush=True)
