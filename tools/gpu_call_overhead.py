#!/usr/bin/env python3
"""Host cost of a BA call against its GPU time (GPU box): the caller loop's 2 x ITER calls on a window graph with the plan
cached — time to enqueue them (no sync), and to their completion."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from batrack_amd import graphgen
from batrack_amd.backend.ba import BA_rgbd_droid
from batrack_amd.backend.lietorch import SE3
from batrack_amd.hostenv import limit_host_threads
limit_host_threads()
dev = "cuda:0"
g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
N, P = g.poses.shape[0], g.patches.shape[0]
poses, patches = f32(g.poses).view(1, N, 7), f32(g.patches).view(1, P, 3, 1, 1)
mono, intr = f32(g.mono_disp).view(1, P, 1), f32(g.intrinsics)
t3, w, wp = f32(g.targets3).view(1, -1, 3), f32(g.weights).view(1, -1, 2), f32(g.weights_pose).view(1, -1, 2)
ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
bounds = list(g.bounds)
def update(iters=4):
    Gs, pat = SE3(poses), patches
    for _ in range(iters):
        Gs, pat = BA_rgbd_droid(Gs, pat, mono, intr, t3[..., :2], t3[..., 2:], wp, 1e-4, ii, jj, kk, bounds, ep=10, fixedp=fixedp, structure_only=False, loss="huber", alpha=0.05)
        Gs, pat = BA_rgbd_droid(Gs, pat, mono, intr, t3[..., :2], t3[..., 2:], w, 1e-4, ii, jj, kk, bounds, ep=10, fixedp=fixedp, structure_only=True, loss="huber", alpha=0.05)
    return Gs, pat
for _ in range(5): update()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(100):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); update(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print(f"8 calls (4 dual iterations), plan cached: enqueue {np.median(enq)*1e6:.1f} us = {np.median(enq)*1e6/8:.1f} us per call on the host; done after {np.median(tot)*1e6:.1f} us")
if os.environ.get("BT_CALL_PROFILE"):
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): update()
    torch.cuda.synchronize(); pr.disable()
    out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(25); print(out.getvalue()[:6000])
