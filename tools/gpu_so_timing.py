import os, sys, time
sys.path[:0]=["/root/repo"]
import numpy as np, torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper
dev="cuda:0"
for wl in ("window","C3"):
    if wl=="window":
        g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
    else:
        g, fixedp = graphgen.make_config("C3", seed=0), 1
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights)
    ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
    plan = Plan(ii, jj, kk, poses.shape[0], patches.shape[0], fixedp)
    st = Stepper(plan, dev)
    Po, Xo = torch.empty_like(poses), torch.empty_like(patches)
    scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
    acc={}
    for k in range(30):
        ms = st.step_timed(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, True)
        if k>=5:
            for n,v in ms.items(): acc.setdefault(n,[]).append(v*1e3)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for k in range(300): st.step(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, True)
    torch.cuda.synchronize(); wall=(time.perf_counter()-t0)/300*1e6
    print(wl, plan.jacobian_kernel, {n: round(float(np.median(v)),2) for n,v in acc.items()}, "wall/step %.1f us"%wall, "p_tot", patches.shape[0])
