#!/usr/bin/env python3
"""Race hunt for the barrier-free solver: the same system solved many times must give the same dX
(bit for bit: the solver is deterministic given [S | y]), on C3 and on a few other graphs."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from batrack_amd import graphgen
from gpu_util import HipProblem

def inputs(g):
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    return dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics), targets3=f(g.targets3),
                weights=f(g.weights), weights_pose=f(g.weights_pose), ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds, np.float64))

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for name, g, fp in (("C3", graphgen.make_config("C3", seed=0), 1), ("band48", graphgen.make_graph(48, 16, 8, seed=3), 1),
                    ("band12", graphgen.make_graph(12, 8, 4, seed=3), 1), ("C1", graphgen.make_config("C1", seed=0), 1)):
    hp = HipProblem(inputs(g))
    o = hp.raw_step("weights_pose", fp)
    st = o["stepper"]
    P = hp.poses[0].contiguous(); pat = hp.patches.reshape(-1, 3).contiguous()
    Pout, pout = torch.empty_like(P), torch.empty_like(pat)
    tg = hp.t3[0]
    args = (P, pat, hp.mono.reshape(-1), hp.intr[0], tg, tg.stride(0), hp.w["weights_pose"][0].contiguous(),
            Pout, pout, hp.bounds, 1e-4, 10.0, 0.05, "huber", False)
    st.step(*args, phase="reduce"); torch.cuda.synchronize()
    sys0 = st.system.clone()
    ref, bad = None, 0
    for it in range(reps):
        st.system.copy_(sys0)                       # identical input every time
        st.step(*args, phase="solve_update")
        torch.cuda.synchronize()
        dx = st.dx.cpu().numpy().copy()
        if ref is None:
            ref = dx
        elif not np.array_equal(dx, ref):
            bad += 1
            if bad <= 3:
                print(f"  {name}: iteration {it}: max |ddX| = {np.abs(dx - ref).max():.3e} (|dX| max {np.abs(ref).max():.3e})")
    print(f"{name}: n={o['plan'].n} {reps} solves of the same system, {bad} differed, status {st.status()}")
