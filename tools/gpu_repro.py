#!/usr/bin/env python3
"""Repeat the reduction of the C3 inputs and report how [S | y] varies from step to step."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from batrack_amd import graphgen
from gpu_util import HipProblem, rel

g = graphgen.make_config("C3", seed=0)
f = lambda a: np.asarray(a, np.float32).astype(np.float64)
d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics), targets3=f(g.targets3),
         weights=f(g.weights), weights_pose=f(g.weights_pose), ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds, np.float64))
hp = HipProblem(d)
o = hp.raw_step("weights_pose", 1)
st, plan = o["stepper"], o["plan"]
P = hp.poses[0].contiguous(); pat = hp.patches.reshape(-1, 3).contiguous()
Pout, pout = torch.empty_like(P), torch.empty_like(pat)
tg = hp.t3[0]
args = (P, pat, hp.mono.reshape(-1), hp.intr[0], tg, tg.stride(0), hp.w["weights_pose"][0].contiguous(),
        Pout, pout, hp.bounds, 1e-4, 10.0, 0.05, "huber", False)
ref = None
D = 6 * plan.n
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    st.step(*args, phase="reduce"); torch.cuda.synchronize()
    sysv = st.system.cpu().numpy().copy()
    st.step(*args, phase="solve_update"); torch.cuda.synchronize()
    if ref is None:
        ref = sysv
    else:
        dS = np.abs(sysv - ref)
        k = int(dS.argmax())
        print(it, "rel", rel(sysv, ref), "max abs diff", dS.max(), "at", divmod(k, D) if k < D * D else ("y", k - D * D),
              "value", ref[k], "nonzero diffs", int((dS > 1e-9 * np.abs(ref).max()).sum()))
