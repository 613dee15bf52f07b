import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan
dev = "cuda:0"
g = graphgen.make_graph(64, 16384, 8, seed=0)
for drop in (0.0, 0.15):
    keep = np.random.default_rng(11).random(np.asarray(g.kk).size) > drop
    ii, jj, kk = (torch.as_tensor(np.asarray(a)[keep], device=dev) for a in (g.ii, g.jj, g.kk))
    for k in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = Plan(ii, jj, kk, g.poses.shape[0], g.patches.shape[0], 1)
        print("drop", drop, "plan ms", (time.perf_counter() - t0) * 1e3, p.jacobian_kernel, p.slots, flush=True)
        del p
