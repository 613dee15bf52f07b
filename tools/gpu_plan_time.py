#!/usr/bin/env python3
"""Where the time of a plan build goes: host analysis vs device upload (C3, repeated)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan

if len(sys.argv) > 1 and sys.argv[1] == "window":
    g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
elif len(sys.argv) > 1 and sys.argv[1] == "large":          # 8.4M edges, 16384 tiles: planned on the device since round 4
    g, fixedp = graphgen.make_graph(64, 16384, 8, seed=0), 1
    print("# 64 keyframes x 16384 tracks per frame x 8 observations = 8.4M edges")
else:
    g, fixedp = graphgen.make_config("C3", seed=0), 1
n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
dev = "cuda:0"
ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
for name, args, kw in (("host arrays, no upload", (g.ii, g.jj, g.kk), dict(upload=False)),
                       ("host arrays, upload", (g.ii, g.jj, g.kk), dict(upload=True)),
                       ("device tensors, upload", (ii, jj, kk), dict(upload=True))):
    ts = []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl = Plan(*args, n_buf, p_tot, fixedp, **kw)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        pl.close()
    print(f"{name:26s}: first {ts[0]:.2f} ms, then median {np.median(ts[1:]):.2f} ms")

if len(sys.argv) > 1 and sys.argv[1] in ("window", "large"):
    # a rank's plan of a sharded solve (world 8, rank 3): the device's passes run on the rank's segment of the sorted list
    from batrack_amd.parallel import partition_tracks, plan_range
    own = plan_range(partition_tracks(g.kk, 8)[3], p_tot)
    for name, args in (("sharded 3/8, host arrays", (g.ii, g.jj, g.kk)), ("sharded 3/8, device tensors", (ii, jj, kk))):
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pl = Plan(*args, n_buf, p_tot, fixedp, own=own)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            note = f"  ({pl.jacobian_kernel}, {pl.tiles} tiles, built on device: {bool(pl.built_on_device)})"
            pl.close()
        print(f"{name:26s}: first {ts[0]:.2f} ms, then median {np.median(ts[1:]):.2f} ms" + note)

from batrack_amd.plan import Stepper
ts = []
for _ in range(6):
    pl = Plan(ii, jj, kk, n_buf, p_tot, fixedp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = Stepper(pl, dev)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"Stepper (workspace) creation: first {ts[0]:.2f} ms, then median {np.median(ts[1:]):.2f} ms")
