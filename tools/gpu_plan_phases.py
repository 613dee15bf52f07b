#!/usr/bin/env python3
"""Phases of ONE plan build (BT_PLAN_PROF / BT_PLAN_API_PROF on stderr): `large` = 8.4M edges, optional `shard` = rank 3 of 8."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan
from batrack_amd.parallel import partition_tracks, plan_range

if "large" in sys.argv:
    g, fixedp = graphgen.make_graph(64, 16384, 8, seed=0), 1
else:
    g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
own = plan_range(partition_tracks(g.kk, 8)[3], p_tot) if "shard" in sys.argv else (0, 0)
idx = [torch.as_tensor(a, device="cuda:0") for a in (g.ii, g.jj, g.kk)]
for rep in range(3):
    print(f"--- build {rep}", file=sys.stderr, flush=True)
    pl = Plan(*idx, n_buf, p_tot, fixedp, own=own)
    torch.cuda.synchronize()
    print(f"--- {pl.jacobian_kernel}, {pl.tiles} tiles, on device: {bool(pl.built_on_device)}", file=sys.stderr, flush=True)
    pl.close()
