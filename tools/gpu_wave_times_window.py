#!/usr/bin/env python3
"""When do the waves of k_etile (the sliding window's Jacobian kernel) start and end?  A wave-times build (-DBT_PROBE_HEADER='"../../tools/probes/wave_times.hpp"'; tools/build_variant.sh,
BT_LIB_PATH) makes every wave write its 100 MHz clock at its start, after the prologue, after its rounds, after the pair-sum merge
and at its end into patches_out; this runs the reduce phase of a step on the real-shape window graph."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper

dev = "cuda:0"
g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4, n_buf=50, window=12, removal=20)
f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights_pose)
ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
plan = Plan(ii, jj, kk, poses.shape[0], patches.shape[0], fixedp)
st = Stepper(plan, dev)
Po, Xo = torch.empty_like(poses), torch.empty_like(patches)
scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
for k in range(8):
    Xo.zero_()
    st.step(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, False, phase="reduce")
    torch.cuda.synchronize()
raw = Xo.cpu().numpy().reshape(-1)
raw = raw[: raw.size // 16 * 16].view(np.int64).reshape(-1, 8)
n = plan.tiles * 8
t = raw[:n, :5].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0
names = ["start", "prologue done", "rounds done", "merge done", "end"]
print(f"{plan.jacobian_kernel} tiles={plan.tiles} waves={n}")
print("percentiles (us after the first wave's start)    min      5 %    25 %    50 %    75 %    95 %     max")
for i, nm in enumerate(names):
    print(f"{nm:44s} " + " ".join(f"{np.percentile(t[:, i], p):7.2f}" for p in (0, 5, 25, 50, 75, 95, 100)))
d = np.diff(t, axis=1)
for i, nm in enumerate(["prologue", "rounds", "wait + merge", "stores + Schur"]):
    print(f"{nm + ' (us per wave)':44s} " + " ".join(f"{np.percentile(d[:, i], p):7.2f}" for p in (0, 5, 25, 50, 75, 95, 100)))
tw = t.reshape(plan.tiles, 8, 5)
print("rounds by wave of the tile (mean us): " + " ".join(f"{(tw[:, k, 2] - tw[:, k, 1]).mean():.2f}" for k in range(8)))
info = raw[:n, 5].reshape(plan.tiles, 8)[:, 0]
ntrk, Dt, nit = info >> 32, (info >> 8) & 0xff, info & 0xff
for d_ in sorted(set(Dt.tolist())):
    m_ = Dt == d_
    print(f"tiles with {d_} rounds per lane: {int(m_.sum()):3d}, tracks {int(ntrk[m_].min())}..{int(ntrk[m_].max())}; rounds done at {tw[m_, :, 2].max(axis=1).mean():.2f} us (slowest wave, mean over the tiles), end {tw[m_, :, 4].max(axis=1).mean():.2f}")
print("start by tile, first 16 / last 16 workgroups (us): " + " ".join(f"{x:.2f}" for x in tw[:16, 0, 0]) + " ... " + " ".join(f"{x:.2f}" for x in tw[-16:, 0, 0]))
