#!/usr/bin/env python3
"""When do the waves of k_tile reach its phases (C3: 256 tiles, one per CU)?  A wave-times build (-DBT_PROBE_HEADER='"../../tools/probes/wave_times.hpp"') of ba_kernels.hip
(tools/build_variant.sh, BT_LIB_PATH) makes every wave write its 100 MHz clock at its start, after the prologue, after its slots,
after the merge + Q, after the Schur product and at its end into patches_out; this runs the reduce phase of a C3 step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper

dev = "cuda:0"
g = graphgen.make_config("C3", seed=0)
f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights_pose)
ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
plan = Plan(ii, jj, kk, poses.shape[0], patches.shape[0], 1)
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
t = raw[:n, :6].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0
names = ["start", "prologue done", "slots done", "merge + Q done", "Schur product issued", "end (atomics acknowledged)"]
print(f"{plan.jacobian_kernel} tiles={plan.tiles} waves={n}")
print("percentiles (us after the first wave's start)    min      5 %    25 %    50 %    75 %    95 %     max")
for i, nm in enumerate(names):
    print(f"{nm:44s} " + " ".join(f"{np.percentile(t[:, i], p):7.2f}" for p in (0, 5, 25, 50, 75, 95, 100)))
d = np.diff(t, axis=1)
for i, nm in enumerate(["prologue", "slots", "merge + Q", "Schur product", "drain"]):
    print(f"{nm + ' (us per wave)':44s} " + " ".join(f"{np.percentile(d[:, i], p):7.2f}" for p in (0, 5, 25, 50, 75, 95, 100)))
tw = t.reshape(plan.tiles, 8, 6)
for i, nm in enumerate(["prologue", "slots", "merge + Q", "Schur product", "drain"]):
    print(f"{nm} by wave of the tile (mean us): " + " ".join(f"{(tw[:, k, i + 1] - tw[:, k, i]).mean():.2f}" for k in range(8)))
print("end by 16th of the launch (mean us): " + " ".join(f"{tw[i * 16:(i + 1) * 16, :, 5].max(axis=1).mean():.2f}" for i in range(16)))
