#!/usr/bin/env python3
"""When do the waves of k_edge2 start and end?  A wave-times build (tools/build_variant.sh times ba_edge2.hip -DBT_PROBE_HEADER='"../../tools/probes/wave_times.hpp"', BT_LIB_PATH) makes every wave write
its start, the end of its tiles and its end (100 MHz clock) into patches_out; this runs the reduce phase of a step on the
benchmark generator's graph (M tracks per frame) and prints the distribution relative to the first start."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = graphgen.make_graph(64, M, 8, seed=0)
f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights_pose)
ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
plan = Plan(ii, jj, kk, poses.shape[0], patches.shape[0], 1)
st = Stepper(plan, dev)
Po, Xo = torch.empty_like(poses), torch.empty_like(patches)
scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
for k in range(6):
    Xo.zero_()
    st.step(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, False, phase="reduce")
    torch.cuda.synchronize()
raw = Xo.cpu().numpy().reshape(-1)
raw = raw[: raw.size // 8 * 8].view(np.int64).reshape(-1, 4)
n = int((raw[:, 0] != 0).sum())
t = raw[:n].astype(np.float64)
t0 = t[:, 0].min()
start, tiles_end, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0       # us
hw = raw[:n, 3]
print(f"{plan.jacobian_kernel} tiles={plan.tiles} waves={n}")
q = lambda x: " ".join(f"{np.percentile(x, p):7.1f}" for p in (0, 5, 25, 50, 75, 95, 100))
print("percentiles (us after the first wave's start)    min      5 %    25 %    50 %    75 %    95 %     max")
print("start                                        " + q(start))
print("end of the wave's tiles                      " + q(tiles_end))
print("end of the wave                              " + q(end))
print("tiles phase of a wave (us)                   " + q(tiles_end - start))
print("combine + atomics of a wave (us)             " + q(end - tiles_end))
xcc = (hw >> 0) & 0xffffffff
late = np.argsort(end)[-16:]
print("the 16 last waves: index, start, end: " + " ".join(f"{i}:{start[i]:.0f}-{end[i]:.0f}" for i in late))
tp = tiles_end - start
wpf = max(1, n // 64)                                   # waves per source frame (the generator's tiles are frame-major)
print("tiles phase by 64th of the launch (consecutive waves, us): " + " ".join(f"{tp[f * wpf:(f + 1) * wpf].mean():.0f}" for f in range(min(64, n // wpf))))
print("tiles phase by wave of the workgroup (index mod 2): " + " ".join(f"{tp[k::2].mean():.1f}" for k in range(2)))
print("tiles phase by workgroup mod 8 (XCD): " + " ".join(f"{tp[(np.arange(n) // 2) % 8 == k].mean():.1f}" for k in range(8)))
print("tiles phase by workgroup mod 32 : " + " ".join(f"{tp[(np.arange(n) // 2) % 32 == k].mean():.0f}" for k in range(32)))
hid = raw[:n, 3] & 0xffffffff
xcc = (raw[:n, 3] >> 32) & 0xf
simd, cu, se = (hid >> 4) & 3, (hid >> 8) & 15, (hid >> 13) & 7
gen = np.arange(n) // (n // 4) if n >= 4 else np.zeros(n, int)
print("HW_ID: tiles phase by SIMD: " + " ".join(f"{tp[simd == k].mean():.1f}({(simd == k).sum()})" for k in range(4)))
print("by XCC: " + " ".join(f"{tp[xcc == k].mean():.1f}({(xcc == k).sum()})" for k in range(8)))
for g_ in range(4):
    m = gen == g_
    print(f"quarter {g_} of the launch (waves {m.nonzero()[0][0]}..{m.nonzero()[0][-1]}): tiles phase {tp[m].mean():.1f} us; SIMDs of its waves " +
          " ".join(f"{k}:{(simd[m] == k).sum()}" for k in range(4)) + f"; distinct (xcc, se, cu): {len(set(zip(xcc[m], se[m], cu[m])))}")
# per physical CU: the order in which its waves started and their times
key = xcc * 1000 + se * 100 + cu
ex = key == key[0]
print("one CU's waves (index, simd, start, tiles phase): " + " ".join(f"{i}/s{simd[i]}/{start[i]:.2f}/{tp[i]:.0f}" for i in np.nonzero(ex)[0]))
