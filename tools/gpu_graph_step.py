#!/usr/bin/env python3
"""Does a captured hipGraph of the step's four launches beat launching them? (C3 and the window graph, GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper

dev = torch.device("cuda:0")
for name in ("C3", "window"):
    if name == "window":
        g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
    else:
        g, fixedp = graphgen.make_config("C3", seed=0), 1
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights_pose)
    ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
    st = Stepper(Plan(ii, jj, kk, poses.shape[0], patches.shape[0], fixedp), dev)
    P = [poses.clone(), torch.empty_like(poses)]
    X = [patches.clone(), torch.empty_like(patches)]
    scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
    def two_steps():
        st.step(P[0], X[0], mono, intr, t3, 3, w, P[1], X[1], *scal, False)
        st.step(P[1], X[1], mono, intr, t3, 3, w, P[0], X[0], *scal, False)
    for _ in range(20):
        two_steps()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        two_steps()
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / 400 * 1e6
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    tc = time.perf_counter()
    with torch.cuda.stream(s):
        two_steps()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            two_steps()
    torch.cuda.synchronize()
    cap_ms = (time.perf_counter() - tc) * 1e3
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        gr.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 400 * 1e6
    print(f"{name}: launched {plain:.2f} us/step, hipGraph replay (2 steps per graph) {graph:.2f} us/step, capture+instantiate {cap_ms:.2f} ms, status {st.status()}")
