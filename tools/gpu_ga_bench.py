#!/usr/bin/env python3
"""Timing of the dense global-alignment losses on a Sintel-sized case (market_5: 50 frames), the
O(Q S N^2) rigidity kernel against the f32 vector peak: per pair 2 x (3 sub, 3 fma-class, 1 sqrt) + 1 sub + masks ~ 20 flop."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from test_gpu_global_refine import make_case, build

for T, N, S in ((50, 400, 12), (50, 1024, 12)):
    d = make_case(T, N, S, seed=1)
    d["grid_query_frames"] = np.arange(T).astype(np.int64)
    for half in (False, True):
        net = build(d, half=half)
        for which, name in ((1, "scale+spatial"), (3, "+ pairwise"), (7, "+ pts3d")):
            net._run(which); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                net._run(which)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 50 * 1e6
            print(f"T={T} N={N} S={S} half={half} {name:14s}: {us:9.1f} us per forward", flush=True)
    net = build(d)
    for alpha, name in ((0.0, "backward, spatial only"), (0.5, "backward, spatial + rigid")):
        net.backward(alpha); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            net.backward(alpha)
        torch.cuda.synchronize()
        print(f"T={T} N={N} S={S} {name:26s}: {(time.perf_counter() - t0) / 30 * 1e6:9.1f} us (incl. the forward pass of the scaled depth)", flush=True)
    net.trajs_scales.requires_grad_(True); net.frame_scales_.requires_grad_(True)
    opt = torch.optim.Adam([net.trajs_scales, net.frame_scales_], lr=1e-2)
    def it():
        opt.zero_grad(); net.loss(0.5).backward(); opt.step()
    it(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        it()
    torch.cuda.synchronize()
    print(f"T={T} N={N} S={S} one Adam iteration (loss, backward, step): {(time.perf_counter() - t0) / 20 * 1e6:9.1f} us", flush=True)
    # the total run_global_refine.py optimises (weights of :61-67, pose and K free): loss, backward, Adam step
    from test_gpu_global_refine import _settings
    net = build(d, **_settings("A"))
    for p_ in (net.trajs_scales, net.frame_scales_, net.pose, net.K):
        p_.requires_grad_(True)
    opt = torch.optim.Adam([{"params": [p_], "lr": 1e-2} for p_ in (net.trajs_scales, net.frame_scales_, net.pose, net.K)], lr=1e-2, betas=(0.9, 0.9))
    def itf():
        opt.zero_grad(); net.loss().backward(); opt.step()
    itf(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        itf()
    torch.cuda.synchronize()
    print(f"T={T} N={N} S={S} one Adam iteration of the FULL total (5 terms; trajs_scales, frame_scales_, pose, K): {(time.perf_counter() - t0) / 20 * 1e6:9.1f} us", flush=True)
    net.backward(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        net.backward()
    torch.cuda.synchronize()
    print(f"T={T} N={N} S={S} backward of the full total: {(time.perf_counter() - t0) / 30 * 1e6:9.1f} us (incl. the forward pass of the scaled depth)", flush=True)
    pairs = T * (S - 1) * N * N
    print(f"  pairwise work: {pairs/1e6:.1f} M pairs x ~20 flop = {pairs*20/1e9:.2f} GFLOP")
