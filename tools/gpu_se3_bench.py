#!/usr/bin/env python3
"""Bandwidth of the element-wise SE3 HIP kernels (row f-1): bytes moved = inputs + outputs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd.backend import lietorch_backends as lb
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
a = 0.3 * torch.randn(B, 6, device=dev)
X = lb.expm(3, a); Y = lb.expm(3, 0.5 * a); p4 = torch.randn(B, 4, device=dev)
ops = {"expm": (lambda: lb.expm(3, a), 6 + 7), "logm": (lambda: lb.logm(3, X), 7 + 6), "inv": (lambda: lb.inv(3, X), 14),
       "mul": (lambda: lb.mul(3, X, Y), 21), "act4": (lambda: lb.act4(3, X, p4), 15), "adjT": (lambda: lb.adjT(3, X, a), 19),
       "as_matrix": (lambda: lb.as_matrix(3, X), 23)}
for name, (fn, floats) in ops.items():
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name:10s} B={B}: {dt*1e6:9.1f} us  {floats*4*B/dt/1e9:8.1f} GB/s  ({floats*4} B/element incl. output allocation)", flush=True)
