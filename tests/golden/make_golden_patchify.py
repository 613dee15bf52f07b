#!/usr/bin/env python3
"""Generates tests/golden/patchify.npz by running the reference's own `altcorr.patchify`
(/root/reference/main/backend/altcorr/correlation.py:51-68, imported unmodified) on seeded inputs.  The compiled gather it
calls (cuda_corr.patchify_forward, CUDA) is replaced by the stand-in of tests/golden/refstubs/cuda_corr — so these vectors
pin the blend (weights, order of the four products and of the sum, the 'nearest' pass-through, shapes) to the reference's
code and the gather to our reading of correlation_kernel.cu:16-47.  Run in the build container only:
    python tests/golden/make_golden_patchify.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refstubs"))
spec = importlib.util.spec_from_file_location("ref_correlation", "/root/reference/main/backend/altcorr/correlation.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
CASES = [(0, "bilinear"), (1, "bilinear"), (0, "nearest"), (2, "nearest"), (3, "bilinear")]
for n, (R, mode) in enumerate(CASES):
    rng = np.random.default_rng(100 + n)
    B, C, H, W, M = 2, 3, 29, 41, 96
    net = rng.standard_normal((B, C, H, W)).astype(np.float32)
    coords = np.stack([rng.uniform(-3, W + 3, (B, M)), rng.uniform(-3, H + 3, (B, M))], -1).astype(np.float32)
    coords[0, :6] = [[0, 0], [W - 1, H - 1], [10.0, 20.0], [W - 0.5, H - 0.5], [-0.5, 3.25], [7.75, -2.0]]
    res = ref.patchify(torch.as_tensor(net), torch.as_tensor(coords), R, mode=mode)
    out[f"case{n}.net"], out[f"case{n}.coords"], out[f"case{n}.out"] = net, coords, res.numpy()
    out[f"case{n}.R"], out[f"case{n}.mode"] = np.int64(R), np.array(mode)
# the two uses of the caller: colour at coords + 0.5 and depth, radius 0 (batrack.py:323,438)
rng = np.random.default_rng(7)
img = (rng.uniform(0, 255, (1, 3, 24, 32))).astype(np.float32)
cc = (rng.uniform(0, 1, (1, 50, 2)) * np.array([31.0, 23.0])).astype(np.float32)
out["caller.img"], out["caller.coords"] = img, cc
out["caller.clr"] = ref.patchify(torch.as_tensor(img), torch.as_tensor(cc) + 0.5, 0).view(1, -1, 3).numpy()
np.savez_compressed(os.path.join(HERE, "patchify.npz"), **out)
print("wrote patchify.npz:", {k: v.shape for k, v in out.items() if k.endswith(".out")})
