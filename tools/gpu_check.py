#!/usr/bin/env python3
"""Prints HIP-vs-reference errors stage by stage for the golden cases (no asserts).
Run on the GPU box:  python tools/gpu_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from gpu_util import HipProblem, rel  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
for name, tag, wkey, fixedp, so in [("c1", "ps_fp1", "weights_pose", 1, False), ("c1", "ps_fp3", "weights_pose", 3, False),
                                    ("c1", "so", "weights", 1, True), ("c1_rough", "ps_fp1", "weights_pose", 1, False),
                                    ("window_small", "ps", "weights_pose", None, False)]:
    d = dict(np.load(os.path.join(GOLD, name + ".npz")))
    fp = int(d["fixedp"]) if fixedp is None else fixedp
    o = HipProblem(d).raw_step(wkey, fp, so)
    msg = f"{name}/{tag}: poses {rel(o['poses_out'], d[tag + '.f64.poses_out']):.2e} patches {rel(o['patches_out'], d[tag + '.f64.patches_out']):.2e}"
    if "S_lower" in o:
        S = d[tag + ".f64.S"]
        msg += f" | S {rel(np.tril(o['S_lower']), np.tril(S)):.2e} y {rel(o['y'], d[tag + '.f64.y']):.2e} dX {rel(o['dX'].reshape(-1), d[tag + '.f64.dX'].reshape(-1)):.2e} status {o['status']}"
        n = S.shape[0] // 6
        blk = np.abs(np.tril(o["S_lower"]) - np.tril(S)).reshape(n, 6, n, 6).max(axis=(1, 3)) / np.abs(S).max()
        msg += f" | worst block {np.unravel_index(blk.argmax(), blk.shape)} {blk.max():.2e}"
    msg += f" | ref-f32 poses {rel(d[tag + '.f32.poses_out'], d[tag + '.f64.poses_out']):.2e}"
    print(msg, flush=True)
