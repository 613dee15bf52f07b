#!/usr/bin/env python3
"""Prints HIP-vs-reference errors stage by stage for the golden cases (no asserts): reduced system, camera update,
state, and the UPDATE itself over the touched poses / tracks.  Run on the GPU box:  python tools/gpu_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from gpu_util import HipProblem, rel, update_err  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [("c1", "ps_fp1", "weights_pose", 1, False, "huber", {}), ("c1", "ps_fp3", "weights_pose", 3, False, "huber", {}),
         ("c1", "so", "weights", 1, True, "huber", {}), ("c1", "triv", "weights_pose", 1, False, "trivial", {}),
         ("c1", "cauchy", "weights_pose", 1, False, "cauchy", {}), ("c1_rough", "ps_fp1", "weights_pose", 1, False, "huber", {}),
         ("c1_rough", "ps_fp2", "weights_pose", 2, False, "huber", dict(alpha=0.5, ep=100.0)), ("c1_rough", "so", "weights", 1, True, "huber", {}),
         ("window_small", "ps", "weights_pose", None, False, "huber", {}), ("window_small", "so", "weights", None, True, "huber", {})]
for name, tag, wkey, fixedp, so, loss, kw in CASES:
    d = dict(np.load(os.path.join(GOLD, name + ".npz")))
    fp = int(d["fixedp"]) if fixedp is None else fixedp
    o = HipProblem(d).raw_step(wkey, fp, so, loss, **kw)
    act = np.unique(d["kk"])
    msg = (f"{name}/{tag}: state poses {rel(o['poses_out'], d[tag + '.f64.poses_out']):.2e} patches {rel(o['patches_out'], d[tag + '.f64.patches_out']):.2e}"
           f" | update disp {update_err(o['patches_out'][:, 2], d[tag + '.f64.patches_out'][:, 2], d['patches'][:, 2].astype(np.float32), act):.2e}"
           f" (ref-f32 {update_err(d[tag + '.f32.patches_out'][:, 2], d[tag + '.f64.patches_out'][:, 2], d['patches'][:, 2].astype(np.float32), act):.2e})")
    if "S_lower" in o:
        S = d[tag + ".f64.S"]
        n = S.shape[0] // 6
        free = np.arange(fp, fp + n)
        msg += (f" pose {update_err(o['poses_out'], d[tag + '.f64.poses_out'], d['poses'].astype(np.float32), free):.2e}"
                f" (ref-f32 {update_err(d[tag + '.f32.poses_out'], d[tag + '.f64.poses_out'], d['poses'].astype(np.float32), free):.2e})"
                f" | S {rel(np.tril(o['S_lower']), np.tril(S)):.2e} y {rel(o['y'], d[tag + '.f64.y']):.2e} dX {rel(o['dX'].reshape(-1), d[tag + '.f64.dX'].reshape(-1)):.2e} status {o['status']}")
    print(msg, flush=True)
