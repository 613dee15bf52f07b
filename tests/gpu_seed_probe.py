#!/usr/bin/env python3
"""One fuzz seed taken apart (tests/test_gpu_fuzz.py: draw): where the update's error against the float64 oracle comes from — the reduced
system, the solve, the float32 dX the step keeps, or the float32 state the update is written to.
GPU box:  python tests/gpu_seed_probe.py seed [seed ...]"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import oracle
import test_gpu_fuzz as F
from gpu_util import HipProblem, rel, update_err

for seed in [int(s) for s in sys.argv[1:]]:
    d, fixedp, so, loss, wkey, desc = F.draw(seed)
    a = (d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"], d["bounds"])
    ref = oracle.ba_step(*a, fixedp=fixedp, structure_only=so, loss=loss, want_system=True, **F.ABI_SCALARS)
    ref32 = oracle.ba_step(*a, fixedp=fixedp, structure_only=so, loss=loss, want_system=True, dtype=np.float32, **F.ABI_SCALARS)
    hp = HipProblem(d)
    o = hp.raw_step(wkey, fixedp, so=so, loss=loss)
    if not so and o["plan"].n > 0:                        # the step's wall time (20 steps back to back)
        import time, torch
        st = o["stepper"]; P = hp.poses[0].contiguous(); pat = hp.patches.reshape(-1, 3).contiguous(); tg = hp.t3[0]
        Po, Xo = torch.empty_like(P), torch.empty_like(pat)
        args = (P, pat, hp.mono.reshape(-1), hp.intr[0], tg, tg.stride(0), hp.w[wkey][0].contiguous(), Po, Xo, hp.bounds, 1e-4, 10.0, 0.05, loss, False)
        for _ in range(3): st.step(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): st.step(*args)
        torch.cuda.synchronize(); print(f"  step {(time.perf_counter() - t0) / 20 * 1e6:.1f} us (n = {o['plan'].n} free poses)")
    print(desc, "| kernel", o["plan"].jacobian_kernel, "precision", o["plan"].edge_precision, "status", o.get("status"))
    if "S" not in ref or "dX" not in o:
        continue
    S = np.tril(ref["S"]) + np.tril(ref["S"], -1).T
    n = S.shape[0] // 6
    # the damped system the reference factors (ba.py:60-70): S + (ep + lm * diag S) I
    A = S + np.diag(10.0 + 1e-4 * np.diag(S))
    y = ref["y"]
    x_ref, x_hip, x_32 = ref["dX"].reshape(-1), o["dX"].astype(np.float64).reshape(-1), ref32["dX"].astype(np.float64).reshape(-1)
    x_np = np.linalg.solve(A, y)
    res = lambda x: np.linalg.norm(A @ x - y) / np.linalg.norm(y)
    print(f"  S {rel(np.tril(o['S_lower']), np.tril(ref['S'])):.2e}  y {rel(o['y'], y):.2e}  cond(A) {np.linalg.cond(A):.2e}")
    print(f"  dX vs oracle f64: hip {rel(x_hip, x_ref):.2e}  oracle f32 {rel(x_32, x_ref):.2e}  numpy solve {rel(x_np, x_ref):.2e}  "
          f"float32(oracle dX) {rel(x_ref.astype(np.float32).astype(np.float64), x_ref):.2e}")
    print(f"  residual |A x - y| / |y|: hip {res(x_hip):.2e}  oracle f64 {res(x_ref):.2e}  float32(oracle dX) {res(x_ref.astype(np.float32).astype(np.float64)):.2e}")
    print(f"  update: poses {update_err(o['poses_out'], ref['poses_out'], d['poses']):.2e}  disparities "
          f"{update_err(o['patches_out'][:, 2], ref['patches_out'][:, 2], d['patches'][:, 2]):.2e}; "
          f"oracle f32: {update_err(ref32['poses_out'], ref['poses_out'], d['poses']):.2e} "
          f"{update_err(ref32['patches_out'][:, 2], ref['patches_out'][:, 2], d['patches'][:, 2]):.2e}; |dX| max {np.abs(x_ref).max():.2e}")
