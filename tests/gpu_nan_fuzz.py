#!/usr/bin/env python3
"""Non-finite INPUTS (a NaN / inf target, weight, disparity or pose in one to three places of a random graph): what the HIP step and the
oracle make of them.  Not a parity test — the reference pins one such case (tests/golden/c1_nan.npz: a NaN target; both follow it) and
nothing else about garbage in: an inf target leaves 0 * inf = NaN in y only and the reference's poses NaN, where in the HIP
kernels the NaN reaches S too, the factorisation fails and the poses stay unmoved.  What this checks is that nothing crashes or hangs, and it counts the cases whose
finite entries or non-finite patterns differ (profiles/r06_fuzz.txt).  GPU box:  python tests/gpu_nan_fuzz.py [first_seed] [count]"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import oracle
import test_gpu_fuzz as F
from gpu_util import HipProblem, rel
bad = 0
first, count = int(sys.argv[1]), int(sys.argv[2])
for seed in range(first, first + count):
    rng = np.random.default_rng(seed + 77)
    d, fixedp, so, loss, wkey, desc = F.draw(seed)
    E = d["ii"].size
    what = str(rng.choice(["target_nan", "target_inf", "weight_nan", "patch_nan", "weight_inf", "pose_nan"]))
    k = rng.choice(E, min(E, int(rng.integers(1, 4))), replace=False)
    if what == "target_nan": d["targets3"][k, 0] = np.nan
    elif what == "target_inf": d["targets3"][k, 1] = np.inf
    elif what == "weight_nan": d[wkey][k, 0] = np.nan
    elif what == "weight_inf": d[wkey][k, 1] = np.inf
    elif what == "patch_nan": d["patches"][d["kk"][k], 2] = np.nan
    else: d["poses"][min(int(d["jj"][k[0]]), d["poses"].shape[0] - 1), 0] = np.nan
    try:
        ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"],
                             d["bounds"], fixedp=fixedp, structure_only=so, loss=loss)
        o = HipProblem(d).raw_step(wkey, fixedp, so=so, loss=loss)
        rp, rx, hp, hx = ref["poses_out"], ref["patches_out"], o["poses_out"], o["patches_out"]
        fin_r, fin_h = np.isfinite(rp).all(1), np.isfinite(hp).all(1)
        finx_r, finx_h = np.isfinite(rx).all(1), np.isfinite(hx).all(1)
        same_mask = np.array_equal(fin_r, fin_h) and np.array_equal(finx_r, finx_h)
        ep = rel(hp[fin_r & fin_h], rp[fin_r & fin_h]) if (fin_r & fin_h).any() else 0.0
        ex = rel(hx[finx_r & finx_h], rx[finx_r & finx_h]) if (finx_r & finx_h).any() else 0.0
        ok = same_mask and ep < 1e-4 and ex < 1e-4
        print(("ok  " if ok else "DIFF"), desc[:110], what, "status", o.get("status"), "oracle failed", ref["failed"], f"finite poses {int(fin_h.sum())}/{int(fin_r.sum())} patches {int(finx_h.sum())}/{int(finx_r.sum())} err {ep:.1e} {ex:.1e}", flush=True)
        bad += not ok
    except Exception as e:
        bad += 1
        print("ERR ", desc[:110], what, type(e).__name__, str(e)[:200], flush=True)
print(f"{count} seeds: {bad} differ")
