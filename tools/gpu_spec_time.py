#!/usr/bin/env python3
"""Host time of the pieces of a frame's first BA call (the one that needs the new plan) in the replay's steady state:
perf_counter around Plan.shifted_spec, _build_plan, _plan_lookup, _store, Stepper.__init__ and Plan.confirm (no profiler: cProfile
multiplies the cost of the many small Python calls).  GPU box:  python tools/gpu_spec_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from batrack_amd.backend import ba as hip_ba
from batrack_amd import plan as P
from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA

acc = {}


def wrap(obj, name, key, cls=False):
    f = getattr(obj, name)
    fn = f.__func__ if cls else f

    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc.setdefault(key, []).append((time.perf_counter() - t) * 1e6); return r
    setattr(obj, name, classmethod(w) if cls else w)


wrap(P.Plan, "shifted_spec", "Plan.shifted_spec (incl. the C call)", cls=True)
wrap(hip_ba, "_build_plan", "_build_plan")
wrap(hip_ba, "_plan_lookup", "_plan_lookup")
wrap(hip_ba, "_store", "_store (incl. destroying the evicted plan)")
wrap(P.Stepper, "__init__", "Stepper.__init__")
wrap(P.Plan, "confirm", "Plan.confirm")
per, st = [[] for _ in range(8)], dict(k=0)


def timed_ba(*a, **k):
    t = time.perf_counter(); r = hip_ba.BA_rgbd_droid(*a, **k); per[st["k"] % 8].append((time.perf_counter() - t) * 1e6); st["k"] += 1; return r


obs = SyntheticObservations(n_frames=int(os.environ.get("FRAMES", 150)), M=256, seed=0)
WindowedBA(obs, timed_ba, SlamConfig(PATCHES_PER_FRAME=256, BUFFER_SIZE=1024), device="cuda:0").run()
print("host time per BA call by position in update(), median of the last 80: " + " ".join(f"{np.median(p[-80:]):.0f}" for p in per) + " us")
for k, v in acc.items():
    print(f"{k:48s} n={len(v):4d} median of the last 80 = {np.median(v[-80:]):7.1f} us")
