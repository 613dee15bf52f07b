#!/usr/bin/env python3
"""Where the first BA call of an update() (the one that needs a new plan) spends its host time: the replay's steady state,
cProfile over those calls only, and the host time of every call by its position in update()."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from batrack_amd.backend import ba as hip_ba
from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA

per = [[] for _ in range(8)]
state = dict(k=0)
pr = cProfile.Profile()
prof = os.environ.get("PROFILE", "1") != "0"
def timed_ba(*a, **k):
    first = state["k"] % 8 == 0 and state["k"] >= 8 * 60
    if first and prof:
        pr.enable()
    t = time.perf_counter(); r = hip_ba.BA_rgbd_droid(*a, **k); dt = (time.perf_counter() - t) * 1e6
    if first and prof:
        pr.disable()
    per[state["k"] % 8].append(dt); state["k"] += 1
    return r
frames = int(os.environ.get("FRAMES", 200))
obs = SyntheticObservations(n_frames=frames, M=256, seed=0)
trk = WindowedBA(obs, timed_ba, SlamConfig(PATCHES_PER_FRAME=256, BUFFER_SIZE=1024), device="cuda:0")
trk.run()
print("host time per BA call by position in update() (no synchronisation), median of the last 100: " + " ".join(f"{np.median(p[-100:]):.0f}" for p in per) + " us")
print(f"update() {1e3 * trk.stats['ba_seconds'] / trk.stats['updates']:.3f} ms")
if prof:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
