#!/usr/bin/env python3
"""Host-side profile of the replayed caller loop with the HIP BA (GPU box): where a frame's time goes."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd.backend.ba import BA_rgbd_droid
from batrack_amd.hostenv import limit_host_threads
print('host threads:', limit_host_threads())
from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA

frames, M = int(os.environ.get("FRAMES", 50)), int(os.environ.get("M", 256))
for rep in range(2):
    obs = SyntheticObservations(n_frames=frames, M=M, seed=0)
    trk = WindowedBA(obs, BA_rgbd_droid, SlamConfig(PATCHES_PER_FRAME=M, BUFFER_SIZE=int(os.environ.get("BUFFER", frames + 1))), device="cuda:0")
    if rep == 1:
        pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter(); trk.run(); wall = time.perf_counter() - t0
    if rep == 1:
        pr.disable()
    s = trk.stats
    print(f"rep {rep}: wall {wall:.3f}s, BA {s['ba_seconds']:.3f}s over {s['updates']} updates = {1e3*s['ba_seconds']/s['updates']:.3f} ms/update")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

# per-call wall time of plan construction and of the first step after it
from batrack_amd import plan as _plan
_orig = _plan.Plan.__init__
_times = []
def _timed(self, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); _orig(self, *a, **k); _times.append((time.perf_counter() - t) * 1e3)
_plan.Plan.__init__ = _timed
obs = SyntheticObservations(n_frames=frames, M=M, seed=0)
trk = WindowedBA(obs, BA_rgbd_droid, SlamConfig(PATCHES_PER_FRAME=M, BUFFER_SIZE=int(os.environ.get("BUFFER", frames + 1))), device="cuda:0")
trk.run()
print("Plan() wall ms per call:", " ".join(f"{t:.2f}" for t in _times))

# wall time per BA call (synchronised), split by position inside update(): call 0 builds the plan
_plan.Plan.__init__ = _orig
per = [[] for _ in range(8)]
state = dict(k=0)
def timed_ba(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = BA_rgbd_droid(*a, **k); torch.cuda.synchronize()
    per[state["k"] % 8].append((time.perf_counter() - t) * 1e6); state["k"] += 1
    return r
obs = SyntheticObservations(n_frames=frames, M=M, seed=0)
trk = WindowedBA(obs, timed_ba, SlamConfig(PATCHES_PER_FRAME=M, BUFFER_SIZE=int(os.environ.get("BUFFER", frames + 1))), device="cuda:0")
trk.run()
import numpy as np
print("median us per BA call by position in update():", " ".join(f"{np.median(p[-30:]):.0f}" for p in per))
print(f"update() total {1e3 * trk.stats['ba_seconds'] / trk.stats['updates']:.3f} ms")
print("mean us per position:", " ".join(f"{np.mean(p):.0f}" for p in per), "| max:", " ".join(f"{np.max(p):.0f}" for p in per))
print("sum of timed calls %.3f s of update total %.3f s" % (sum(map(sum, per)) * 1e-6, trk.stats["ba_seconds"]))
print("position-0 calls (ms):", " ".join(f"{t/1e3:.1f}" for t in per[0]))

# the map-filtering reprojection over the final edge list: fused kernel vs composed tensor operations
from batrack_amd.backend import projective_ops as pops
from batrack_amd.backend.lietorch import SE3
for fused in (True, False):
    for _ in range(3):
        pops.transform(SE3(trk.poses), trk.patches, trk.intrinsics, trk.ii, trk.jj, trk.kk, fused=fused)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        pops.transform(SE3(trk.poses), trk.patches, trk.intrinsics, trk.ii, trk.jj, trk.kk, fused=fused)
    torch.cuda.synchronize()
    print(f"transform over {trk.ii.numel()} edges, fused={fused}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us")
