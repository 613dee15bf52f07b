#!/usr/bin/env python3
"""Where the HOST time of a BA call goes (GPU box): cProfile over the caller loop's BA calls of a 100-frame replay."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from batrack_amd.hostenv import limit_host_threads
from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA
from batrack_amd.backend.ba import BA_rgbd_droid
limit_host_threads()
def run(frames):
    obs = SyntheticObservations(n_frames=frames, M=256, seed=0)
    trk = WindowedBA(obs, BA_rgbd_droid, SlamConfig(PATCHES_PER_FRAME=256, BUFFER_SIZE=1024), device="cuda:0")
    trk.run()
    return trk
run(30)
pr = cProfile.Profile()
pr.enable()
trk = run(100)
pr.disable()
s = trk.stats
print(f"updates={s['updates']} ba_calls={s['ba_calls']} BA time {1e3 * s['ba_seconds'] / s['updates']:.3f} ms/update, {1e6 * s['ba_seconds'] / s['ba_calls']:.1f} us/call")
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue()[:9000])
