#!/usr/bin/env python3
"""Soak run of the caller loop: a long replayed sequence through BA_rgbd_droid with prefetch_plan, watching device memory,
host RSS and the solver status — plans are created / destroyed every frame, buffers come from the pools.
GPU box:  python tools/gpu_soak.py [frames] [M]"""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
from batrack_amd import evaluation
from batrack_amd.hostenv import limit_host_threads
from batrack_amd.backend.ba import BA_rgbd_droid, prefetch_plan
from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA

limit_host_threads()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
obs = SyntheticObservations(n_frames=frames, M=M, seed=2)
trk = WindowedBA(obs, BA_rgbd_droid, SlamConfig(PATCHES_PER_FRAME=M, BUFFER_SIZE=max(1024, frames + 1)), device="cuda:0", prefetch=prefetch_plan)
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
marks = []
t0 = time.perf_counter()
for f in range(frames):
    trk()
    if f % 50 == 49 or f == frames - 1:
        torch.cuda.synchronize()
        marks.append((f + 1, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, rss()))
        print(f"frame {f + 1:4d}: device allocated {marks[-1][1]:8.1f} MiB reserved {marks[-1][2]:8.1f} MiB | host max RSS {marks[-1][3]:8.1f} MiB | "
              f"ba_calls {trk.stats['ba_calls']} edges_max {trk.stats['edges_max']}", flush=True)
poses = trk.poses_[:trk.n].detach().cpu().numpy().astype(np.float64)
ate = evaluation.ate_rmse(evaluation.camera_centres(poses), obs.centres_gt())
print(f"{frames} frames in {time.perf_counter() - t0:.1f} s, ATE {ate:.4e}, finite poses: {bool(np.isfinite(poses).all())}")
if len(marks) >= 3:
    print(f"growth between frame {marks[1][0]} and {marks[-1][0]}: device {marks[-1][1] - marks[1][1]:+.1f} MiB, host RSS {marks[-1][3] - marks[1][3]:+.1f} MiB")
