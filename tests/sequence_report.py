#!/usr/bin/env python3
"""Full-size replay of the caller loop (sintel.yaml window: 256 tracks/frame, S_slam 12, ITER 4) on the
GPU box: HIP BA vs the CPU oracle on the same synthetic sequence — ATE of both, their difference, and where
the time of a frame goes (plan build per new edge list vs the 2*ITER BA calls).  Writes the text that is
committed as profiles/rNN_sequence_ate.txt.   python tests/sequence_report.py [--frames 50] [--M 256]"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402
from batrack_amd import evaluation  # noqa: E402
from batrack_amd.hostenv import limit_host_threads  # noqa: E402
from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA  # noqa: E402
from sequence_util import oracle_BA_rgbd_droid  # noqa: E402
from oracle.se3_torch import SE3Ref  # noqa: E402

limit_host_threads()
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=50)
ap.add_argument("--M", type=int, default=256)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--buffer", type=int, default=1024, help="BUFFER_SIZE (sintel.yaml: 1024 pose slots, x M patch slots)")
ap.add_argument("--skip-oracle", action="store_true")
args = ap.parse_args()

from batrack_amd.backend.ba import BA_rgbd_droid  # noqa: E402
rows = {}
from batrack_amd.backend.ba import prefetch_plan  # noqa: E402
for name, ba, dev in (("hip", BA_rgbd_droid, "cuda:0"), ("hip+prefetch", BA_rgbd_droid, "cuda:0"), ("oracle", oracle_BA_rgbd_droid, "cpu")):
    if name == "oracle" and args.skip_oracle:
        continue
    for rep in range(2 if name.startswith("hip") else 1):            # second HIP run: warm allocator / code objects
        obs = SyntheticObservations(n_frames=args.frames, M=args.M, seed=args.seed)
        trk = WindowedBA(obs, ba, SlamConfig(PATCHES_PER_FRAME=args.M, BUFFER_SIZE=max(args.buffer, args.frames + 1)), device=dev,
                         prefetch=prefetch_plan if name == "hip+prefetch" else None, **({} if dev != "cpu" else dict(se3=SE3Ref)))
        t0 = time.perf_counter()
        poses = trk.run()
        wall = time.perf_counter() - t0
    rows[name] = dict(poses=poses, wall=wall, stats=trk.stats,
                      ate=evaluation.ate_rmse(evaluation.camera_centres(poses), obs.centres_gt()))
    s = trk.stats
    print(f"{name:12s}: frames={args.frames} M={args.M} buffer={max(args.buffer, args.frames + 1)} updates={s['updates']} ba_calls={s['ba_calls']} edges_max={s['edges_max']} "
          f"ATE={rows[name]['ate']:.6e}  BA time={s['ba_seconds']:.3f}s ({1e3 * s['ba_seconds'] / s['updates']:.3f} ms/update, "
          f"{1e6 * s['ba_seconds'] / s['ba_calls']:.1f} us/call incl. plan builds)  loop wall={wall:.2f}s", flush=True)
if "oracle" in rows:
    a, b = rows["hip"]["ate"], rows["oracle"]["ate"]
    print(f"ATE difference: {abs(a - b):.3e} = {100 * abs(a - b) / b:.4f} % of the oracle-driven ATE (bar: 1 %)")
    print(f"max |pose_hip - pose_oracle| = {np.abs(rows['hip']['poses'] - rows['oracle']['poses']).max():.3e}")
    print(f"BA time per update(): oracle {1e3 * rows['oracle']['stats']['ba_seconds'] / rows['oracle']['stats']['updates']:.2f} ms"
          f" vs HIP {1e3 * rows['hip']['stats']['ba_seconds'] / rows['hip']['stats']['updates']:.3f} ms")
