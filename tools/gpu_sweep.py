#!/usr/bin/env python3
"""Edge-count sweep of the Jacobian kernel (k_tile): same generator as C3 (64 frames, 8
observations per track), M tracks per frame scaled.  Prints per-kernel HIP-event times and the
algorithmic-bytes bandwidth of k_tile (40 B/edge + 20 B/track + 72 B/pose)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper

dev = "cuda:0"
for M in [int(x) for x in (sys.argv[1:] or ["256", "1024", "4096", "16384"])]:
    g = graphgen.make_graph(64, M, int(os.environ.get("BT_SWEEP_K", "8")), seed=0)      # BT_SWEEP_K: observations per track (default 8)
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights_pose)
    ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
    t0 = time.perf_counter()
    plan = Plan(ii, jj, kk, poses.shape[0], patches.shape[0], 1)
    plan_ms = (time.perf_counter() - t0) * 1e3
    st = Stepper(plan, dev)
    Po, Xo = torch.empty_like(poses), torch.empty_like(patches)
    scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
    acc = {}
    for k in range(13):
        ms = st.step_timed(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, False)
        if k >= 3:
            for n, v in ms.items():
                acc.setdefault(n, []).append(v * 1e3)
    med = {n: float(np.median(v)) for n, v in acc.items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(100):
        st.step(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, False)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 100 * 1e6
    so = []
    for k in range(8):                                         # the structure-only step's pass over the edges (k_edge2u / k_stream / k_tile SO)
        ms = st.step_timed(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, True)
        if k >= 3:
            so.append(ms["tile"] * 1e3)
    med["so_tile"] = float(np.median(so))
    alg = 40 * plan.E + 20 * plan.m + 72 * plan.n_all
    print(f"E={plan.E:9d} tracks={plan.m:8d} tiles={plan.tiles:6d} plan={plan_ms:8.1f}ms | " +
          " ".join(f"{n}={v:9.2f}us" for n, v in med.items()) + f" step={wall:8.1f}us {plan.jacobian_kernel}" +
          f" | k_tile: {alg/1e6:8.2f} MB algorithmic -> {alg/med['tile']/1e3:8.1f} GB/s = {alg/med['tile']/1e3/8000*100:5.2f}% of 8 TB/s, {plan.E/med['tile']:.0f} edges/us", flush=True)
    if int(os.environ.get("BT_DEBUG_MODE", "0")) & 64:          # an edge2_phases build of k_edge2 (tools/probes/edge2_phases.hpp, tools/build_variant.sh)
        torch.cuda.synchronize()
        off = (st._lib.bt_ba_dx(plan.handle, st.ws.data_ptr()) - st.ws.data_ptr())
        raw = st.ws.cpu().numpy()
        stat_off = off + 2 * (((6 * plan.n * 4 + 64 + 255) // 256) * 256)
        names = ["tile top", "operands+project", "jacobian+W", "gather issue", "products+group_sum", "E stores+Q", "pair fma", "schur", "flush", "drain", "tiles", "tile end+rotate", "flush checks", "geometry+X0"]
        for w, nm in enumerate(("wave 0", "wave mid")):
            pf = np.frombuffer(raw[stat_off + 16 + 160 + 160 * w: stat_off + 16 + 160 + 160 * w + 112].tobytes(), dtype=np.int64)
            print(f"  k_edge2 {nm} cycles: " + " ".join(f"{n}={v}" for n, v in zip(names, pf)))
    del st, plan
    torch.cuda.empty_cache()
