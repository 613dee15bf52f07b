#!/usr/bin/env python3
"""Per-kernel HIP-event timings of one BA step for a workload (GPU box)."""
import os, sys, argparse
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--reps", type=int, default=50)
args = ap.parse_args()
dev = "cuda:0"
if args.workload == "window":
    g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4, n_buf=int(os.environ.get("BUFFER", 50)), window=int(os.environ.get("WINDOW", 12)), removal=int(os.environ.get("REMOVAL", 20)))
else:
    g, fixedp = graphgen.make_config(args.workload, seed=0), 1
f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights_pose)
ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
plan = Plan(ii, jj, kk, poses.shape[0], patches.shape[0], fixedp)
st = Stepper(plan, dev)
Po, Xo = torch.empty_like(poses), torch.empty_like(patches)
scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
acc = {}
for k in range(args.reps + 5):
    ms = st.step_timed(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, False)
    if k >= 5:
        for n, v in ms.items():
            acc.setdefault(n, []).append(v * 1e3)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for k in range(200):
    st.step(poses, patches, mono, intr, t3, 3, w, Po, Xo, *scal, False)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 200 * 1e6
if int(os.environ.get("BT_DEBUG_MODE", "0")) & 32:
    off = (st._lib.bt_ba_dx(plan.handle, st.ws.data_ptr()) - st.ws.data_ptr())
    raw = st.ws.cpu().numpy()
    stat_off = off + 2 * (((6 * plan.n * 4 + 64 + 255) // 256) * 256)            # dx, dx0, then the status block
    pf = np.frombuffer(raw[stat_off + 16 + 160: stat_off + 16 + 320].tobytes(), dtype=np.int64).reshape(2, 10)
    names = ["prologue", "loadwait", "math+E", "pairreduce", "CwQ", "Esave", "mfma", "epilogue", "drain", "-"]
    if plan.jacobian_kernel == "k_etile":
        names = ["prologue", "rounds", "pair-sum merge (2 barriers)", "Schur + stores", "drain", "pair sums -> workspace", "stride_sum", "-", "-", "-"]
    for w, nm in enumerate(("tile0", "tileMid")):
        print(f"  k_tile {nm} wave0 cycles: " + " ".join(f"{n}={v}" for n, v in zip(names, pf[w])) + f" total={pf[w].sum()}")
if int(os.environ.get("BT_DEBUG_MODE", "0")) & 16:
    import ctypes
    off = (st._lib.bt_ba_dx(plan.handle, st.ws.data_ptr()) - st.ws.data_ptr())
    raw = st.ws.cpu().numpy()
    # status region follows dx region: find by scanning from plan layout is not exposed; use the status call offset
    stat_off = off + 2 * (((6 * plan.n * 4 + 64 + 255) // 256) * 256)            # dx, dx0, then the status block
    pf = np.frombuffer(raw[stat_off + 16: stat_off + 16 + 160].tobytes(), dtype=np.int64).reshape(2, 10)
    names = ["load", "updates", "chol", "trsm", "store", "barrier", "Mprep", "backsub", "tail", "-"]
    ph = np.frombuffer(raw[stat_off + 16 + 320: stat_off + 16 + 320 + 16 * 12].tobytes(), dtype=np.int64).reshape(12, 2)
    print("  solver per-wave busy cycles (phase1, phase2): " + " ".join(f"w{w}:{a}/{b}" for w, (a, b) in enumerate(ph)))
    forced_solver = dict(t.partition("=")[::2] for t in os.environ.get("BT_FORCE", "").split(",") if t).get("solver")
    if forced_solver is None:
        g = pf.reshape(-1)
        print("  pipe solver per-wave (waiting, working) cycles: " + " ".join(f"w{w}:{a}/{b}" for w, (a, b) in enumerate(ph)))
        print(f"  pipe solver: load={g[0]} sweep_end={g[1]} total={g[2]} | row wave 2: row_update={g[13]} wait_diag={g[14]} chol+trsm={g[15]} waits_at_level_start={g[16]} (for a column {g[17]}, for the helpers {g[18]}) | diagonal wave 0: waits for a column {g[7]}, for the helpers {g[8]}")
    elif forced_solver == "fused":
        g = pf.reshape(-1)
        print(f"  fused solver: load={g[0]} sweep_end={g[1]} total={g[2]} | tail: Linv_end={g[3]} Mform_end={g[4]} backsub_end={g[5]} | row wave: row_update={g[16]} wait+load+chol={g[17]} trsm+store={g[18]}")
    for w in range(2):
        print(f"  solver wave{w} cycles: " + " ".join(f"{n}={v}" for n, v in zip(names, pf[w])) + f" total={pf[w].sum()}")
print(f"mode={os.environ.get('BT_DEBUG_MODE','0')} {args.workload} E={plan.E} n={plan.n} tiles={plan.tiles} nnzb={plan.nnz_blocks}: " +
      " ".join(f"{n}={np.median(v):.2f}us" for n, v in acc.items()) + f" | wall/step={wall:.1f}us status={st.status()}", flush=True)
