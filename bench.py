#!/usr/bin/env python3
"""bench.py — BA iterations/s on the 64-KF / 131,072-edge factor graph (BASELINE.json
`metric`; workload C3 of SURVEY.md §8d), on N GPUs of one node.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one BA iteration = one pose+structure `BA_rgbd_droid` call (ba.py:217)
executed through the C ABI (bt_ba_step) with inputs resident in HBM; state is
ping-ponged so the K timed steps are K real Gauss-Newton iterations.  The plan of
the (fixed) edge list is built before the timed region (its cost is reported as
`plan_build_ms`), exactly as the reference's caller reuses one edge list for
2*ITER calls (batrack.py:869-875).  N > 1: tracks are sharded over the ranks, one
RCCL all-reduce of the reduced system's non-zero blocks per step (batrack_amd/parallel.py); the
graph is the same, so scaling is "strong".

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel named in
BASELINE.json (the Jacobian kernel, k_tile): algorithmic bytes (SURVEY.md §8d:
40 B/edge + 20 B/track + 72 B/pose) over the kernel's own duration measured with
HIP events recorded by the launch on its stream (bt_ba_step_timed); `cold_kernel_us` / `cold_frac` the
same behind a 512 MB sweep (L2 and Infinity Cache hold nothing of the graph); `roofline.large` the same
two numbers for the 8.4M-edge graph of the same generator, where the kernel (k_edge2) streams.
`ms_per_step_blocks`: the K steps timed five more times (min / median / max); `value_large`: the 2.1M-edge
graph's iterations/s (the workload that shards).
`cpu_baseline` = `oracle.refseq`, the torch-CPU restatement that keeps the reference's
operator sequence, timed on this host at 8 threads and at the
container's CPU quota (rank 0, N = 1 only); the scalar C port of the checker is reported beside it.
Exactly --steps steps are timed (default 1000: 80 ms of C3 steps).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)     # (80 ms of C3 steps; the K given is the K timed)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3", choices=["C1", "C3"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--probe-ipc", action="store_true", help="internal: child process of an N > 1 run (see ipc_probe)")
    ap.add_argument("--no-large", action="store_true", help="skip the 2.1M-edge sharded workload reported beside the headline")
    return ap.parse_args()


def ipc_probe(world, local, timeout=240.0):
    """exchange='ipc' writes into peers' memory through hipIpc mappings: a fault there kills the process and a torchrun job
    with it.  So before the bench's own ranks touch that path, every rank runs it once in a CHILD process (this script with
    --probe-ipc: gloo rendezvous on a neighbouring port, the rank's own GPU, one sharded step through both exchanges, poses
    compared across ranks bit for bit).  A child that crashes, hangs (killed after `timeout`) or disagrees returns non-zero and
    the bench stays on the RCCL all-reduce — the first multi-GPU run still prints its line."""
    import subprocess
    env = dict(os.environ, BT_BENCH_BACKEND="gloo", BT_PROBE_DEVICE=str(local),
               MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 23))
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
        del env[k]
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe-ipc", "--gpus", str(world)], env=env, timeout=timeout,
                           capture_output=True, text=True)
    except subprocess.TimeoutExpired:
        return False, "probe timed out"
    return r.returncode == 0, (r.stderr or r.stdout)[-400:]


def probe_main(args):
    import torch
    import torch.distributed as dist
    from batrack_amd import graphgen
    from batrack_amd.parallel import ShardedBA
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dev = torch.device("cuda", int(os.environ.get("BT_PROBE_DEVICE", "0")) % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    g = graphgen.make_config("C3", seed=args.seed)
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights_pose)
    ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
    scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
    ok = True
    ea = ShardedBA(ii, jj, kk, poses.shape[0], patches.shape[0], 1, dev, exchange="rccl")      # (host-staged sum under gloo)
    eb = ShardedBA(ii, jj, kk, poses.shape[0], patches.shape[0], 1, dev, exchange="ipc")
    Pa, Xa, Pb, Xb = torch.empty_like(poses), torch.empty_like(patches), torch.empty_like(poses), torch.empty_like(patches)
    ea.step(poses, patches, mono, intr, t3, 3, w, Pa, Xa, *scal, False)
    for _ in range(3):                                                                       # both parities of the exchange buffer
        eb.step(poses, patches, mono, intr, t3, 3, w, Pb, Xb, *scal, False)
    torch.cuda.synchronize()
    ok = ok and eb.exchange_status() == 0 and eb.stepper.status() == 0 and bool(torch.isfinite(Pb).all())
    ok = ok and float((Pa - Pb).abs().max()) <= 5e-7
    allp = [torch.empty_like(Pb, device="cpu") for _ in range(world)]
    dist.all_gather(allp, Pb.cpu())
    ok = ok and all(torch.equal(allp[0], q) for q in allp)           # rank-ordered sums: the same bits on every rank
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    eb.close()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


def main():
    args = parse()
    if args.probe_ipc:
        probe_main(args)
        return
    import torch
    import torch.distributed as dist
    from batrack_amd import graphgen
    from batrack_amd.plan import Plan, Stepper
    from batrack_amd.parallel import ShardedBA

    from batrack_amd.hostenv import limit_host_threads
    limit_host_threads()          # a CPU pool wider than the cgroup quota freezes the launching thread (hostenv.py)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback)")
    # BT_BENCH_BACKEND=gloo: test hook to run several ranks on ONE GPU (RCCL refuses duplicate devices)
    backend = os.environ.get("BT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    g = graphgen.make_config(args.workload, seed=args.seed)
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    poses, patches, mono, intr = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics)
    t3, w_pose, w_all = f32(g.targets3), f32(g.weights_pose), f32(g.weights)
    ii, jj, kk = (torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk))
    n_buf, p_tot, fixedp = poses.shape[0], patches.shape[0], 1
    scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")        # bounds, lmbda, ep, alpha, loss (batrack.py:861-875)

    t0 = time.perf_counter()
    exchange = None
    engines = {}
    xnote = None
    if world > 1:
        # The exchange of the packed [S | y].  "rccl": dist.all_reduce on the compute stream.  "ipc": the one-shot peer-write
        # path over hipIpc-mapped buffers (BT_BENCH_EXCHANGE=ipc, the default) — probed in child processes first
        # (ipc_probe), then checked here on one step: finite, no time-out, the same bits on every rank (the slots are summed
        # in rank order) and within float32 rounding of the RCCL result (whose summation order is RCCL's own).  Dropped for
        # RCCL if anything fails — nothing is assumed about a topology this code has not run on.
        engines["rccl"] = ShardedBA(ii, jj, kk, n_buf, p_tot, fixedp, dev, exchange="rccl")
        exchange = "rccl"
        if os.environ.get("BT_BENCH_EXCHANGE", "ipc") == "ipc":
            ok, why = (True, "") if os.environ.get("BT_BENCH_IPC_PROBE", "1") == "0" else ipc_probe(world, local)
            flag = torch.tensor([1 if ok else 0], device=dev if backend == "nccl" else "cpu", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                xnote = "exchange='ipc' failed its child-process probe on some rank; RCCL all-reduce only" + (f" (rank {rank}: {why.strip()[-200:]})" if not ok else "")
                if not ok:
                    print(f"bench.py rank {rank}: {xnote}", file=sys.stderr, flush=True)
            else:
                eng2, ok = None, 1
                try:
                    eng2 = ShardedBA(ii, jj, kk, n_buf, p_tot, fixedp, dev, exchange="ipc")
                    Pa, Xa, Pb, Xb = (torch.empty_like(poses), torch.empty_like(patches), torch.empty_like(poses), torch.empty_like(patches))
                    engines["rccl"].step(poses, patches, mono, intr, t3, t3.stride(0), w_pose, Pa, Xa, *scal, False)
                    eng2.step(poses, patches, mono, intr, t3, t3.stride(0), w_pose, Pb, Xb, *scal, False)
                    torch.cuda.synchronize()
                    if eng2.exchange_status() != 0 or not bool(torch.isfinite(Pb).all()) or float((Pa - Pb).abs().max()) > 5e-7:
                        ok = 0
                    same = [torch.empty_like(Pb) for _ in range(world)] if backend == "nccl" else [torch.empty_like(Pb, device="cpu") for _ in range(world)]
                    dist.all_gather(same, Pb if backend == "nccl" else Pb.cpu())
                    if not all(torch.equal(same[0], q) for q in same):
                        ok = 0                                                   # the ranks' poses must agree bit for bit
                except Exception as e:                                       # noqa: BLE001  (reported, not swallowed)
                    print(f"bench.py rank {rank}: exchange='ipc' unavailable ({e}); using the RCCL all-reduce", file=sys.stderr, flush=True)
                    ok = 0
                flag = torch.tensor([ok], device=dev if backend == "nccl" else "cpu", dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    engines["ipc"], exchange = eng2, "ipc"
                else:
                    xnote = "exchange='ipc' disagreed with the RCCL all-reduce, timed out or could not be set up; RCCL all-reduce only"
        eng = engines[exchange]
        plan, stepper = eng.plan, eng.stepper
        tg, wp_l, wa_l = t3, w_pose, w_all
        step = eng.step
    else:
        plan = Plan(ii, jj, kk, n_buf, p_tot, fixedp)
        stepper = Stepper(plan, dev)
        tg, wp_l, wa_l = t3, w_pose, w_all
        step = stepper.step
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t0) * 1e3
    # steady state: the same edge list planned again (library loaded, pools warm) — what a new edge list costs per update()
    plan_ms_steady = None
    if world == 1:
        t1 = time.perf_counter()
        plan2 = Plan(ii, jj, kk, n_buf, p_tot, fixedp)
        torch.cuda.synchronize()
        plan_ms_steady = (time.perf_counter() - t1) * 1e3
        del plan2

    P = [poses.clone(), torch.empty_like(poses)]
    X = [patches.clone(), torch.empty_like(patches)]

    def ba_iter(k):
        a, b = k & 1, (k + 1) & 1
        step(P[a], X[a], mono, intr, tg, tg.stride(0), wp_l, P[b], X[b], *scal, False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn, warmup, steps):
        """EXACTLY `steps` calls of fn(k) between fences (barrier + synchronize on both sides), MAX over ranks."""
        for k in range(warmup):
            fn(k)
        fence()
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            fn(k)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return steps, el

    steps, elapsed = timed(ba_iter, args.warmup, args.steps)
    # the spread of that number: the same K steps five more times (each block between its own fences; `value` stays the first)
    blocks = [1e3 * el / st_ for st_, el in (timed(ba_iter, 0, args.steps) for _ in range(5))]
    xrates = None
    if world > 1:
        # both exchanges for the record (the headline `value` is the one named in config.parallelism)
        xrates = {exchange: round(steps / elapsed, 2)}
        for name, e2 in engines.items():
            if name != exchange:
                def other(k, e2=e2):
                    a, b = k & 1, (k + 1) & 1
                    e2.step(P[a], X[a], mono, intr, tg, tg.stride(0), wp_l, P[b], X[b], *scal, False)
                s2, el2 = timed(other, min(args.warmup, 5), args.steps)
                xrates[name] = round(s2 / el2, 2)
        eng.check_exchange()
    status = stepper.status()

    extra = {}
    roofline = None
    cpu_baseline = None
    if world == 1:
        # dual iteration (what BATRACK.update really runs): pose+structure then structure-only
        P2 = [poses.clone(), torch.empty_like(poses)]
        X2 = [patches.clone(), torch.empty_like(patches), torch.empty_like(patches)]
        nd = max(args.steps // 2, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nd):
            a, b = k & 1, (k + 1) & 1
            stepper.step(P2[a], X2[a], mono, intr, tg, tg.stride(0), wp_l, P2[b], X2[2], *scal, False)
            stepper.step(P2[b], X2[2], mono, intr, tg, tg.stride(0), wa_l, P2[b], X2[b], *scal, True)
        torch.cuda.synchronize()
        extra["dual_iterations_per_s"] = nd / (time.perf_counter() - t0)

        # full BA to convergence from the perturbed start (BASELINE.json configs[2]): dual iterations until the pose
        # update of a pose+structure step is below 1e-6 (max |dX|), the norm read back every iteration
        P3 = [poses.clone(), torch.empty_like(poses)]
        X3 = [patches.clone(), torch.empty_like(patches), torch.empty_like(patches)]
        float(stepper.dx.abs().max())                     # (first use of the reduction: code-object load, not BA time)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        conv_it, dxmax = 0, float("inf")
        while conv_it < 200 and dxmax > 1e-6:
            a, b = conv_it & 1, (conv_it + 1) & 1
            stepper.step(P3[a], X3[a], mono, intr, tg, tg.stride(0), wp_l, P3[b], X3[2], *scal, False)
            dxmax = float(stepper.dx.abs().max())
            stepper.step(P3[b], X3[2], mono, intr, tg, tg.stride(0), wa_l, P3[b], X3[b], *scal, True)
            conv_it += 1
        torch.cuda.synchronize()
        extra["full_ba_to_convergence"] = {"dual_iterations": conv_it, "ms": round((time.perf_counter() - t0) * 1e3, 3),
                                           "last_max_abs_dX": dxmax, "stop": "max|dX| < 1e-6"}

        # drop-in Python entry point (allocation + plan-cache lookup per call included)
        from batrack_amd.backend.ba import BA_rgbd_droid
        from batrack_amd.backend.lietorch import SE3
        Gs, pat = SE3(poses[None]), patches[None, :, :, None, None]
        t3b = t3[None]
        na = max(args.steps // 2, 10)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(na):
                Gs, pat = BA_rgbd_droid(Gs, pat, mono[None, :, None], intr[None], t3b[..., :2], t3b[..., 2:],
                                        w_pose[None], 1e-4, ii, jj, kk, list(g.bounds), ep=10, fixedp=1,
                                        structure_only=False, loss="huber", alpha=0.05)
            torch.cuda.synchronize()
            extra["python_api_iterations_per_s"] = na / (time.perf_counter() - t0)

        # the real-shape sliding window (what the pipeline runs every frame: 138k edges with repeats, 15 free poses; the
        # headline graph above is BASELINE.json's 64-keyframe one): step time and the per-kernel durations, for the record
        try:
            gw, fpw = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
            Wp, Wx, Wm, Wi = f32(gw.poses), f32(gw.patches), f32(gw.mono_disp), f32(gw.intrinsics)
            Wt, Ww = f32(gw.targets3), f32(gw.weights_pose)
            wplan = Plan(*(torch.as_tensor(a_, device=dev) for a_ in (gw.ii, gw.jj, gw.kk)), Wp.shape[0], Wx.shape[0], fpw)
            wst = Stepper(wplan, dev)
            Wo, Wxo = torch.empty_like(Wp), torch.empty_like(Wx)
            wscal = (list(gw.bounds), 1e-4, 10.0, 0.05, "huber")
            for _ in range(5):
                wst.step(Wp, Wx, Wm, Wi, Wt, 3, Ww, Wo, Wxo, *wscal, False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                wst.step(Wp, Wx, Wm, Wi, Wt, 3, Ww, Wo, Wxo, *wscal, False)
            torch.cuda.synchronize()
            wus = (time.perf_counter() - t0) / 200 * 1e6
            wacc = {}
            for _ in range(30):
                for name, v in wst.step_timed(Wp, Wx, Wm, Wi, Wt, 3, Ww, Wo, Wxo, *wscal, False).items():
                    wacc.setdefault(name, []).append(v)
            extra["sliding_window"] = {"edges": int(wplan.E), "free_poses": int(wplan.n), "tiles": int(wplan.tiles), "jacobian_kernel": wplan.jacobian_kernel,
                                       "planned_on_device": bool(wplan.built_on_device), "step_us": round(wus, 1),
                                       "kernel_us": {k: round(1e3 * float(np.mean(v)), 2) for k, v in wacc.items() if np.mean(v) > 0}}
            wst_keep = wst
        except Exception as e:                                   # (a record beside the headline number: never its failure)
            extra["sliding_window"] = {"error": repr(e)}
        try:
            # the caller loop itself (batrack.py:856-895 as batrack_amd/sequence.py replays it on synthetic observations: a new edge
            # list every frame, 2 x ITER BA calls on it): update() as the caller times it — synchronise, the calls, synchronise —,
            # median of the steady state.  The plans come from BA_rgbd_droid's own cache (clones of the previous window, made ahead).
            if "error" not in extra["sliding_window"]:
                from batrack_amd.backend import ba as _hip_ba
                from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA
                _hip_ba.clear_plan_cache()
                _obs = SyntheticObservations(n_frames=160, M=256, seed=0)
                _trk = WindowedBA(_obs, _hip_ba.BA_rgbd_droid, SlamConfig(PATCHES_PER_FRAME=256, BUFFER_SIZE=1024), device=dev)
                _times, _upd = [], _trk.update

                def _timed_update():
                    t_before = _trk.stats["ba_seconds"]
                    _upd()
                    _times.append(_trk.stats["ba_seconds"] - t_before)
                _trk.update = _timed_update
                _trk.run()
                _hip_ba.clear_plan_cache()
                _steady = np.array(_times[-100:]) * 1e3
                extra["sliding_window"]["update_ms"] = {
                    "median": round(float(np.median(_steady)), 4), "p10": round(float(np.percentile(_steady, 10)), 4),
                    "p90": round(float(np.percentile(_steady, 90)), 4), "ba_calls_per_update": 8,
                    "what": "replay of 160 frames (256 tracks per frame, window of 15 free poses, ~140k edges), last 100 update()s: synchronise, "
                            "4 x (pose+structure, structure-only) BA_rgbd_droid calls on the frame's new edge list, synchronise"}
        except Exception as e:  # noqa: BLE001
            extra["sliding_window"]["update_ms"] = repr(e)

        # per-kernel durations from HIP events recorded by the launches themselves
        acc = {}
        nt = min(args.steps, 100)
        for k in range(nt):
            a, b = k & 1, (k + 1) & 1
            ms = stepper.step_timed(P[a], X[a], mono, intr, tg, tg.stride(0), wp_l, P[b], X[b], *scal, False)
            for name, v in ms.items():
                acc.setdefault(name, []).append(v)
        kern_us = {k: 1e3 * float(np.mean(v)) for k, v in acc.items()}
        extra["kernel_us"] = {k: round(v, 3) for k, v in kern_us.items() if v > 0}
        alg_bytes = 40 * plan.E + 20 * plan.m + 72 * plan.n_all        # SURVEY.md §8d, Jacobian kernel only
        tile_s = kern_us["tile"] * 1e-6
        achieved = alg_bytes / tile_s / 1e9 if tile_s > 0 else 0.0
        # HBM bytes per launch from the PMC passes (their own rocprofv3 runs of tools/gpu_profile_round.sh, committed under
        # profiles/): quoted only if the file was made from the kernel sources this run was built from
        traffic, traffic_source = None, None
        try:
            from batrack_amd import _lib as _bl
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_k_tile.json")))
            if pmc.get("kernel_sources_sha16") != _bl.kernel_sources_sha16():
                traffic_source = ("profiles/pmc_k_tile.json was measured on other kernel sources (sha16 "
                                  f"{pmc.get('kernel_sources_sha16')} != {_bl.kernel_sources_sha16()}): not quoted")
            elif pmc.get(args.workload, {}).get("edges") == plan.E:
                traffic = pmc[args.workload]["traffic_bytes"]
                traffic_source = ("profiles/pmc_k_tile.json: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this workload on "
                                  "these kernel sources, not measured in this run")
        except Exception as e:
            traffic, traffic_source = None, f"profiles/pmc_k_tile.json unreadable ({e!r})"
        # ... and COLD: the graph's 5.6 MB live in L2 / Infinity Cache between the steps of the loop above; a 512 MB sweep
        # between launches evicts them (L2 32 MB, Infinity Cache 256 MB), so this is the kernel reading its inputs from HBM
        flush_buf = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

        def cold_tile_us(stp, call, reps=12):
            ts = []
            for _ in range(reps):
                flush_buf.add_(1)
                ts.append(1e3 * call(stp)["tile"])
            return float(np.median(ts))
        cold_us = cold_tile_us(stepper, lambda stp: stp.step_timed(P[0], X[0], mono, intr, tg, tg.stride(0), wp_l, P[1], X[1], *scal, False))
        roofline = {"bound": "hbm", "kernel": plan.jacobian_kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes": alg_bytes, "kernel_us": round(kern_us["tile"], 3),
                    "cold_kernel_us": round(cold_us, 3), "cold_frac": round(alg_bytes / (cold_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                    "cold_how": "median of 12 launches, each behind a 512 MB read-modify-write sweep (L2 + Infinity Cache evicted)"}
        if "sliding_window" in extra and "kernel_us" in extra["sliding_window"]:
            try:
                extra["sliding_window"]["cold_kernel_us_tile"] = round(cold_tile_us(None, lambda _: wst_keep.step_timed(Wp, Wx, Wm, Wi, Wt, 3, Ww, Wo, Wxo, *wscal, False)), 2)
            except Exception as e:                               # (a record beside the headline number)
                extra["sliding_window"]["cold_kernel_us_tile"] = repr(e)

        if not args.no_cpu_baseline:
            import oracle
            from oracle import refseq
            from batrack_amd.hostenv import cpu_quota
            tc = lambda a: torch.as_tensor(np.asarray(a, np.float32))
            rin = (tc(g.poses), tc(g.patches), tc(g.mono_disp), tc(g.intrinsics), tc(g.targets3), tc(g.weights_pose),
                   torch.as_tensor(g.ii), torch.as_tensor(g.jj), torch.as_tensor(g.kk), list(g.bounds))
            quota = cpu_quota()
            runs = {}
            for nthr in sorted({min(8, quota), quota}):
                torch.set_num_threads(nthr)
                for _ in range(3):
                    refseq.ba_step(*rin, fixedp=1)
                ts = []
                t_end = time.perf_counter() + args.cpu_seconds
                while len(ts) < 10 or (time.perf_counter() < t_end and len(ts) < 200):
                    t0 = time.perf_counter()
                    refseq.ba_step(*rin, fixedp=1)
                    ts.append(time.perf_counter() - t0)
                runs[nthr] = (float(np.median(ts)), len(ts))
            limit_host_threads()
            best = min(runs, key=lambda k: runs[k][0])
            f64 = lambda a: np.asarray(a, np.float32).astype(np.float64)
            cin = (f64(g.poses), f64(g.patches), f64(g.mono_disp), f64(g.intrinsics), f64(g.targets3),
                   f64(g.weights_pose), g.ii, g.jj, g.kk, g.bounds)
            oracle.ba_step(*cin, fixedp=1, dtype=np.float32)
            t0 = time.perf_counter()
            nc = 0
            while time.perf_counter() - t0 < args.cpu_seconds:
                oracle.ba_step(*cin, fixedp=1, dtype=np.float32)
                nc += 1
            c_rate = nc / (time.perf_counter() - t0)
            cpu_baseline = {"value": round(1.0 / runs[best][0], 3), "unit": "BA iterations/s", "cores": best, "kind": "port", "port_of": "refseq: the reference's operator sequence (ba.py:253-337) on torch-CPU, oracle/refseq.py",
                            "host_cores": os.cpu_count(), "cpu_quota": quota,
                            "by_threads": {str(k): {"iterations_per_s": round(1.0 / v[0], 3), "median_ms": round(1e3 * v[0], 2), "calls": v[1]}
                                           for k, v in runs.items()},
                            "c_port_1core": {"iterations_per_s": round(c_rate, 3), "calls": nc,
                                             "what": "scalar float32 C port of the algorithm (oracle/ba_oracle_impl.h), edge-major, no block materialisation"},
                            "sample": f"median of {runs[best][1]} pose+structure steps of the same {args.workload} graph, float32, torch {torch.__version__} CPU ops in the "
                                      "reference's operator sequence (oracle/refseq.py: block materialisation, 12 scatter-adds, dense E, GEMM Schur, "
                                      "cholesky_ex), 3 warm-up calls"}

    # A sharded workload whose edge work dominates (the headline graph's step is mostly the replicated 378 x 378 solve, which
    # no rank count shortens): 64 keyframes x 4096 tracks per frame x 8 observations = 2.1M edges, the same generator.
    large = None
    if not args.no_large:
        from batrack_amd.plan import wave_per_tile_kernels
        gl = graphgen.make_graph(64, 4096, 8, seed=args.seed)
        Lp, Lx, Lm, Li, Lt, Lw = (f32(a_) for a_ in (gl.poses, gl.patches, gl.mono_disp, gl.intrinsics, gl.targets3, gl.weights_pose))
        lidx = [torch.as_tensor(a_, device=dev) for a_ in (gl.ii, gl.jj, gl.kk)]
        lscal = (list(gl.bounds), 1e-4, 10.0, 0.05, "huber")

        def measure_large(wpt):
            """The default (from 2048 tiles the wave-per-tile kernels, mixed precision: inside the 1e-5 bar) and, beside it, the
            float64 tile kernels at every size (bt_config_wave_per_tile_kernels(0))."""
            prev = wave_per_tile_kernels(wpt)
            try:
                if world > 1:
                    leng = ShardedBA(*lidx, Lp.shape[0], Lx.shape[0], 1, dev, exchange=exchange)
                    lstep, lplan = leng.step, leng.plan
                else:
                    leng = None
                    lplan = Plan(*lidx, Lp.shape[0], Lx.shape[0], 1)
                    lstep = Stepper(lplan, dev).step
            finally:
                wave_per_tile_kernels(prev)
            LP, LX = [Lp.clone(), torch.empty_like(Lp)], [Lx.clone(), torch.empty_like(Lx)]

            def large_iter(k):
                a, b = k & 1, (k + 1) & 1
                lstep(LP[a], LX[a], Lm, Li, Lt, 3, Lw, LP[b], LX[b], *lscal, False)
            ls, lel = timed(large_iter, 5, 50)
            r = {"iterations_per_s": round(ls / lel, 2), "ms_per_step": round(1e3 * lel / ls, 4), "steps": ls,
                 "iterations_per_s_blocks": [round(st_ / el, 2) for st_, el in (timed(large_iter, 0, 50) for _ in range(4))],
                 "edges_this_rank": int(lplan.E), "jacobian_kernel_this_rank": lplan.jacobian_kernel,
                 "edge_precision_this_rank": {8: "float64 per edge", 6: "mixed: float64 reprojection and residual, float32 Jacobians", 4: "float32 per edge"}[lplan.edge_precision],
                 "planned_on_device": bool(lplan.built_on_device)}
            if leng is not None:
                leng.check_exchange()
                leng.close()
            return r
        try:
            large = {"workload": f"64 keyframes, {len(gl.ii)} edges, {len(np.unique(gl.kk))} tracks, 63 free poses (make_graph(64, 4096, 8), seed {args.seed})"}
            large.update(measure_large(True))
            large["float64_tile_kernels_only"] = measure_large(False)
        except Exception as e:                                   # (a record beside the headline number: never its failure)
            large = {"error": repr(e)}
            if world > 1:
                raise                                            # (but under N > 1 a rank that dropped out would leave the others waiting)
        del Lp, Lx, Lt, Lw

    # The roofline of the Jacobian kernel where it can approach it: 8.4M edges (64 keyframes x 16384 tracks per frame x 8
    # observations, the same generator), the kernel's own duration from HIP events, warm and behind the flush
    if world == 1 and not args.no_large and roofline is not None:
        try:
            gb = graphgen.make_graph(64, 16384, 8, seed=args.seed)
            Bp, Bx, Bm, Bi, Bt, Bw = (f32(a_) for a_ in (gb.poses, gb.patches, gb.mono_disp, gb.intrinsics, gb.targets3, gb.weights_pose))
            bplan = Plan(*(torch.as_tensor(a_, device=dev) for a_ in (gb.ii, gb.jj, gb.kk)), Bp.shape[0], Bx.shape[0], 1)
            bst = Stepper(bplan, dev)
            Bo, Bxo = torch.empty_like(Bp), torch.empty_like(Bx)
            bscal = (list(gb.bounds), 1e-4, 10.0, 0.05, "huber")
            bcall = lambda _=None: bst.step_timed(Bp, Bx, Bm, Bi, Bt, 3, Bw, Bo, Bxo, *bscal, False)
            for _ in range(3):
                bcall()
            runs = [bcall() for _ in range(12)]
            warm = float(np.median([1e3 * r_["tile"] for r_ in runs]))
            # every kernel of that step ("depth": the back-substitution's pass over the edges, k_edge2u)
            step_kernels = {k_: round(float(np.median([1e3 * r_[k_] for r_ in runs])), 2) for k_ in runs[0] if np.median([r_[k_] for r_ in runs]) > 0}
            cold = cold_tile_us(None, bcall)
            balg = 40 * bplan.E + 20 * bplan.m + 72 * bplan.n_all
            roofline["large"] = {"workload": f"64 keyframes, {bplan.E} edges, {bplan.m} tracks (make_graph(64, 16384, 8), seed {args.seed})",
                                 "kernel": bplan.jacobian_kernel, "edge_precision": bplan.edge_precision, "algorithmic_bytes": balg,
                                 "kernel_us": round(warm, 2), "achieved": round(balg / (warm * 1e-6) / 1e9, 1), "frac": round(balg / (warm * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                 "cold_kernel_us": round(cold, 2), "cold_frac": round(balg / (cold * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                 "step_kernel_us": step_kernels}
            del bst, bplan, Bp, Bx, Bt, Bw
        except Exception as e:                                   # (a record beside the headline number: never its failure)
            roofline["large"] = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "BA iterations/s on 64-KF/128k-edge graph",
            "value": round(steps / elapsed, 2), "unit": "BA iterations/s",
            "n_gpus": world, "steps": steps, "steps_requested": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 5),
            "ms_per_step_blocks": {"n": len(blocks), "min": round(min(blocks), 5), "median": round(float(np.median(blocks)), 5), "max": round(max(blocks), 5),
                                   "what": "the same K steps timed five more times behind the headline block"},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {g.n_frames} keyframes, {plan.E if world == 1 else len(g.ii)} edges, "
                                   f"{len(np.unique(g.kk))} tracks, {plan.n} free poses, pose+structure GN step, huber, "
                                   f"make_graph seed {args.seed}",
                       "parallelism": "single GPU" if world == 1 else (f"track-sharded x{world}, 1 exchange of the non-zero blocks of [S|y] per step: " +
                                                                         ("one-shot peer writes over hipIpc-mapped buffers" if exchange == "ipc" else "RCCL all-reduce")),
                       "plan_build_ms": round(plan_ms_steady if plan_ms_steady is not None else plan_ms, 2),
                       "plan_build_ms_cold": round(plan_ms, 2), "edge_precision": "float64 per edge" if plan.edge_precision == 8 else "float32 per edge",
                       "solver_status": status,
                       # what the process group itself reports (WORLD_SIZE is what the launcher said)
                       "ranks_reported_by_process_group": (dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1),
                       **extra},
        }
        if large is not None:
            out["config"]["sharded_large" if world > 1 else "large_graph"] = large
            if "iterations_per_s" in large:
                # the edge work that shards, beside the headline (whose step is mostly the replicated solve): first-class
                out["value_large"] = {"value": large["iterations_per_s"], "unit": "BA iterations/s", "workload": large["workload"],
                                      "jacobian_kernel_this_rank": large.get("jacobian_kernel_this_rank")}
        if xrates is not None:
            out["config"]["exchange"] = exchange
            out["config"]["exchange_iterations_per_s"] = xrates
            out["config"]["exchange_note"] = xnote
            out["config"]["what_this_line_shows"] = ("strong scaling of ONE fixed 131k-edge graph: the 378 x 378 solve is replicated on every rank and is most of the "
                                                     "step, so `value` is not expected to grow with N; the edge work that shards is in config.sharded_large")
        if roofline is None:
            # N > 1: the Jacobian kernel of rank 0's shard (its own plan: this rank's tracks), same definition
            acc = []
            for k in range(min(args.steps, 50)):
                a, b = k & 1, (k + 1) & 1
                acc.append(stepper.step_timed(P[a], X[a], mono, intr, tg, tg.stride(0), wp_l, P[b], X[b], *scal, False)["tile"])
            tile_us = 1e3 * float(np.mean(acc))
            alg_bytes = 40 * plan.E + 20 * plan.m + 72 * plan.n_all
            achieved = alg_bytes / (tile_us * 1e-6) / 1e9 if tile_us > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": "k_tile", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                        "algorithmic_bytes": alg_bytes, "kernel_us": round(tile_us, 3), "scope": f"rank 0's shard of {world}"}
        out["roofline"] = roofline
        if cpu_baseline is not None:
            out["cpu_baseline"] = cpu_baseline
        print(json.dumps(out), flush=True)
    if world > 1:
        for e in engines.values():
            e.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
