"""-m gpu: the track-sharded step end to end on the device, through BOTH exchange paths.  Two or four ranks share
cuda:0 (the GPU box has one GPU):
  exchange="rccl"  bt_ba_reduce_pack -> all-reduce -> bt_ba_unpack_solve_update; the all-reduce is staged through gloo for
                   this test only (RCCL needs one GPU per rank)
  exchange="ipc"   bt_ba_reduce_push -> bt_ba_pull_solve_update: the production one-shot exchange as it is — hipIpc-mapped
                   buffers, peer writes and flags on the compute stream — with the peers' buffers living on the same GPU.
Everything else (per-shard plans with the global n_all, the kernels, the gather of the disparities) is the production path.
The result is compared with the float64 ORACLE (two chained steps) and with the 1-GPU HIP step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _inputs(seed=3, shape="band"):
    from batrack_amd import graphgen
    if shape == "few":          # three tracks only: with four ranks at least one owns nothing and still reduces, solves, gathers
        g = graphgen.make_graph(6, 1, 4, seed=seed)
        keep = np.isin(g.kk, [1, 3, 4])
        import dataclasses
        g = dataclasses.replace(g, ii=g.ii[keep], jj=g.jj[keep], kk=g.kk[keep], targets3=g.targets3[keep], weights=g.weights[keep],
                       weights_pose=np.ones_like(g.weights_pose[keep]))
    elif shape == "shibuya":      # BASELINE.json configs[3] stand-in: Shibuya camera, sliding-window edge list
        g, _ = graphgen.make_window_graph(n_frames=24, M=64, seed=seed, cam=graphgen.SHIBUYA)
    elif shape == "shibuya_full":  # the same at FULL size: 50 frames, M = 256 (shibuya.yaml:10), 640 x 352 (calibs/tartan_shibuya.txt:1 after the crop)
        g, _ = graphgen.make_window_graph(n_frames=50, M=256, seed=seed, cam=graphgen.SHIBUYA_CROP)
    else:
        g = graphgen.make_graph(16, 64, 8, seed=seed)
    f = lambda a: np.asarray(a, np.float32)
    return g, dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intr=f(g.intrinsics),
                   t3=f(g.targets3), w=f(g.weights_pose), ii=g.ii, jj=g.jj, kk=g.kk)


def _lmbda_vector(kk):
    """One damping value per distinct track of the full edge list, ascending patch order (ba.py:299-300)."""
    m = len(np.unique(kk))
    return np.random.default_rng(11).uniform(1e-4, 0.5, m).astype(np.float32)


def _worker(rank, world, port, out, shape, fixedp, exchange="rccl", per_track_lmbda=False, absent=-1):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from batrack_amd.parallel import ShardedBA
        dev = torch.device("cuda:0")
        g, d = _inputs(shape=shape)
        T = lambda a: torch.as_tensor(a, device=dev)
        poses, patches, mono, intr, t3, w = (T(d[k]) for k in ("poses", "patches", "mono", "intr", "t3", "w"))
        ii, jj, kk = T(d["ii"]), T(d["jj"]), T(d["kk"])
        eng = ShardedBA(ii, jj, kk, poses.shape[0], patches.shape[0], fixedp, dev, exchange=exchange)
        if absent >= 0:
            # a peer that never arrives: rank `absent` builds its engine (the set-up collectives) and then does not step
            res = None
            if rank != absent:
                Pn, Xn = torch.empty_like(poses), torch.empty_like(patches)
                eng.step(poses, patches, mono, intr, t3, t3.stride(0), w, Pn, Xn, list(g.bounds), 1e-4, 10.0, 0.05, "huber", False)
                torch.cuda.synchronize()
                raised = False
                try:
                    eng.check_exchange()
                except RuntimeError as e:
                    raised = "BT_XCHG_TIMEOUT" in str(e)
                res = (Pn.cpu().numpy(), Xn.cpu().numpy(), eng.stepper.status(), eng.exchange_status(), raised)
            out[rank] = res
            eng.close()
            return
        tg, wl = t3, w
        scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
        P, X = [poses, torch.empty_like(poses)], [patches, torch.empty_like(patches)]
        lm = T(_lmbda_vector(d["kk"])) if per_track_lmbda else None
        for k in range(2):                                     # two chained pose+structure steps
            eng.step(P[k & 1], X[k & 1], mono, intr, tg, tg.stride(0), wl, P[(k + 1) & 1], X[(k + 1) & 1], *scal, False, lmbda_per_track=lm)
        full = eng.gather_patches(X[0])
        torch.cuda.synchronize()
        out[rank] = (P[0].cpu().numpy(), full.cpu().numpy(), int(eng.plan.E), eng.stepper.status(), eng.exchange_status())
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,fixedp,world,exchange", [(s, f, w, x) for x in ("rccl", "ipc") for (s, f, w) in
                                                         [("band", 1, 2), ("shibuya", 9, 2), ("band", 1, 4), ("few", 1, 4)]] +
                         [("band", 1, 8, "ipc"), ("shibuya_full", 35, 8, "ipc"), ("shibuya_full", 35, 8, "rccl")])    # BASELINE.json configs[3]: eight ranks
def test_sharded_step_equals_oracle_and_single_gpu(shape, fixedp, world, exchange):
    import oracle
    from batrack_amd.plan import Plan, Stepper
    g, d = _inputs(shape=shape)
    dev = "cuda:0"
    T = lambda a: torch.as_tensor(a, device=dev)
    poses, patches, mono, intr, t3, w = (T(d[k]) for k in ("poses", "patches", "mono", "intr", "t3", "w"))
    ii, jj, kk = T(d["ii"]), T(d["jj"]), T(d["kk"])
    st = Stepper(Plan(ii, jj, kk, poses.shape[0], patches.shape[0], fixedp), dev)
    scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
    P, X = [poses, torch.empty_like(poses)], [patches, torch.empty_like(patches)]
    for k in range(2):
        st.step(P[k & 1], X[k & 1], mono, intr, t3, 3, w, P[(k + 1) & 1], X[(k + 1) & 1], *scal, False)
    torch.cuda.synchronize()
    ref_pose, ref_pat = P[0].cpu().numpy(), X[0].cpu().numpy()

    port = 29700 + (os.getpid() % 1000)
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, shape, fixedp, exchange), nprocs=world, join=True)
    # the float64 oracle on the same float32 inputs, two chained pose+structure steps
    f64 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    op, ox = f64(d["poses"]), f64(d["patches"])
    for _ in range(2):
        r = oracle.ba_step(op, ox, f64(d["mono"]), f64(d["intr"]), f64(d["t3"]), f64(d["w"]), d["ii"], d["jj"], d["kk"], g.bounds, fixedp=fixedp)
        op, ox = r["poses_out"], r["patches_out"]
    assert len(out) == world and sum(out[r][2] for r in range(world)) == len(d["ii"])
    rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)
    for r in range(world):
        pose, pat, _, status, xstatus = out[r]
        assert status == 0 and xstatus == 0
        ep, ex = rel(pose, ref_pose), rel(pat, ref_pat)
        tol = 1e-5 if shape == "few" else 2e-6       # three tracks barely constrain the poses: rounding of the sums is amplified
        assert ep < tol and ex < tol, (r, ep, ex)     # summation order differs (atomics, shard split)
        # against the oracle: float64 per-edge maths, float32 state written twice
        eo, exo = rel(pose, op), rel(pat, ox)
        assert eo < 1e-6 and exo < 1e-6, (r, eo, exo)
    for r in range(1, world):
        assert np.array_equal(out[0][0], out[r][0])    # identical solve on every rank after the all-reduce
    if shape == "few":
        assert sum(1 for r in range(world) if out[r][2] == 0) >= 1     # a rank without a single edge took part


def test_per_track_lmbda_on_a_sharded_plan():
    """The reference's lmbda TENSOR (one value per distinct track, ba.py:299-300) with the tracks split over two ranks: every
    rank reads its own tracks' entries of the full array (plan field trk_off).  Equal to the 1-GPU step with the same tensor."""
    from batrack_amd.plan import Plan, Stepper
    g, d = _inputs(shape="band")
    dev = "cuda:0"
    T = lambda a: torch.as_tensor(a, device=dev)
    poses, patches, mono, intr, t3, w = (T(d[k]) for k in ("poses", "patches", "mono", "intr", "t3", "w"))
    ii, jj, kk = T(d["ii"]), T(d["jj"]), T(d["kk"])
    st = Stepper(Plan(ii, jj, kk, poses.shape[0], patches.shape[0], 1), dev)
    lm = T(_lmbda_vector(d["kk"]))
    scal = (list(g.bounds), 1e-4, 10.0, 0.05, "huber")
    P, X = [poses, torch.empty_like(poses)], [patches, torch.empty_like(patches)]
    for k in range(2):
        st.step(P[k & 1], X[k & 1], mono, intr, t3, 3, w, P[(k + 1) & 1], X[(k + 1) & 1], *scal, False, lmbda_per_track=lm)
    plain = [poses, torch.empty_like(poses)], [patches, torch.empty_like(patches)]
    st.step(poses, patches, mono, intr, t3, 3, w, plain[0][1], plain[1][1], *scal, False)
    torch.cuda.synchronize()
    ref_pose, ref_pat = P[0].cpu().numpy(), X[0].cpu().numpy()
    port = 29900 + (os.getpid() % 1000)
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out, "band", 1, "ipc", True), nprocs=2, join=True)
    rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)
    for r in range(2):
        assert out[r][3] == 0 and out[r][4] == 0
        assert rel(out[r][0], ref_pose) < 2e-6 and rel(out[r][1], ref_pat) < 2e-6, (r, rel(out[r][0], ref_pose), rel(out[r][1], ref_pat))
    assert rel(plain[1][1].cpu().numpy(), X[1].cpu().numpy()) > 1e-5          # and the tensor is not the scalar


def test_a_peer_that_never_arrives_fails_the_step_instead_of_solving_garbage():
    """exchange='ipc' with rank 1 absent from the step: rank 0's pull gives up after BT_XCHG_SPIN_LIMIT polls, the solver that
    follows sees a failed factorisation — the poses stay where they were (dX = 0, the reference's reaction to a failed
    Cholesky, ba.py:9-13), never the solution of a partial system — and ShardedBA.check_exchange raises."""
    g, d = _inputs(shape="band")
    port = 30100 + (os.getpid() % 1000)
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    old = os.environ.get("BT_XCHG_SPIN_LIMIT")
    os.environ["BT_XCHG_SPIN_LIMIT"] = "20000"          # (read once per process by the spawned workers)
    try:
        mp.spawn(_worker, args=(2, port, out, "band", 1, "ipc", False, 1), nprocs=2, join=True)
    finally:
        if old is None:
            del os.environ["BT_XCHG_SPIN_LIMIT"]
        else:
            os.environ["BT_XCHG_SPIN_LIMIT"] = old
    assert out[1] is None
    pose, pat, status, xstatus, raised = out[0]
    assert xstatus == 1 and raised                      # BT_XCHG_TIMEOUT, reported
    assert status == 1                                  # BT_SOLVE_CHOL_FAILED: pose update skipped
    assert np.abs(pose - d["poses"]).max() < 1e-6       # Exp(0) * G (the quaternion is renormalised)
    assert np.isfinite(pat).all()
