"""Host-side plan (batrack_amd/csrc/ba_plan.cpp) without a GPU: structure checks and a
float64 numpy execution of the plan (tests/plan_emulator.py) against the oracle."""
import os

import numpy as np
import pytest

import oracle
from batrack_amd import graphgen
from batrack_amd.plan import Plan
from plan_emulator import run as emulate

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def load(name):
    d = dict(np.load(os.path.join(GOLD, name + ".npz")))
    return d


def host_plan(d, fixedp):
    return Plan(d["ii"], d["jj"], d["kk"], d["poses"].shape[0], d["patches"].shape[0], fixedp, upload=False)


@pytest.mark.parametrize("name,fixedp", [("c1", 1), ("c1", 3), ("c1_rough", 1), ("c1_rough", 2),
                                         ("window_small", None)])
def test_plan_structure(name, fixedp):
    d = load(name)
    fixedp = int(d["fixedp"]) if fixedp is None else fixedp
    pl = host_plan(d, fixedp)
    A = pl.arrays()
    E = len(d["ii"])
    n_all = int(max(d["ii"].max(), d["jj"].max())) + 1
    assert pl.n_all == n_all and pl.n == n_all - fixedp and pl.E == E
    kx = np.unique(d["kk"])                                   # torch.unique(sorted=True), ba.py:276
    assert np.array_equal(A["kx"], kx) and pl.m == len(kx)
    pairs = np.unique(np.stack([d["ii"], d["jj"]], 1), axis=0)
    assert np.array_equal(np.stack([A["pair_i"], A["pair_j"]], 1), pairs)
    used = A["slot_edge"][A["slot_edge"] >= 0]
    assert len(used) == E and np.array_equal(np.sort(used), np.arange(E))   # each edge exactly once
    assert A["tile_ntrk"].sum() == pl.m and A["tile_ntrk"].max() <= 64
    assert A["tile_ncam"].max() == pl.max_tile_cams <= 64
    for t in range(pl.tiles):
        cams = A["tile_cams"][A["tile_cam0"][t]:A["tile_cam0"][t] + A["tile_ncam"][t]]
        assert np.all(np.diff(cams) > 0) and cams.min() >= 0 and cams.max() < pl.n
    assert A["col_ptr"][0] == 0 and A["col_ptr"][-1] == pl.nnz_blocks == len(A["row_idx"])
    for j in range(pl.n):
        rows = A["row_idx"][A["col_ptr"][j]:A["col_ptr"][j + 1]]
        assert rows[0] == j and np.all(np.diff(rows) > 0)


CASES = [("c1", "weights_pose", 1, False, "huber", {}), ("c1", "weights_pose", 3, False, "huber", {}),
         ("c1", "weights", 1, True, "huber", {}), ("c1", "weights_pose", 1, False, "cauchy", {}),
         ("c1_rough", "weights_pose", 1, False, "huber", {}),
         ("c1_rough", "weights_pose", 2, False, "huber", dict(alpha=0.5, ep=100.0)),
         ("c1_rough", "weights", 1, True, "huber", {}),
         ("window_small", "weights_pose", None, False, "huber", {}),
         ("c1", "weights_pose", 8, False, "huber", {})]


@pytest.mark.parametrize("name,wkey,fixedp,so,loss,kw", CASES)
def test_plan_execution_matches_oracle(name, wkey, fixedp, so, loss, kw):
    d = load(name)
    fixedp = int(d["fixedp"]) if fixedp is None else fixedp
    pl = host_plan(d, fixedp)
    em = emulate(pl, pl.arrays(), d, wkey, structure_only=so, loss=loss, **kw)
    o = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey],
                       d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=fixedp, structure_only=so,
                       loss=loss, want_system=True, **kw)
    if "S" in o:
        assert rel(np.tril(em["S_lower"]), np.tril(o["S"])) < 1e-10
        assert rel(em["y"], o["y"]) < 1e-10
        assert rel(em["dX"], o["dX"]) < 1e-7
    assert rel(em["patches_out"], o["patches_out"]) < 1e-10


def test_plan_rejects_bad_indices():
    ii = np.array([0, 1], np.int64); jj = np.array([1, 5], np.int64); kk = np.array([0, 1], np.int64)
    with pytest.raises(RuntimeError):
        Plan(ii, jj, kk, 4, 8, 1, upload=False)          # jj >= n_buf
    with pytest.raises(RuntimeError):
        Plan(ii, np.array([1, 2], np.int64), np.array([0, 9], np.int64), 4, 8, 1, upload=False)  # kk >= p_tot


def test_plan_c3_shape():
    g = graphgen.make_config("C3", seed=0)
    pl = Plan(g.ii, g.jj, g.kk, g.poses.shape[0], g.patches.shape[0], 1, upload=False)
    assert (pl.E, pl.n, pl.m) == (131072, 63, 16384)
    assert pl.tiles == 256 and pl.max_tile_cams <= 10 and pl.sorted_input == 1
    A = pl.arrays()
    # observation pattern is regular: every slot of every tile holds a single camera pair
    sp = A["slot_pair"].reshape(-1, 64)
    assert np.all(sp == sp[:, :1])
    # banded reduced system, two-ended elimination: two chains meet at a 7-camera separator, so the
    # sequential depth drops from 63 to 35 block columns with (almost) no fill beyond the band
    lv = A["lvl_ptr"]
    assert len(lv) - 1 == 35 and np.diff(lv).max() == 2
    assert pl.nnz_blocks <= sum(min(8, 63 - j) for j in range(63)) + 28
    perm = A["perm"]
    assert perm[:28].tolist() == list(range(28)) and perm[28:56].tolist() == list(range(62, 34, -1))


def test_c3_solver_schedules_solve_the_system():
    """Both level schedules (two-phase and fused) of the 63-pose plan, run by the emulator on a
    random SPD system with the plan's sparsity pattern, reproduce the dense solve."""
    from plan_emulator import sparse_chol_solve, sparse_chol_solve_fused
    g = graphgen.make_config("C3", seed=0)
    pl = Plan(g.ii, g.jj, g.kk, g.poses.shape[0], g.patches.shape[0], 1, upload=False)
    A = pl.arrays()
    n = pl.n
    rng = np.random.default_rng(5)
    cams = np.unique(np.stack([g.ii, g.jj], 1), axis=0) - 1
    pat = np.eye(n, dtype=bool)
    ix = g.ii[np.argsort(g.kk, kind="stable")]
    seen = {}
    for i, j, k in zip(g.ii, g.jj, g.kk):
        seen.setdefault(int(k), set()).update((int(i) - 1, int(j) - 1))
    for s in seen.values():
        s = [c for c in s if c >= 0]
        for a in s:
            for b in s:
                pat[a, b] = True
    S = np.zeros((6 * n, 6 * n))
    for a in range(n):
        for b in range(a + 1):
            if pat[a, b]:
                S[6*a:6*a + 6, 6*b:6*b + 6] = rng.normal(size=(6, 6))
    S = np.tril(S) + np.tril(S, -1).T
    S += np.eye(6 * n) * (np.abs(S).sum(1).max() + 1.0)
    y = rng.normal(size=6 * n)
    Sd = S.copy()
    Sd[np.diag_indices(6 * n)] += 10.0 + 1e-4 * np.diag(S)
    want = np.linalg.solve(Sd, y).reshape(n, 6)
    for fn in (sparse_chol_solve, sparse_chol_solve_fused):
        got = fn(A, np.tril(S), y, n, 10.0, 1e-4)
        assert rel(got, want) < 1e-10


def as_inputs(g):
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    return dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
                targets3=f(g.targets3), weights=f(g.weights), weights_pose=f(g.weights_pose),
                ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds, np.float64))


@pytest.mark.parametrize("seed,N,M,fixedp,far,groups", [(0, 10, 6, 1, 0.3, 1), (1, 17, 5, 2, 0.1, 1), (2, 24, 4, 1, 0.0, 1),
                                                 (3, 30, 3, 3, 0.5, 1), (4, 12, 8, 1, 1.0, 1), (5, 40, 2, 1, 0.05, 1),
                                                 (7, 33, 64, 0, 0.0, 4), (8, 25, 64, 1, 0.2, 3)])
def test_random_covisibility_graphs(seed, N, M, fixedp, far, groups):
    """Irregular sparsity (loop-closure-like edges, self edges, repeats): every plan table - tiles, pairs,
    elimination order, both level schedules with their hazard flags - executed in numpy gives the oracle's step."""
    g = graphgen.make_random_graph(N, M, seed=seed, far_frac=far, groups=groups)
    d = as_inputs(g)
    pl = host_plan(d, fixedp)
    em = emulate(pl, pl.arrays(), d, "weights_pose")
    o = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"],
                       d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=fixedp, want_system=True)
    assert rel(np.tril(em["S_lower"]), np.tril(o["S"])) < 1e-10
    assert rel(em["y"], o["y"]) < 1e-10
    assert rel(em["dX"], o["dX"]) < 1e-7
    assert rel(em["patches_out"], o["patches_out"]) < 1e-10


def _chain_edges(N, K=3):
    """One track per frame, observed from the K following frames (a banded system with N - fixedp free poses)."""
    kk = np.repeat(np.arange(N, dtype=np.int64), K)
    ii = kk.copy()
    jj = np.clip(ii + np.tile(np.arange(1, K + 1, dtype=np.int64), N), 0, N - 1)
    return ii, jj, kk


def test_plan_limits():
    """The documented limits (include/batrack_ba.h): 2048 free poses (beyond 255 the dense solver: no symbolic factorisation, every
    lower block in the packed form), 64 free cameras per track, one source frame per track.  At the limit the plan builds; one past
    it the call is refused (BT_EUNSUPPORTED), never a silent wrong answer."""
    ii, jj, kk = _chain_edges(256)
    pl = Plan(ii, jj, kk, 256, 256, 1, upload=False)
    assert pl.n == 255                                  # (block-sparse or dense by its price, see below: this band sits at the crossover)
    ii, jj, kk = _chain_edges(257)
    pl = Plan(ii, jj, kk, 257, 257, 1, upload=False)
    assert pl.n == 256 and pl.nnz_blocks == 256 * 257 // 2 and np.array_equal(pl.array("perm"), np.arange(256))
    ii, jj, kk = _chain_edges(2049)
    assert Plan(ii, jj, kk, 2049, 2049, 1, upload=False).n == 2048
    # fewer than 256 poses whose factor does not fit LDS as double: a long thin band stays block-sparse (float32 factor, refined); a graph
    # with long-range edges fills in and is priced cheaper DENSE (ba_plan.cpp, tools/gpu_solver_choice.py) — perm the identity, every block
    rng = np.random.default_rng(5)
    N, M, K = 120, 64, 8                                  # (a tile of 64 tracks per frame: the band is as wide as a track's span)
    kk = np.repeat(np.arange(N * M, dtype=np.int64), K); ii = kk // M
    jb = np.clip(ii + np.tile(np.arange(K, dtype=np.int64) - 3, N * M), 0, N - 1)
    band = Plan(ii, jb, kk, N, N * M, 1, upload=False)
    assert band.n == N - 1 and band.nnz_blocks < (N - 1) * N // 2 // 4 and band.updates > 0
    jf = np.where(rng.random(ii.size) < 0.3, rng.integers(0, N, ii.size), jb)
    filled = Plan(ii, jf, kk, N, N * M, 1, upload=False)
    assert filled.n == N - 1 and filled.nnz_blocks == (N - 1) * N // 2 and filled.updates == 0
    assert np.array_equal(filled.array("perm"), np.arange(N - 1))
    ii, jj, kk = _chain_edges(2050)
    with pytest.raises(RuntimeError, match="unsupported"):
        Plan(ii, jj, kk, 2050, 2050, 1, upload=False)
    # one track of frame 0 seen by 32 / 33 / 65 free cameras: the largest tile that keeps its E as double / a LOOSE track in no tile
    # (ba_loose.hip: a plan with a few such landmarks stays float64 per edge) / loose whatever the plan, coupling all its cameras
    for ncam in (32, 33, 65):
        N = ncam + 1
        jj = np.arange(1, N, dtype=np.int64); ii = np.zeros_like(jj); kk = np.zeros_like(jj)
        pl = Plan(ii, jj, kk, N, 4, 1, upload=False)
        if ncam == 32:
            assert pl.max_tile_cams == 32 and pl.tiles == 1 and pl.array("trk_loc")[0] == 0
        else:
            assert pl.tiles == 0 and pl.array("trk_loc")[0] == -1 and pl.nnz_blocks == ncam * (ncam + 1) // 2
    # MANY tracks that long (more than 64 of them): tiles of up to 64 cameras again (float32 per edge), 65 still loose
    for ncam in (64, 65):
        N = ncam + 1
        jj = np.tile(np.arange(1, N, dtype=np.int64), 70); kk = np.repeat(np.arange(70, dtype=np.int64), ncam); ii = np.zeros_like(jj)
        pl = Plan(ii, jj, kk, N, 128, 1, upload=False)
        loc = pl.array("trk_loc")
        if ncam == 64:
            assert pl.max_tile_cams == 64 and (loc >= 0).all()
        else:
            assert pl.tiles == 0 and (loc == -1).all()
    # a track whose edges name two source frames (the caller's invariant ii = ix[kk], batrack.py:199)
    with pytest.raises(RuntimeError, match="unsupported"):
        Plan(np.array([0, 1], np.int64), np.array([2, 3], np.int64), np.array([5, 5], np.int64), 4, 8, 1, upload=False)


def test_barrier_free_solver_schedule_orders_every_conflict():
    """k_solve_pipe synchronises its waves with three kinds of flags instead of a barrier per level; the waits it
    makes must order every pair of steps that touch the same block, y segment or scratch with a write involved."""
    from plan_emulator import check_pipe_protocol, pipe_schedule_applies
    g = graphgen.make_config("C3", seed=0)
    pl = Plan(g.ii, g.jj, g.kk, g.poses.shape[0], g.patches.shape[0], 1, upload=False)
    A = pl.arrays()
    assert pipe_schedule_applies(A)
    assert check_pipe_protocol(A) > 300
    for N, M, K, fp in ((48, 16, 8, 1), (12, 8, 4, 1), (9, 8, 8, 1)):
        gg = graphgen.make_graph(N, M, K, seed=3)
        A = Plan(gg.ii, gg.jj, gg.kk, gg.poses.shape[0], gg.patches.shape[0], fp, upload=False).arrays()
        assert pipe_schedule_applies(A)
        check_pipe_protocol(A)


def test_patch_activity_tables_and_track_records():
    """k_update's tables: bitmap + rank give every patch its track (or none), also with the reference's buffer of
    1024 x 256 patch slots of which a window uses a few thousand; one 32-byte record per track."""
    from batrack_amd import graphgen
    from batrack_amd.plan import Plan
    g, fixedp = graphgen.make_window_graph(n_frames=30, M=64, seed=2, n_buf=1024)
    n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
    assert p_tot == 1024 * 64
    for graph, fp in ((g, fixedp), (graphgen.make_random_graph(14, 40, seed=9), 1)):
        pl = Plan(graph.ii, graph.jj, graph.kk, graph.poses.shape[0], graph.patches.shape[0], fp, upload=False)
        A = pl.arrays()
        P = graph.patches.shape[0]
        bits, rank, top = A["act_bits"], A["act_rank"], A["trk_of_patch"]
        assert bits.dtype == np.uint32 and bits.shape[0] == rank.shape[0] == (P + 31) // 32
        p = np.arange(P)
        w, b = p >> 5, (p & 31).astype(np.uint32)
        on = (bits[w] >> b) & 1
        below = bits[w] & ((np.uint32(1) << b) - np.uint32(1))
        pop = np.array([bin(int(x)).count("1") for x in below])
        trk = np.where(on == 1, rank[w] + pop, -1)
        assert np.array_equal(trk, top)
        assert np.array_equal(np.flatnonzero(on), A["kx"]) and pl.m == int(on.sum())
        # the depth back-substitution runs per tile (k_update re-evaluates the edges): every track sits in the lane
        # trk_loc names, and tile_kx gives that lane its patch
        loc = A["trk_loc"]
        tile, lane = loc >> 6, loc & 63
        assert np.array_equal(A["tile_kx"].reshape(-1, 64)[tile, lane], A["kx"])
        pl.close()


@pytest.mark.parametrize("case", ["one_edge", "one_track", "two_frames", "all_fixed", "all_masked", "self_edges"])
def test_degenerate_problems_through_the_plan(case):
    """Single edge / single track / two frames / every pose fixed / every edge masked / self edges only: the host plan
    executed in float64 equals the oracle (the same problems run on the device in tests/test_gpu_edge_cases.py)."""
    from edge_problems import CASES
    make, fixedp = CASES[case]
    d = make()
    pl = host_plan(d, fixedp)
    out = emulate(pl, pl.arrays(), d, "weights_pose", structure_only=pl.n == 0)
    so = pl.n == 0
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"], d["ii"], d["jj"], d["kk"],
                         d["bounds"], fixedp=fixedp, structure_only=so, want_system=not so)
    assert rel(out["patches_out"], ref["patches_out"]) < 1e-10
    if not so:
        assert np.abs(out["dX"] - ref["dX"].reshape(-1, 6)).max() <= 1e-9 * max(np.abs(ref["dX"]).max(), 1e-30) + 1e-15


@pytest.mark.parametrize("name", ["c1", "c1_rough", "window_small"])
def test_wave_cuts_partition_the_slots_and_avoid_runs(name):
    """tile_cut8 / tile_cut16: k_tile's waves take [cut[w], cut[w+1]) — a partition of the tile's slots; a cut sits inside a run
    of repeated observations (same track, same target camera in consecutive slots) only when no run-free boundary exists
    within a chunk of the even split."""
    d = load(name)
    pl = host_plan(d, int(d["fixedp"]) if name == "window_small" else 1)
    A = pl.arrays()
    lab = A["slot_lab"].reshape(-1, 64)
    edge = A["slot_edge"].reshape(-1, 64)
    crossed_cuts = total_cuts = 0
    for t in range(pl.tiles):
        s0, ns = int(A["tile_slot0"][t]), int(A["tile_nslot"][t])
        lb = lab[s0:s0 + ns] >> 8
        act = edge[s0:s0 + ns] >= 0
        crossed = np.zeros(ns + 1, bool)
        crossed[1:ns] = ((lb[1:] == lb[:-1]) & (lb[1:] != 0xff) & act[1:] & act[:-1]).any(axis=1)
        for W, key in ((8, "tile_cut8"), (16, "tile_cut16")):
            cut = A[key].reshape(-1, W + 1)[t].astype(int)
            assert cut[0] == 0 and cut[W] == ns and (np.diff(cut) >= 0).all()
            chunk = -(-ns // W)
            for w in range(1, W):
                c = cut[w]
                total_cuts += 1
                if 0 < c < ns and crossed[c]:
                    crossed_cuts += 1
                    lo, hi = max(cut[w - 1], min(ns, w * chunk) - chunk), min(ns, min(ns, w * chunk) + chunk)
                    assert crossed[max(lo, 1):hi].all() or c == cut[w - 1], (t, W, w, c)     # nothing better was in reach
    assert total_cuts > 0


PM_SCRIPT = r"""
import sys
import numpy as np
from batrack_amd import graphgen
from batrack_amd.plan import Plan
kind = sys.argv[1]
if kind == "c1":
    g, fixedp = graphgen.make_config("C1", seed=0), 1
elif kind == "c3":
    g, fixedp = graphgen.make_config("C3", seed=0), 1
elif kind == "window":
    g, fixedp = graphgen.make_window_graph(n_frames=30, M=48, seed=2)
else:
    g, fixedp = graphgen.make_random_graph(20, 30, seed=4, far_frac=0.3, groups=2), 2
pl = Plan(g.ii, g.jj, g.kk, g.poses.shape[0], g.patches.shape[0], fixedp, upload=False)
A = pl.arrays()
rec = A["pm_rec"].reshape(-1, 4)
assert rec.shape[0] == pl.tiles and A["pm_edge"].size % 64 == 0
pm_edge = A["pm_edge"].reshape(-1, 64)
seen = np.zeros(len(g.ii), np.int64)
lp_of, lab_of = {}, {}
se, lab, slp = A["slot_edge"], A["slot_lab"], A["slot_lp"]
for i in np.flatnonzero(se >= 0):
    lp_of[int(se[i])] = int(slp[i]); lab_of[int(se[i])] = int(lab[i])
for t in range(pl.tiles):
    r0, lg, D, nit = int(rec[t, 0]), int(rec[t, 1]) & 0xff, int(rec[t, 1]) >> 8, int(rec[t, 2])
    S, G = 1 << lg, 64 >> lg
    assert S >= int(A["tile_npair"][t]) and nit == (int(A["tile_ntrk"][t]) + G - 1) // G
    mult = {}
    for it in range(nit):
        for d in range(D):
            row = pm_edge[r0 + it * D + d]
            for lane in np.flatnonzero(row >= 0):
                e = int(row[lane])
                s, track = lane & (S - 1), it * G + (lane >> lg)
                seen[e] += 1
                assert lp_of[e] == s
                assert int(A["kx"][int(A["tile_trk0"][t]) + track]) == int(g.kk[e])
                assert int(A["pm_lb"][t * 64 + s]) == lab_of[e] >> 8 and int(A["pm_la"][t * 64 + track]) == lab_of[e] & 0xff
                mult[(track, s)] = mult.get((track, s), 0) + 1
    assert D == max(max(mult.values(), default=1), 1)
assert np.all(seen == 1)
print("PM_OK", pl.tiles)
"""


@pytest.mark.parametrize("kind", ["c1", "window", "random", "c3"])
def test_pair_major_tables_cover_every_edge_once(kind):
    """The pair-major layout of k_etile (ba_plan.cpp; BT_FORCE kernel=k_etile builds it for every plan, read once per process): every edge
    of the list sits in exactly one (tile, iteration, round, lane); the lane's local pair is the edge's pair, its track the
    edge's track; pm_lb / pm_la are the local target / source cameras the track-major tables give the same edge; the number
    of rounds is the largest multiplicity of a (track, pair)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", PM_SCRIPT, kind], cwd=root, env=dict(os.environ, BT_FORCE="kernel=k_etile", PYTHONPATH=root),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_window_plans_are_tiled_for_the_pair_major_kernel():
    """A sliding-window edge list (few tiles of 64 tracks, deep edge lists) is laid out in tiles of 16 tracks with the
    pair-major tables; the 64-keyframe benchmark graph keeps its 64-track tiles."""
    g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
    pl = Plan(g.ii, g.jj, g.kk, g.poses.shape[0], g.patches.shape[0], fixedp, upload=False)
    assert int(pl.array("tile_ntrk").max()) == 16 and pl.tiles == 160 and pl.array("pm_rec").size == 4 * pl.tiles
    # the tables behind the tiles' partial sums (k_pair_finalize): groups of consecutive tiles with the same cameras (at most 32
    # tiles each) partition the tiles, and every pair's list holds each (tile, local pair) that carries the pair exactly once
    sg, cam0, ncam, cams = pl.array("sg_ptr"), pl.array("tile_cam0"), pl.array("tile_ncam"), pl.array("tile_cams")
    assert sg[0] == 0 and sg[-1] == pl.tiles and (np.diff(sg) > 0).all() and (np.diff(sg) <= 32).all()
    for a, b in zip(sg[:-1], sg[1:]):
        ref = cams[cam0[a]:cam0[a] + ncam[a]]
        assert all(ncam[t] == ncam[a] and (cams[cam0[t]:cam0[t] + ncam[t]] == ref).all() for t in range(a, b))
    assert len(sg) - 1 < pl.tiles                      # (a window: the tiles of one source frame share their cameras)
    pp, pi, tp0, tnp, tps = pl.array("pp_ptr"), pl.array("pp_idx"), pl.array("tile_pair0"), pl.array("tile_npair"), pl.array("tile_pairs")
    assert pp[0] == 0 and pp[-1] == tps.size == pi.size
    seen = set()
    for p in range(pp.size - 1):
        for e in pi[pp[p]:pp[p + 1]]:
            t, q = int(e) >> 6, int(e) & 63
            assert q < tnp[t] and tps[tp0[t] + q] == p and (t, q) not in seen
            seen.add((t, q))
    assert len(seen) == tps.size
    g = graphgen.make_config("C3", seed=0)
    pl = Plan(g.ii, g.jj, g.kk, g.poses.shape[0], g.patches.shape[0], 1, upload=False)
    assert int(pl.array("tile_ntrk").max()) == 64 and pl.tiles == 256 and pl.array("pm_rec").size == 0 and pl.array("sg_ptr").size == 0


def _many_small_tiles_graph(m=3000, K=26, n_buf=120, seed=0):
    """Tracks with random source frames and 26 observations each: every track closes its own tile (the 16-camera limit), so
    a plan first tiled for k_etile (16 tracks per tile) reaches the tile count of the wave-per-tile kernels."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n_buf - K - 4, m)
    kk = np.repeat(np.arange(m), K).astype(np.int64)
    ii = np.repeat(src, K).astype(np.int64)
    jj = (ii + np.tile(np.arange(K), m)).astype(np.int64)
    return ii, jj, kk, n_buf, m


def test_small_tile_plan_that_reaches_the_stream_tile_count_does_not_crash():
    """Round-3 advisor finding: an UPLOADED plan (no host slot arrays) tiled with 16 tracks per tile whose tile count reached
    stream_min_tiles() read the empty slot arrays (SIGSEGV at ba_plan.cpp want_stream_tables).  Without a GPU the upload
    itself fails AFTER the host analysis: an error return, not a crash; on a GPU box the plan is created."""
    ii, jj, kk, n_buf, m = _many_small_tiles_graph()
    host = Plan(ii, jj, kk, n_buf, m, 1, upload=False)
    assert host.tiles >= 2048 and host.arrays()["tile_ntrk"].max() <= 64
    try:
        up = Plan(ii, jj, kk, n_buf, m, 1, upload=True)
    except RuntimeError as e:                      # no device here: out of memory / HIP error from the upload
        assert "bt_plan_create failed" in str(e)
    else:
        assert up.tiles == host.tiles


def _ragged(g, drop=0.15, seed=11):
    import copy
    keep = np.random.default_rng(seed).random(np.asarray(g.kk).size) > drop
    h = copy.copy(g)
    for name in ("ii", "jj", "kk", "targets3", "weights", "weights_pose"):
        setattr(h, name, np.ascontiguousarray(np.asarray(getattr(g, name))[keep]))
    return h


@pytest.mark.parametrize("drop", [0.0, 0.15, 0.5])
def test_aligned_slots_make_a_ragged_graph_slot_uniform(drop):
    """Plans of >= 2048 tiles lay a tile's slots out per (camera pair, repeat) — ba_plan.cpp "ALIGNED SLOTS": with observations missing
    at random (tracks of different lengths) every tile is still slot-uniform, so the edge-major tables (it_edge, tile_sinfo) exist;
    every edge sits in exactly one slot and one lane of an iteration, under its own pair and track; a tile has ONE source frame."""
    from batrack_amd import graphgen
    g = graphgen.make_graph(64, 2048 if drop < 0.3 else 2560, 8, seed=3)
    if drop:
        g = _ragged(g, drop)
    E = g.ii.size
    pl = Plan(g.ii, g.jj, g.kk, g.poses.shape[0], g.patches.shape[0], 1, upload=False)
    A = pl.arrays()
    T = pl.tiles
    assert T >= 2048
    assert A["it_edge"].size > 0 and A["tile_sinfo"].size == T * 64, "the plan is not slot-uniform"
    se, sp = A["slot_edge"].reshape(-1, 64), A["slot_pair"].reshape(-1, 64)
    live = se >= 0
    assert np.array_equal(np.sort(se[live]), np.arange(E))                       # every edge once
    pair_i, pair_j = A["pair_i"], A["pair_j"]
    assert np.array_equal(pair_i[sp[live]], g.ii[se[live]]) and np.array_equal(pair_j[sp[live]], g.jj[se[live]])
    # slot-uniform: one pair per (tile, slot)
    slot0, nslot, trk0, ntrk = A["tile_slot0"], A["tile_nslot"], A["tile_trk0"], A["tile_ntrk"]
    kx = A["kx"]
    for t in list(range(0, T, 97)) + [T - 1]:
        rows = slice(slot0[t], slot0[t] + nslot[t])
        for s in range(slot0[t], slot0[t] + nslot[t]):
            p = sp[s][live[s]]
            assert p.size == 0 or (p == p[0]).all()
        # lanes = the tile's tracks, all of one source frame
        e = se[rows][live[rows]]
        assert np.unique(g.ii[e]).size == 1
        assert set(np.unique(g.kk[e])) <= set(kx[trk0[t]:trk0[t] + ntrk[t]])
        assert nslot[t] <= 64
    # it_edge: the same edges, lane = track_in_iteration * S + slot
    ie = A["it_edge"]
    assert np.array_equal(np.sort(ie[ie >= 0]), np.arange(E))
    rec = A["tile_rec"].reshape(-1, 8)
    for t in list(range(0, T, 131)) + [T - 1]:
        it0, lg, nit = rec[t, 6], rec[t, 7] & 0xff, rec[t, 7] >> 8
        assert nit % 2 == 0 and (1 << lg) >= nslot[t]
        blk = ie[it0 * 64:(it0 + nit) * 64].reshape(nit, 64)
        G = 64 >> lg
        for it in range(nit):
            for ln in np.nonzero(blk[it] >= 0)[0]:
                tr, s = it * G + (ln >> lg), ln & ((1 << lg) - 1)
                assert se[slot0[t] + s, tr] == blk[it, ln]
    if drop == 0.0:
        # nothing missing: the aligned layout IS the old one (a track's s-th edge in slot s) wherever no observation repeats
        full = nslot == 8
        assert full.mean() > 0.8


def test_a_mostly_fixed_trajectory_stays_inside_the_pair_table_of_a_tile():
    """With most of a long trajectory fixed, the free cameras of a track are few and 64 tracks fill a tile long before its camera budget
    — but every (source, target) pair of the tile, fixed frames included, has its geometry in the tile's table (kMaxTilePairs = 192).
    The planner closes tiles on the pairs too; such lists used to be BT_EUNSUPPORTED (found by tests/test_gpu_fuzz.py)."""
    rng = np.random.default_rng(3)
    N, M = 100, 3
    kk = np.repeat(np.arange(N * M, dtype=np.int64), 9)
    ii = kk // M
    jj = np.clip(ii + rng.integers(-6, 7, kk.size), 0, N - 1).astype(np.int64)
    for fixedp in (99, 97, 50, 1):
        pl = Plan(ii, jj, kk, N, N * M, fixedp, upload=False)
        assert pl.n == N - fixedp
        tp0, tnp = pl.array("tile_pair0"), pl.array("tile_npair")
        assert tnp.max() <= 192 and tnp.sum() >= len(set(zip(ii.tolist(), jj.tolist())))
        assert tp0[-1] + tnp[-1] == tnp.sum()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ranks_of_a_sharded_solve_agree_on_the_pattern_of_an_irregular_graph(world):
    """The ranks exchange [S | y] block by block of the factor's pattern (bt_ba_reduce_pack / _push), so the plans of all ranks —
    each laying out only its own tracks — must hold the same pattern, order and factor structure.  On a banded graph the cameras of a
    tile are the cameras of each of its tracks and anything agrees; on an irregular one a tile couples more than its tracks do, and a
    rank's own tiles used to leak into its pattern (found by tests/test_gpu_fuzz_sharded.py: the all-reduce sizes differed)."""
    from batrack_amd.parallel import partition_tracks, plan_range
    rng = np.random.default_rng(17)
    N, M = 40, 5
    kk = np.repeat(np.arange(N * M, dtype=np.int64), rng.integers(1, 9, N * M))
    ii = kk // M
    near = np.clip(ii + rng.integers(-4, 5, kk.size), 0, N - 1)
    jj = np.where(rng.random(kk.size) < 0.2, rng.integers(0, N, kk.size), near).astype(np.int64)
    ranges = partition_tracks(kk, world)
    plans = [Plan(ii, jj, kk, N, N * M, 2, upload=False, own=plan_range(r, N * M)) for r in ranges]
    assert sum(p.E for p in plans) == kk.size
    ref = plans[0]
    for p in plans[1:]:
        assert p.n == ref.n and p.nnz_blocks == ref.nnz_blocks
        for name in ("perm", "col_ptr", "row_idx"):
            assert np.array_equal(p.array(name), ref.array(name)), name
    # one plan for the whole list may couple more (its tiles' cameras pairwise), never less
    full = Plan(ii, jj, kk, N, N * M, 2, upload=False)
    assert full.nnz_blocks >= ref.nnz_blocks
