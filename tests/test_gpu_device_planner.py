"""The device-side planner (csrc/plan_device.hip) against the host's analysis of the same edge lists: the pair-major table,
the tile records and every host-side table of the two plans are IDENTICAL — for regular windows, windows with edges removed
at random, shuffled edge order (the radix sort must reproduce the host's (target frame, original index) order of a track's
repeated observations) and several keyframe strides — and lists the device path does not take come back planned by the host."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from batrack_amd import graphgen  # noqa: E402
from batrack_amd.plan import Plan  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FORCED = bool(os.environ.get("BT_FORCE"))         # (kernel / planner selection forced: the suites of test_gpu_jacobian_kernels.py)
TABLES = ("pm_edge", "pm_rec", "pm_lb", "pm_la", "kx", "pair_i", "pair_j", "tile_trk0", "tile_ntrk", "tile_ncam", "tile_cams", "tile_pair0",
          "tile_npair", "tile_pairs", "tile_flags", "tile_ij", "tile_kx", "pp_ptr", "pp_idx", "sg_ptr", "col_ptr", "row_idx", "perm", "act_bits", "act_rank")


def both_plans(ii, jj, kk, n_buf, p_tot, fixedp):
    dev = Plan(*(torch.as_tensor(a, device=DEV) for a in (ii, jj, kk)), n_buf, p_tot, fixedp)
    host = Plan(np.asarray(ii), np.asarray(jj), np.asarray(kk), n_buf, p_tot, fixedp)
    return dev, host


@pytest.mark.skipif(FORCED, reason="kernel / planner selection forced by the environment")
@pytest.mark.parametrize("variant", ["regular", "M64_stride1", "M128_stride3", "thinned", "shuffled", "thinned_shuffled"])
def test_device_planned_tables_equal_the_hosts(variant):
    kw = dict(n_frames=50, M=256, seed=4)
    if variant == "M64_stride1":
        kw = dict(n_frames=40, M=64, seed=7, kf_stride=1, window=10)
    if variant == "M128_stride3":
        kw = dict(n_frames=45, M=128, seed=9, kf_stride=3, window=14)
    g, fixedp = graphgen.make_window_graph(**kw)
    ii, jj, kk = (np.asarray(a) for a in (g.ii, g.jj, g.kk))
    rng = np.random.default_rng(11)
    if "thinned" in variant:                    # irregular: a fifth of the observations gone, some tracks short
        keep = rng.random(ii.size) > 0.2
        ii, jj, kk = ii[keep], jj[keep], kk[keep]
    if "shuffled" in variant:
        p = rng.permutation(ii.size)
        ii, jj, kk = ii[p], jj[p], kk[p]
    dev, host = both_plans(ii, jj, kk, g.poses.shape[0], g.patches.shape[0], fixedp)
    assert dev.built_on_device and not host.built_on_device
    assert dev.jacobian_kernel == host.jacobian_kernel == "k_etile"
    for f in ("E", "m", "n", "tiles", "pairs", "nnz_blocks", "workspace_bytes"):
        assert getattr(dev, f) == getattr(host, f), f
    for name in TABLES:
        a, b = dev.array(name), host.array(name)
        assert a.shape == b.shape and (a == b).all(), name


@pytest.mark.skipif(FORCED, reason="kernel / planner selection forced by the environment")
def test_lists_outside_the_device_path_are_planned_by_the_host():
    # (a) a graph laid out for the wave-per-tile kernels (2048 tiles) is planned on the device whichever kernels it is laid out
    # for; (b) a short list is not; (c) a track with two source frames is refused by both paths alike; (d) an index out of range
    # is reported, not planned
    from batrack_amd.plan import wave_per_tile_kernels
    g = graphgen.make_graph(64, 2048, 8, seed=0)
    idx = [torch.as_tensor(a, device=DEV) for a in (g.ii, g.jj, g.kk)]
    p = Plan(*idx, g.poses.shape[0], g.patches.shape[0], 1)
    assert p.built_on_device and p.jacobian_kernel == "k_edge2" and p.tiles == 2048      # (slot-uniform tiles: k_edge2; ragged ones as well since round 6: aligned slots, below)
    prev = wave_per_tile_kernels(False)
    try:
        p = Plan(*idx, g.poses.shape[0], g.patches.shape[0], 1)
    finally:
        wave_per_tile_kernels(prev)
    assert p.built_on_device and p.jacobian_kernel == "k_tile" and p.tiles == 2048
    g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
    ii, jj, kk = (np.asarray(a).copy() for a in (g.ii, g.jj, g.kk))
    p = Plan(*(torch.as_tensor(a[:2000], device=DEV) for a in (ii, jj, kk)), g.poses.shape[0], g.patches.shape[0], fixedp)
    assert not p.built_on_device
    bad = ii.copy()
    bad[5] = bad[5] + 1 if bad[5] + 1 < g.poses.shape[0] else bad[5] - 1           # a second source frame for that edge's track
    for idx in ((torch.as_tensor(bad, device=DEV), torch.as_tensor(jj, device=DEV), torch.as_tensor(kk, device=DEV)), (bad, jj, kk)):
        with pytest.raises(Exception):
            Plan(*idx, g.poses.shape[0], g.patches.shape[0], fixedp)
    oob = kk.copy()
    oob[7] = g.patches.shape[0]
    with pytest.raises(Exception):
        Plan(torch.as_tensor(ii, device=DEV), torch.as_tensor(jj, device=DEV), torch.as_tensor(oob, device=DEV), g.poses.shape[0], g.patches.shape[0], fixedp)
    # and the next list after the refused ones is planned on the device again (the per-patch table was left clean)
    dev, host = both_plans(ii, jj, kk, g.poses.shape[0], g.patches.shape[0], fixedp)
    assert dev.built_on_device and (dev.array("pm_edge") == host.array("pm_edge")).all()


WPT_TABLES = ("slot_code", "tile_la", "tile_rec", "it_edge", "tile_sinfo")


def compare_wave_per_tile_tables(dev, host):
    """slot_code, tile_la, tile_rec always; the edge-major tables of k_edge2 where the host found every tile slot-uniform (and
    the device must have reached the same verdict: the two plans launch the same kernel)."""
    assert dev.jacobian_kernel == host.jacobian_kernel
    uniform = host.array("it_edge").size > 0
    for name in WPT_TABLES:
        if name in ("it_edge", "tile_sinfo") and not uniform:
            continue
        a, b = dev.array(name), host.array(name)
        if name == "tile_rec" and not uniform:
            a, b = a.reshape(-1, 8)[:, :6], b.reshape(-1, 8)[:, :6]      # (words 6 and 7 belong to the edge-major layout)
        assert a.shape == b.shape and (a == b).all(), name


SLOT_TABLES = ("slot_edge", "slot_pair", "slot_lab", "slot_lp", "tile_cut8", "tile_cut16", "kx", "pair_i", "pair_j", "tile_trk0", "tile_ntrk", "tile_ncam",
               "tile_cams", "tile_slot0", "tile_nslot", "tile_pair0", "tile_npair", "tile_pairs", "tile_flags", "tile_ij", "tile_kx", "col_ptr", "row_idx",
               "perm", "act_bits", "act_rank")


@pytest.mark.skipif(FORCED, reason="kernel / planner selection forced by the environment")
@pytest.mark.parametrize("variant", ["C3", "C3_shuffled", "small", "repeats", "thinned", "large2048", "large8192_shuffled"])
def test_device_planned_slot_arrays_equal_the_hosts(variant):
    """64-track layouts (k_tile): the [slots][64] arrays and the waves' slot cuts written by kernels — since round 4 for graphs
    of any tile count (2048 and 8192 tiles: the sizes the host used to analyse for the float32 wave-per-tile kernels)."""
    rng = np.random.default_rng(3)
    if variant.startswith("C3"):
        g, fixedp = graphgen.make_config("C3", seed=0), 1
    elif variant.startswith("large"):
        g, fixedp = graphgen.make_graph(64, 2048 if "2048" in variant else 8192, 8, seed=6), 1
    elif variant == "small":
        g, fixedp = graphgen.make_graph(16, 128, 6, seed=2), 2
    else:
        g, fixedp = graphgen.make_graph(32, 256, 8, seed=5), 1
    ii, jj, kk = (np.asarray(a) for a in (g.ii, g.jj, g.kk))
    if variant == "repeats":                    # repeated observations: runs of one target camera inside a track (the cuts avoid them)
        extra = rng.integers(0, ii.size, ii.size // 2)
        ii, jj, kk = np.concatenate([ii, ii[extra]]), np.concatenate([jj, jj[extra]]), np.concatenate([kk, kk[extra]])
    if variant == "thinned":
        keep = rng.random(ii.size) > 0.3
        ii, jj, kk = ii[keep], jj[keep], kk[keep]
    if variant in ("C3_shuffled", "repeats", "large8192_shuffled"):
        p = rng.permutation(ii.size)
        ii, jj, kk = ii[p], jj[p], kk[p]
    from batrack_amd.plan import wave_per_tile_kernels
    prev = wave_per_tile_kernels(False) if variant.startswith("large") else None      # (the float64 tile layout at every size)
    try:
        dev, host = both_plans(ii, jj, kk, g.poses.shape[0], g.patches.shape[0], fixedp)
    finally:
        if prev is not None:
            wave_per_tile_kernels(prev)
    assert dev.built_on_device and not host.built_on_device and dev.jacobian_kernel == host.jacobian_kernel == "k_tile"
    for f in ("E", "m", "n", "tiles", "pairs", "slots", "nnz_blocks", "workspace_bytes"):
        assert getattr(dev, f) == getattr(host, f), f
    for name in SLOT_TABLES:
        a, b = dev.array(name), host.array(name)
        assert a.shape == b.shape and (a == b).all(), name


@pytest.mark.skipif(FORCED, reason="kernel / planner selection forced by the environment")
@pytest.mark.parametrize("seed", range(12))
def test_random_edge_lists_plan_alike_on_device_and_host(seed):
    """Random edge lists (frames, tracks per frame, targets within +-20 frames of the source, repeats, self edges, gaps in the
    patch range, any fixedp, shuffled): whichever layout the planner picks, the device-planned plan has the host's tables."""
    rng = np.random.default_rng(100 + seed)
    n_frames, M = int(rng.integers(8, 48)), int(rng.choice([8, 16, 64, 128]))
    n_buf, p_tot = n_frames + int(rng.integers(0, 8)), (n_frames + 2) * M
    lo = int(rng.integers(0, max(1, n_frames // 3)))                       # the frames before `lo` carry no track
    src = np.repeat(np.arange(lo, n_frames), M)
    pat = src * M + np.tile(np.arange(M), n_frames - lo)
    alive = rng.random(pat.size) < rng.uniform(0.5, 1.0)                  # gaps in the patch range
    src, pat = src[alive], pat[alive]
    span = int(rng.integers(2, 21))
    ii, jj, kk = [], [], []
    for s, p in zip(src, pat):
        tg = np.arange(max(0, s - span), min(n_frames, s + span + 1))
        tg = tg[rng.random(tg.size) < rng.uniform(0.3, 1.0)]
        if seed % 3 and tg.size:
            tg = tg[tg != s] if tg.size > 1 else tg                       # (self edges only in every third list)
        if tg.size == 0:
            continue
        rep = rng.integers(1, 4, tg.size) if seed % 2 else np.ones(tg.size, np.int64)
        tg = np.repeat(tg, rep)
        ii.append(np.full(tg.size, s)); jj.append(tg); kk.append(np.full(tg.size, p))
    ii, jj, kk = (np.concatenate(a).astype(np.int64) for a in (ii, jj, kk))
    while ii.size < 4200:                                                 # (the device path starts at 4096 edges)
        ii, jj, kk = np.concatenate([ii, ii]), np.concatenate([jj, jj]), np.concatenate([kk, kk])
    p = rng.permutation(ii.size)
    ii, jj, kk = ii[p], jj[p], kk[p]
    fixedp = int(rng.integers(0, max(1, int(max(ii.max(), jj.max())))))
    dev, host = both_plans(ii, jj, kk, n_buf, p_tot, fixedp)
    assert dev.jacobian_kernel == host.jacobian_kernel and dev.built_on_device
    for f in ("E", "m", "n", "tiles", "pairs", "slots", "nnz_blocks", "workspace_bytes"):
        assert getattr(dev, f) == getattr(host, f), f
    names = TABLES if dev.jacobian_kernel == "k_etile" else SLOT_TABLES
    for name in names:
        a, b = dev.array(name), host.array(name)
        assert a.shape == b.shape and (a == b).all(), (name, dev.jacobian_kernel)
    if host.tiles >= 2048:
        compare_wave_per_tile_tables(dev, host)


@pytest.mark.skipif(FORCED, reason="kernel / planner selection forced by the environment")
@pytest.mark.parametrize("graph", ["window", "window_shuffled", "C3", "random"])
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_plans_are_laid_out_on_the_device_too(graph, world):
    """A rank's plan of a sharded solve (own = its range of the patch buffer, parallel.py: plan_range): the device's sort keeps
    the rank's edges in one segment of the sorted list, the passes run on that segment, and the pattern of the all-reduced system
    comes from the other ranks' (source frame, target mask) — every table equals the host's analysis of the same list and range."""
    from batrack_amd.parallel import partition_tracks, plan_range
    rng = np.random.default_rng(17)
    if graph.startswith("window"):
        g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
    elif graph == "C3":
        g, fixedp = graphgen.make_config("C3", seed=0), 1
    else:
        g, fixedp = graphgen.make_graph(24, 512, 7, seed=8), 2
    ii, jj, kk = (np.asarray(a) for a in (g.ii, g.jj, g.kk))
    if graph in ("window_shuffled", "random"):
        p = rng.permutation(ii.size)
        ii, jj, kk = ii[p], jj[p], kk[p]
    n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
    ranges = partition_tracks(kk, world)
    built = 0
    for r in range(world):
        own = plan_range(ranges[r], p_tot)
        dev = Plan(*(torch.as_tensor(a, device=DEV) for a in (ii, jj, kk)), n_buf, p_tot, fixedp, own=own)
        host = Plan(ii, jj, kk, n_buf, p_tot, fixedp, own=own)
        assert not host.built_on_device and dev.jacobian_kernel == host.jacobian_kernel
        built += int(dev.built_on_device)
        for f in ("E", "m", "n", "tiles", "pairs", "slots", "nnz_blocks", "workspace_bytes"):
            assert getattr(dev, f) == getattr(host, f), (f, r)
        names = (TABLES if dev.jacobian_kernel == "k_etile" else SLOT_TABLES) + ("trk_off",)
        for name in names:
            a, b = dev.array(name), host.array(name)
            assert a.shape == b.shape and (a == b).all(), (name, r, dev.jacobian_kernel)
    assert built == world                       # (every rank of these lists holds 4096 edges or more in total and has tracks)


@pytest.mark.skipif(FORCED, reason="kernel / planner selection forced by the environment")
@pytest.mark.parametrize("variant", ["k_edge_2048", "k_edge_8192", "k_edge_8192_shuffled", "repeats", "ragged", "sharded"])
def test_device_planned_tables_of_the_wave_per_tile_kernels_equal_the_hosts(variant):
    """Graphs of 2048 tiles and more (k_stream, k_edge2): slot_code, tile_la, the tile records with their straddle flag, and — where
    every tile is slot-uniform — it_edge and tile_sinfo come from kernels (plan_device.hip) and equal the host's."""
    rng = np.random.default_rng(23)
    g, fixedp = graphgen.make_graph(64, 2048 if variant == "k_edge_2048" else 8192 if "8192" in variant else 4096, 8, seed=6), 1
    ii, jj, kk = (np.asarray(a) for a in (g.ii, g.jj, g.kk))
    if variant == "repeats":                    # repeated observations: runs across the half-chunk boundary (straddle flags), repeat bits
        extra = rng.integers(0, ii.size, ii.size // 2)
        ii, jj, kk = np.concatenate([ii, ii[extra]]), np.concatenate([jj, jj[extra]]), np.concatenate([kk, kk[extra]])
    if variant == "ragged":                     # tracks of different lengths: slot-uniform all the same (aligned slots with null entries, ba_plan.cpp)
        keep = rng.random(ii.size) > 0.15
        ii, jj, kk = ii[keep], jj[keep], kk[keep]
    if variant in ("k_edge_8192_shuffled", "repeats"):
        p = rng.permutation(ii.size)
        ii, jj, kk = ii[p], jj[p], kk[p]
    n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
    own = (0, 0)
    if variant == "sharded":
        from batrack_amd.parallel import partition_tracks, plan_range
        own = plan_range(partition_tracks(kk, 2)[1], p_tot)
    dev = Plan(*(torch.as_tensor(a, device=DEV) for a in (ii, jj, kk)), n_buf, p_tot, fixedp, own=own)
    host = Plan(ii, jj, kk, n_buf, p_tot, fixedp, own=own)
    assert dev.built_on_device and not host.built_on_device and host.tiles >= 2048
    assert dev.jacobian_kernel == host.jacobian_kernel and host.jacobian_kernel in ("k_stream", "k_edge2")
    if variant.startswith("k_"):
        assert host.jacobian_kernel == "k_edge2"
    if variant == "ragged":
        assert host.jacobian_kernel == "k_edge2"           # (round 5: k_stream — a missing observation used to shift the rest of its track)
    for f in ("E", "m", "n", "tiles", "pairs", "slots", "nnz_blocks", "workspace_bytes"):
        assert getattr(dev, f) == getattr(host, f), f
    for name in SLOT_TABLES + ("trk_off",):
        a, b = dev.array(name), host.array(name)
        assert a.shape == b.shape and (a == b).all(), name
    compare_wave_per_tile_tables(dev, host)


@pytest.mark.skipif(FORCED, reason="kernel / planner selection forced by the environment")
@pytest.mark.parametrize("seed", range(4))
def test_random_many_tile_lists_plan_alike_on_device_and_host(seed):
    """Random lists of 2048 tiles and more (tracks per frame in the thousands, 2..9 targets per track within +-12 frames, repeats
    in every second list, shuffled, any fixedp; odd seeds: a rank's range of a sharded solve): whichever wave-per-tile kernel the
    planner picks, every table of the device-planned plan is the host's."""
    rng = np.random.default_rng(900 + seed)
    n_frames, M = int(rng.integers(28, 40)), int(rng.choice([6144, 8192])) * (2 if seed % 2 or seed == 0 else 1)     # (a rank of two keeps 2048 tiles; seed 0: 4096 for k_edge2)
    n_buf, p_tot = n_frames + 2, (n_frames + 1) * M
    src = np.repeat(np.arange(n_frames), M)
    pat = src * M + np.tile(np.arange(M), n_frames)
    alive = rng.random(pat.size) < (2.0 if seed == 0 else 0.97)          # (seed 0: no gaps, tiles inside a frame: slot-uniform, k_edge2)
    src, pat = src[alive], pat[alive]
    # per track: a window of targets around the source frame, the same for the tracks of a frame in even seeds (slot-uniform
    # tiles: k_edge2), thinned per track in odd ones (k_stream)
    span = int(rng.integers(2, 5))                                       # (at most 10 cameras per tile: the wave-per-tile kernels' limit)
    off = np.arange(-span, span + 1)
    off = off[off != 0]
    tg = src[:, None] + off[None, :]
    ok = (tg >= 0) & (tg < n_frames)
    if seed % 2:
        ok &= rng.random(tg.shape) < 0.8
    ii = np.broadcast_to(src[:, None], tg.shape)[ok]
    kk = np.broadcast_to(pat[:, None], tg.shape)[ok]
    jj = tg[ok]
    if seed >= 2:                                                        # repeated observations
        extra = rng.integers(0, ii.size, ii.size // 3)
        ii, jj, kk = np.concatenate([ii, ii[extra]]), np.concatenate([jj, jj[extra]]), np.concatenate([kk, kk[extra]])
    p = rng.permutation(ii.size)
    ii, jj, kk = (a[p].astype(np.int64) for a in (ii, jj, kk))
    fixedp = int(rng.integers(1, 4))
    own = (0, 0)
    if seed % 2:
        from batrack_amd.parallel import partition_tracks, plan_range
        own = plan_range(partition_tracks(kk, 2)[seed // 2 % 2], p_tot)
    dev = Plan(*(torch.as_tensor(a, device=DEV) for a in (ii, jj, kk)), n_buf, p_tot, fixedp, own=own)
    host = Plan(ii, jj, kk, n_buf, p_tot, fixedp, own=own)
    assert dev.built_on_device and not host.built_on_device and host.tiles >= 2048, host.tiles
    assert dev.jacobian_kernel == host.jacobian_kernel
    for f in ("E", "m", "n", "tiles", "pairs", "slots", "nnz_blocks", "workspace_bytes"):
        assert getattr(dev, f) == getattr(host, f), f
    for name in SLOT_TABLES + ("trk_off",):
        a, b = dev.array(name), host.array(name)
        assert a.shape == b.shape and (a == b).all(), name
    compare_wave_per_tile_tables(dev, host)            # (the tables exist whichever kernel the tiles' camera counts admit)
    assert seed != 0 or host.jacobian_kernel == "k_edge2", (host.jacobian_kernel, host.tiles)


@pytest.mark.skipif(FORCED, reason="kernel / planner selection forced by the environment")
def test_small_tile_layout_that_reaches_2048_tiles_is_laid_out_again_on_the_device():
    """The round-3 advisor's list (3000 tracks x 26 observations, random source frames: every track closes its own 16-track tile
    at the camera limit, 2048 tiles and more) through the DEVICE planner: the second layout (64 tracks per tile) and its
    wave-per-tile tables come from the device's passes too, and equal the host's."""
    rng = np.random.default_rng(0)
    m, K, n_buf = 3000, 26, 120
    src = rng.integers(0, n_buf - K - 4, m)
    kk = np.repeat(np.arange(m), K).astype(np.int64)
    ii = np.repeat(src, K).astype(np.int64)
    jj = (ii + np.tile(np.arange(K), m)).astype(np.int64)
    dev, host = both_plans(ii, jj, kk, n_buf, m, 1)
    assert host.tiles >= 2048 and dev.built_on_device and dev.jacobian_kernel == host.jacobian_kernel
    for f in ("E", "m", "n", "tiles", "pairs", "slots", "nnz_blocks", "workspace_bytes"):
        assert getattr(dev, f) == getattr(host, f), f
    for name in SLOT_TABLES:
        a, b = dev.array(name), host.array(name)
        assert a.shape == b.shape and (a == b).all(), name
    compare_wave_per_tile_tables(dev, host)
    # and it steps like the host-planned one
    from batrack_amd.plan import Stepper
    g = graphgen.make_graph(8, 64, 4, seed=1)            # (only for tensors of the right kind: poses, intrinsics)
    torch.manual_seed(0)
    poses = torch.zeros(n_buf, 7, device=DEV); poses[:, 6] = 1; poses[:, :3] = 0.01 * torch.randn(n_buf, 3, device=DEV)
    patches = torch.rand(m, 3, 3, 3, device=DEV) * 100 + 50; patches[:, 2] = 0.5 + torch.rand(m, 1, 1, device=DEV)
    intr = torch.tensor([[300.0, 300.0, 320.0, 176.0]], device=DEV).repeat(n_buf, 1)
    mono = patches[:, 2, 1, 1].contiguous()
    targets = torch.rand(ii.size, 2, device=DEV) * 300 + 20
    weights = torch.rand(ii.size, 2, device=DEV)
    outs = []
    for pl in (dev, host):
        st = Stepper(pl, torch.device(DEV))
        po, pa = torch.empty_like(poses), patches.clone()
        st.step(poses, patches, mono, intr, targets, 2, weights, po, pa, (0, 0, 640, 352), 1e-4, 10.0, 0.05, "huber", False)
        torch.cuda.synchronize()
        outs.append((po.cpu(), pa.cpu()))
    assert torch.allclose(outs[0][0], outs[1][0], atol=1e-6, equal_nan=True) and torch.allclose(outs[0][1], outs[1][1], atol=1e-5, equal_nan=True)


@pytest.mark.skipif(FORCED, reason="kernel / planner selection forced by the environment")
@pytest.mark.parametrize("reach", [31, 45, 63, 64])
def test_targets_far_from_their_source_frame(reach):
    """A track's targets up to 63 frames from its source frame fit the device's 128-bit mask (round 3: 31); one frame more and
    the list goes to the host's analysis.  Either way the plan's tables are the host's."""
    rng = np.random.default_rng(reach)
    n_frames, M = 140, 48
    src = np.repeat(np.arange(n_frames), M)
    pat = src * M + np.tile(np.arange(M), n_frames)
    ii, jj, kk = [], [], []
    for s, p in zip(src, pat):
        near = np.arange(max(0, s - 3), min(n_frames, s + 4))
        far = np.array([f for f in (s - reach, s + reach) if 0 <= f < n_frames], np.int64)
        tg = np.concatenate([near[near != s], far])
        ii.append(np.full(tg.size, s)); jj.append(tg); kk.append(np.full(tg.size, p))
    ii, jj, kk = (np.concatenate(a).astype(np.int64) for a in (ii, jj, kk))
    p = rng.permutation(ii.size)
    ii, jj, kk = ii[p], jj[p], kk[p]
    dev, host = both_plans(ii, jj, kk, n_frames + 2, (n_frames + 1) * M, 5)
    assert dev.built_on_device == (reach <= 63) and dev.jacobian_kernel == host.jacobian_kernel
    for f in ("E", "m", "n", "tiles", "pairs", "slots", "nnz_blocks", "workspace_bytes"):
        assert getattr(dev, f) == getattr(host, f), f
    names = TABLES if dev.jacobian_kernel == "k_etile" else SLOT_TABLES
    for name in names:
        a, b = dev.array(name), host.array(name)
        assert a.shape == b.shape and (a == b).all(), name


def test_track_partition_on_the_device_equals_the_hosts():
    """partition_tracks of a device tensor (torch.unique / cumsum / searchsorted there, `world` numbers back) gives the bounds the
    numpy statement gives — every rank derives every rank's range from them."""
    from batrack_amd.parallel import partition_tracks
    rng = np.random.default_rng(3)
    for trial in range(6):
        m = int(rng.integers(1, 5000))
        kk = rng.choice(np.arange(0, 20000, 3), size=m, replace=True)
        kk = np.repeat(kk, rng.integers(1, 12, kk.size)).astype(np.int64)
        rng.shuffle(kk)
        for world in (1, 2, 3, 8, 16):
            assert partition_tracks(torch.as_tensor(kk, device=DEV), world) == partition_tracks(kk, world), (trial, world)
    assert partition_tracks(torch.as_tensor(np.array([5, 5, 5], np.int64), device=DEV), 4) == partition_tracks(np.array([5, 5, 5], np.int64), 4)
