"""-m gpu: the Jacobian kernels.  A plan picks k_tile, k_stream or k_edge2 (+ k_edge2u for the depth pass) from its size and shape; the
fixtures are small and all take k_tile, so (a) graphs of the benchmark generator large enough for the plan to pick k_stream
and k_edge2 BY ITSELF are compared with the float64 oracle, and (b) the whole parity suite is re-run in child processes with
the selection forced (the thresholds are read once per process), so every fixture that fits a kernel's layout goes through it."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from batrack_amd import graphgen
import force
from gpu_util import HipProblem, rel, update_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# gates (state, S and y, dX, update poses, update disparities) by the plan's precision of the per-edge maths: 8 float64, 6 mixed
# (k_stream / k_edge2: float64 reprojection and residual, float32 Jacobians), 4 float32 (BT_FORCE prec=f32: measurement)
GATES = {8: (2e-7, 1e-10, 1e-5, 1e-5, 1e-5), 6: (2e-7, 1e-6, 1e-5, 1e-5, 1e-5), 4: (5e-6, 5e-6, 3e-4, 3e-4, 1e-4)}


def deep(g, times=9):
    """Every observation `times` times: more than 64 (pair, repeat) slots per track — the edge-major layout of k_edge2 does not hold
    such tiles, the plan stays with k_stream (what is left for that kernel since the aligned slots of round 6)."""
    import copy
    h = copy.copy(g)
    for name in ("ii", "jj", "kk", "targets3", "weights", "weights_pose"):
        setattr(h, name, np.ascontiguousarray(np.repeat(np.asarray(getattr(g, name)), times, axis=0)))
    return h


def ragged(g, drop=0.15):
    """The graph without a random 15 % of its observations: tracks of different lengths.  Round 5: tiles that are not slot-uniform,
    k_stream's graphs; round 6: slot-uniform again by aligned slots with null entries (ba_plan.cpp), k_edge2's."""
    keep = np.random.default_rng(11).random(np.asarray(g.kk).size) > drop
    import copy
    h = copy.copy(g)
    for name in ("ii", "jj", "kk", "targets3", "weights", "weights_pose"):
        setattr(h, name, np.ascontiguousarray(np.asarray(getattr(g, name))[keep]))
    return h


@pytest.mark.parametrize("frames,M,kernel,so,wpt", [(16, 16, "k_tile", False, True), (64, 1024, "k_tile", False, True),
                                                  # from 2048 tiles the wave-per-tile kernels, mixed precision: inside the 1e-5 bar:
                                                  # k_edge2 / k_edge2u where the tiles are slot-uniform, k_stream where they are not ...
                                                  (64, 2048, "k_stream", False, True), (64, 2048, "k_stream", True, True),
                                                  (64, 2048, "ragged", False, True), (64, 2048, "ragged", True, True), (64, 8192, "ragged", False, True),
                                                  (64, 2048, "k_edge2", False, True), (64, 4096, "k_edge2", False, True),
                                                  (64, 6144, "k_edge2", False, True), (64, 6144, "k_edge2", True, True),
                                                  # the size the roofline figures are quoted at: 8.4M edges, 16384 tiles, 8 tiles per wave
                                                  (64, 16384, "k_edge2", False, True), (64, 16384, "k_edge2", True, True),
                                                  # ... or, switched off by the caller, the float64 tile kernel at every size
                                                  (64, 2048, "k_tile", False, False), (64, 2048, "k_tile", True, False), (64, 6144, "k_tile", False, False)])
def test_plan_selected_kernel_vs_oracle(frames, M, kernel, so, wpt):
    from batrack_amd.plan import wave_per_tile_kernels
    g = graphgen.make_graph(frames, M, 8, seed=5)
    if kernel == "k_stream":
        g = deep(g)
    if kernel == "ragged":
        g, kernel = ragged(g), "k_edge2"
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics), targets3=f(g.targets3),
             weights=f(g.weights), weights_pose=f(g.weights_pose), ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds))
    wkey = "weights" if so else "weights_pose"
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"],
                         d["bounds"], fixedp=1, structure_only=so, want_system=not so)
    prev = wave_per_tile_kernels(wpt)
    try:
        o = HipProblem(d).raw_step(wkey, 1, so)
    finally:
        wave_per_tile_kernels(prev)
    assert o["plan"].jacobian_kernel == kernel, (o["plan"].jacobian_kernel, o["plan"].tiles)
    prec = o["plan"].edge_precision
    assert prec == (4 if force.tokens().get("prec") == "f32" and kernel == "k_tile" else 6 if kernel in ("k_stream", "k_edge2") else 8)
    act = np.unique(g.kk)
    t_state, t_sys, t_dx, t_upd_p, t_upd_d = GATES[prec]
    assert update_err(o["patches_out"][:, 2], ref["patches_out"][:, 2], d["patches"][:, 2], act) < t_upd_d
    assert rel(o["patches_out"], ref["patches_out"]) < t_state
    if not so:
        assert o["status"] == 0
        assert rel(np.tril(o["S_lower"]), np.tril(ref["S"])) < t_sys and rel(o["y"], ref["y"]) < t_sys
        assert rel(o["dX"].reshape(-1), ref["dX"].reshape(-1)) < t_dx
        assert update_err(o["poses_out"], ref["poses_out"], d["poses"], np.arange(1, 1 + o["plan"].n)) < t_upd_p
        assert rel(o["poses_out"], ref["poses_out"]) < t_state


@pytest.mark.skipif("wpt" in force.tokens(), reason="the default of the setting is what the test starts from")
def test_the_kernel_choice_is_the_plans_own():
    """A plan keeps the layout it was built with: switching the setting afterwards changes neither its kernel nor its tables."""
    import torch
    from batrack_amd.plan import Plan, wave_per_tile_kernels
    g = deep(graphgen.make_graph(64, 2048, 8, seed=2))
    T = lambda a: torch.as_tensor(a, device="cuda:0")
    ii, jj, kk = T(g.ii), T(g.jj), T(g.kk)
    assert wave_per_tile_kernels() is True                              # the default
    pw = Plan(ii, jj, kk, g.poses.shape[0], g.patches.shape[0], 1)
    prev = wave_per_tile_kernels(False)
    try:
        pt = Plan(ii, jj, kk, g.poses.shape[0], g.patches.shape[0], 1)
        assert (pw.jacobian_kernel, pw.edge_precision) == ("k_stream", 6) and (pt.jacobian_kernel, pt.edge_precision) == ("k_tile", 8)
    finally:
        wave_per_tile_kernels(prev)
    assert (pt.jacobian_kernel, pt.edge_precision) == ("k_tile", 8) and (pw.jacobian_kernel, pw.edge_precision) == ("k_stream", 6)
    assert wave_per_tile_kernels() is True


# BT_FORCE kernel=...: k_edge2 wherever the tiles are slot-uniform (else k_stream), k_etile = the pair-major tile kernel for every plan
FORCED = {k: dict(BT_FORCE=f"kernel={k}") for k in ("k_edge2", "k_stream", "k_tile", "k_etile")}


@pytest.mark.parametrize("kernel", ["k_edge2", "k_stream", "k_tile", "k_etile"])
def test_parity_suite_with_the_selection_forced(kernel):
    env = dict(os.environ, **FORCED[kernel])
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_solver_variants.py", "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    # and the forced choice is what a slot-uniform graph really takes
    code = ("import numpy as np, torch; from batrack_amd import graphgen; from batrack_amd.plan import Plan; g = graphgen.make_graph(16, 64, 8, seed=1);"
            "T = lambda a: torch.as_tensor(a, device='cuda:0'); p = Plan(T(g.ii), T(g.jj), T(g.kk), g.poses.shape[0], g.patches.shape[0], 1); print(p.jacobian_kernel)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == kernel, r.stdout + r.stderr[-2000:]


def test_uploaded_plan_of_many_one_track_tiles_steps_like_the_oracle():
    """The advisor's round-3 crash case on the device path: 3000 tracks x 26 observations with random source frames — first
    tiled for k_etile, 2967 tiles of one track each, laid out again with 64-track tiles and stepped by a wave-per-tile kernel."""
    from test_plan_cpu import _many_small_tiles_graph
    from edge_problems import problem
    ii, jj, kk, n_buf, m = _many_small_tiles_graph(m=3000)
    d = problem(ii, jj, kk, n_buf, m, seed=3)
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"], d["ii"], d["jj"], d["kk"],
                         d["bounds"], fixedp=1, want_system=True)
    o = HipProblem(d).raw_step("weights_pose", 1)
    assert o["plan"].tiles >= 2048 and o["plan"].jacobian_kernel in ("k_stream", "k_edge2", "k_tile")
    assert o["status"] == 0
    assert rel(np.tril(o["S_lower"]), np.tril(ref["S"])) < 5e-6 and rel(o["y"], ref["y"]) < 5e-6
    assert rel(o["poses_out"], ref["poses_out"]) < 5e-6 and rel(o["patches_out"], ref["patches_out"]) < 5e-6


_ONE_OBS = r"""
import numpy as np, sys
sys.path.insert(0, "tests")
import oracle
from edge_problems import problem
from gpu_util import HipProblem, rel
# tracks with ONE observation each (S = 1 slot per track: an iteration is a whole 64-track tile), three full tiles per source frame and
# a partial one at the end: the advisor's round-5 case — an odd iteration count per tile used to shift every later tile by a row
n_buf = 6
ii = np.concatenate([np.repeat(np.arange(n_buf - 1), 192), np.full(100, n_buf - 1)])
jj = (ii + 1) % n_buf                                              # every track of a frame sees the next frame: slot-uniform tiles
kk = np.arange(ii.size)
d = problem(ii, jj, kk, n_buf, ii.size, seed=9)
ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"], d["ii"], d["jj"], d["kk"],
                     d["bounds"], fixedp=1, want_system=True)
o = HipProblem(d).raw_step("weights_pose", 1)
assert o["plan"].jacobian_kernel == "k_edge2", o["plan"].jacobian_kernel
assert o["status"] == 0
es, ey = rel(np.tril(o["S_lower"]), np.tril(ref["S"])), rel(o["y"], ref["y"])
ep, ed = rel(o["poses_out"], ref["poses_out"]), rel(o["patches_out"], ref["patches_out"])
print("one observation per track:", o["plan"].tiles, "tiles", es, ey, ep, ed)
# (a track with ONE observation leaves y = v - E Q w' as the small difference of two nearly equal vectors — each residual is all
#  but absorbed by its own depth: the mixed-precision kernels' 1e-7 on v and E shows as 1e-5 on y; an off-by-one row shows as O(1))
assert es < 5e-6 and ey < 1e-4 and ep < 3e-4 and ed < 5e-6, (es, ey, ep, ed)
"""


def test_one_observation_per_track_through_the_two_edge_kernel():
    """S = 1 tiles (one observation per track) forced onto k_edge2: every tile's iteration count is padded to even (ba_plan.cpp
    em_iterations), the second row of a step is empty and masked."""
    env = dict(os.environ, **FORCED["k_edge2"])
    r = subprocess.run([sys.executable, "-c", _ONE_OBS], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
