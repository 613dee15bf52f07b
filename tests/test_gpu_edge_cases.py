"""-m gpu: degenerate inputs the reference handles (or masks) by construction, against the float64 oracle:
a single edge, a single track, two frames, every edge masked out, every pose fixed (ba.py:316 takes the structure-only
branch), tracks far apart in a mostly empty patch buffer, a self edge only (ii == jj)."""
import numpy as np
import pytest

import oracle
from edge_problems import problem
from gpu_util import HipProblem, rel

pytestmark = pytest.mark.gpu


def check(d, fixedp, so=False):
    wkey = "weights" if so else "weights_pose"
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"], d["bounds"],
                         fixedp=fixedp, structure_only=so, want_system=False)
    o = HipProblem(d).raw_step(wkey, fixedp, so)
    assert np.isfinite(o["poses_out"]).all() and np.isfinite(o["patches_out"]).all()
    assert rel(o["poses_out"], ref["poses_out"]) < 5e-6 and rel(o["patches_out"], ref["patches_out"]) < 5e-6
    return o, ref


def test_one_edge():
    o, ref = check(problem([0], [1], [3], n_buf=2, p_tot=8), fixedp=1)
    assert o["plan"].E == 1 and o["plan"].m == 1 and o["plan"].n == 1


def test_one_track_many_frames():
    check(problem([0] * 5, [1, 2, 3, 4, 5], [7] * 5, n_buf=6, p_tot=16), fixedp=1)


def test_two_frames_both_directions():
    ii = [0] * 20 + [1] * 20
    jj = [1] * 20 + [0] * 20
    kk = list(range(20)) + list(range(32, 52))
    check(problem(ii, jj, kk, n_buf=2, p_tot=64), fixedp=1)


def test_every_pose_fixed_takes_the_structure_only_branch():
    d = problem([0] * 10 + [1] * 10, [1] * 10 + [2] * 10, list(range(10)) + list(range(16, 26)), n_buf=3, p_tot=32)
    o, ref = check(d, fixedp=3)
    assert o["plan"].n == 0
    assert np.array_equal(o["poses_out"], np.asarray(d["poses"], np.float32))          # untouched (ba.py:316-318 returns the input poses)


def test_every_edge_masked_out():
    """Targets 300 px away: |r| > 250 masks every edge (ba.py:233); the damped system is ep * I, dX = 0, depths move by the
    prior term only."""
    d = problem([0] * 12, [1] * 6 + [2] * 6, list(range(6)) * 2, n_buf=3, p_tot=8, target_shift=300.0)
    o, ref = check(d, fixedp=1)
    assert o["status"] == 0 and float(np.abs(o["dX"]).max()) == 0.0


def test_sparse_use_of_a_large_patch_buffer():
    """Three tracks at the ends and the middle of a 262,144-slot buffer (BUFFER_SIZE x M of sintel.yaml): every other
    slot is copied and clamped only."""
    kk = [5, 131072, 262143]
    d = problem([0, 0, 0, 1, 1, 1], [1, 1, 1, 2, 2, 2], kk + kk, n_buf=4, p_tot=262144)
    d["ii"] = np.array([0, 0, 0, 0, 0, 0], np.int64)                     # one source frame per track (batrack.py:199)
    check(d, fixedp=1)


def test_self_edges_only():
    """ii == jj: the relative pose is the identity, the pose Jacobians cancel (Ji = -Jj); only the depths move."""
    d = problem([1] * 8, [1] * 8, list(range(8)), n_buf=3, p_tot=8)
    check(d, fixedp=1)
