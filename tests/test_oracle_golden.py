"""The CPU oracle (oracle/ba_oracle_impl.h) against the golden vectors produced by
the reference's own ba.py / projective_ops.py (tests/golden/make_golden.py).

float64: the restatement must reproduce the reference to round-off (<= 1e-9
relative on states, <= 1e-8 on the reduced system whose entries reach 1e7).
float32: only to fp32 noise, since summation order differs."""
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def call(d, wkey, fixedp, so, dtype, loss="huber", poses=None, patches=None, **kw):
    return oracle.ba_step(d["poses"] if poses is None else poses,
                          d["patches"] if patches is None else patches,
                          d["mono"], d["intrinsics"], d["targets3"], d[wkey],
                          d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=fixedp,
                          structure_only=so, loss=loss, dtype=dtype, want_system=True, **kw)


@pytest.mark.parametrize("name", ["c1", "c1_rough"])
def test_edge_jacobians_match_reference_transform(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    o = oracle.edges(d["poses"], d["patches"], d["intrinsics"], d["targets3"], d["weights"],
                     d["ii"], d["jj"], d["kk"], d["bounds"])
    assert np.array_equal(o["valid"] >= 0, np.ones_like(o["valid"], bool))
    assert rel(o["coords"], d["tf64.coords"]) < 1e-12
    assert rel(o["Ji"], d["tf64.Ji"]) < 1e-11
    assert rel(o["Jj"], d["tf64.Jj"]) < 1e-12
    assert rel(o["Jz"], d["tf64.Jz"]) < 1e-12
    # reference 'valid' from transform() is the depth test only (projective_ops.py:100)
    Zok = d["tf64.valid"] > 0
    assert np.all((o["valid"] > 0) <= Zok)


CASES = [
    ("c1", "ps_fp1", "weights_pose", 1, False, "huber", {}),
    ("c1", "ps_fp3", "weights_pose", 3, False, "huber", {}),
    ("c1", "so", "weights", 1, True, "huber", {}),
    ("c1", "triv", "weights_pose", 1, False, "trivial", {}),
    ("c1", "cauchy", "weights_pose", 1, False, "cauchy", {}),
    ("c1", "allfixed", "weights_pose", 8, False, "huber", {}),
    ("c1_rough", "ps_fp1", "weights_pose", 1, False, "huber", {}),
    ("c1_rough", "ps_fp2", "weights_pose", 2, False, "huber", dict(alpha=0.5, ep=100.0)),
    ("c1_rough", "so", "weights", 1, True, "huber", {}),
    ("window_small", "ps", "weights_pose", None, False, "huber", {}),
    ("window_small", "so", "weights", None, True, "huber", {}),
]


@pytest.mark.parametrize("name,tag,wkey,fixedp,so,loss,kw", CASES)
def test_step_f64_matches_reference(name, tag, wkey, fixedp, so, loss, kw):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    fixedp = int(d["fixedp"]) if fixedp is None else fixedp
    o = call(d, wkey, fixedp, so, np.float64, loss, **kw)
    assert rel(o["poses_out"], d[f"{tag}.f64.poses_out"]) < 1e-10
    assert rel(o["patches_out"], d[f"{tag}.f64.patches_out"]) < 1e-10
    if f"{tag}.f64.S" in d.files:
        assert rel(o["S"], d[f"{tag}.f64.S"]) < 1e-9
        assert rel(o["y"], d[f"{tag}.f64.y"]) < 1e-9
        assert rel(o["dX"], d[f"{tag}.f64.dX"]) < 1e-7
    else:
        assert "S" not in o
        if so or fixedp >= 8:
            assert np.array_equal(o["poses_out"], d["poses"])   # poses untouched


@pytest.mark.parametrize("name,tag,wkey,fixedp,so,loss,kw", CASES)
def test_step_f32_within_fp32_noise(name, tag, wkey, fixedp, so, loss, kw):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    fixedp = int(d["fixedp"]) if fixedp is None else fixedp
    o = call(d, wkey, fixedp, so, np.float32, loss, **kw)
    # the reference's own fp32-vs-fp64 gap on these graphs is 1e-5..2e-4
    assert rel(o["poses_out"], d[f"{tag}.f64.poses_out"]) < 5e-4
    assert rel(o["patches_out"], d[f"{tag}.f64.patches_out"]) < 5e-4


@pytest.mark.parametrize("name,fixedp", [("c1", 1), ("c1_rough", 2), ("window_small", None)])
def test_dual_iteration_pattern(name, fixedp):
    """BATRACK.update(): ITER x {pose+structure(weights_pose); structure-only(weights)}."""
    d = np.load(os.path.join(GOLD, name + ".npz"))
    fixedp = int(d["fixedp"]) if fixedp is None else fixedp
    poses, patches = d["poses"], d["patches"]
    for _ in range(2):
        o = call(d, "weights_pose", fixedp, False, np.float64, poses=poses, patches=patches)
        o = call(d, "weights", fixedp, True, np.float64, poses=o["poses_out"], patches=o["patches_out"])
        poses, patches = o["poses_out"], o["patches_out"]
    assert rel(poses, d["dual2.f64.poses_out"]) < 1e-9
    assert rel(patches, d["dual2.f64.patches_out"]) < 1e-9


def test_c3_full_size():
    """64 KF / 131,072 edges: inputs regenerated from the seed, outputs from the reference."""
    from batrack_amd import graphgen
    d = np.load(os.path.join(GOLD, "c3.npz"))
    g = graphgen.make_config("C3", seed=int(d["seed"]))
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    o = oracle.ba_step(f(g.poses), f(g.patches), f(g.mono_disp), f(g.intrinsics), f(g.targets3),
                       f(g.weights_pose), g.ii, g.jj, g.kk, g.bounds, fixedp=1, want_system=True)
    assert rel(o["poses_out"], d["ps.f64.poses_out"]) < 1e-10
    assert rel(o["patches_out"][:, 2], d["ps.f64.disp_out"]) < 1e-10
    assert rel(o["dX"], d["ps.f64.dX"]) < 1e-7
    assert rel(np.diag(o["S"]), d["ps.f64.S_diag"]) < 1e-10
    o2 = oracle.ba_step(o["poses_out"], o["patches_out"], f(g.mono_disp), f(g.intrinsics), f(g.targets3),
                        f(g.weights), g.ii, g.jj, g.kk, g.bounds, fixedp=1, structure_only=True)
    assert rel(o2["patches_out"][:, 2], d["so.f64.disp_out"]) < 1e-10


@pytest.mark.parametrize("loss", ["huber", "cauchy", "trivial"])
def test_nan_target_matches_reference(loss):
    """tests/golden/c1_nan.npz (make_golden.py nan): one NaN target on the C1 graph.  The reference masks by
    multiplication, so 0 * NaN poisons the right-hand side (huber / trivial: retry, every free pose and active
    disparity NaN) or the matrix (cauchy: failed factorisation, dX = 0, one disparity NaN).  Same NaN pattern and,
    where finite, the same numbers from the oracle."""
    d = dict(np.load(os.path.join(GOLD, "c1.npz")))
    gold = dict(np.load(os.path.join(GOLD, "c1_nan.npz")))
    t = d["targets3"].copy()
    t[int(gold["e0"]), 0] = np.nan
    o = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], t, d["weights_pose"],
                       d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1, loss=loss, want_system=True)
    for name in ("poses_out", "patches_out"):
        ref = gold[f"{loss}.{name}"]
        assert np.array_equal(np.isnan(ref), np.isnan(o[name])), name
        ok = ~np.isnan(ref)
        assert np.abs(o[name][ok] - ref[ok]).max() < 1e-9
    assert o["failed"] == (loss == "cauchy")
    if loss == "cauchy":
        assert int(gold["cauchy.n_solves"]) == 1 and np.all(gold["cauchy.dX"] == 0) and np.all(o["dX"] == 0)
    else:
        assert int(gold[f"{loss}.n_solves"]) == 2 and np.isnan(gold[f"{loss}.dX"]).all() and np.isnan(o["dX"]).all()


# ---- oracle.refseq: the torch-CPU restatement with the reference's operator sequence (the timed CPU baseline)
@pytest.mark.parametrize("name,tag,wkey,fixedp,so,loss,kw", CASES)
def test_refseq_f64_matches_reference(name, tag, wkey, fixedp, so, loss, kw):
    import torch
    from oracle import refseq
    d = np.load(os.path.join(GOLD, name + ".npz"))
    fixedp = int(d["fixedp"]) if fixedp is None else fixedp
    t = lambda a: torch.as_tensor(np.asarray(a, np.float64))
    o = refseq.ba_step(t(d["poses"]), t(d["patches"]), t(d["mono"]), t(d["intrinsics"]), t(d["targets3"]), t(d[wkey]),
                       torch.as_tensor(d["ii"]), torch.as_tensor(d["jj"]), torch.as_tensor(d["kk"]),
                       [float(x) for x in d["bounds"]], fixedp=fixedp, structure_only=so, loss=loss, want_system=True, **kw)
    assert rel(o["poses_out"].numpy(), d[f"{tag}.f64.poses_out"]) < 1e-10
    assert rel(o["patches_out"].numpy(), d[f"{tag}.f64.patches_out"]) < 1e-10
    if f"{tag}.f64.S" in d.files:
        assert rel(o["S"].numpy(), d[f"{tag}.f64.S"]) < 1e-9
        assert rel(o["y"].numpy(), d[f"{tag}.f64.y"]) < 1e-9
        assert rel(o["dX"].numpy(), d[f"{tag}.f64.dX"]) < 1e-7


def test_refseq_f32_c3_close_to_reference_float32():
    """Full-size C3 in float32, as bench.py times it: within float32 noise of the reference's float64 result."""
    import torch
    from oracle import refseq
    from batrack_amd import graphgen
    g = graphgen.make_config("C3", seed=0)
    gold = np.load(os.path.join(GOLD, "c3.npz"))
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    o = refseq.ba_step(t(g.poses), t(g.patches), t(g.mono_disp), t(g.intrinsics), t(g.targets3), t(g.weights_pose),
                       torch.as_tensor(g.ii), torch.as_tensor(g.jj), torch.as_tensor(g.kk), list(g.bounds), fixedp=1)
    assert rel(o["poses_out"].numpy(), gold["ps.f64.poses_out"]) < 5e-4
    assert rel(o["patches_out"].numpy()[:, 2], gold["ps.f64.disp_out"]) < 5e-4
