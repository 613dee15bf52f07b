"""-m gpu: the HIP path (C ABI -> gfx950 kernels) against the golden vectors of the
reference and against the float64 oracle.

Tolerances (tests/gpu_util.py: TOL; tools/gpu_check.py prints the measured numbers, profiles/r03_parity_numbers.txt):
the per-edge maths of every graph that takes k_tile — all fixtures, every real window — is float64 on the float32
inputs, so against the reference's float64 result
  reduced system S, y            <= 1e-10                                 (measured <= 7e-14)
  camera update dX               <= 1e-5   (north_star)                   (measured <= 2.6e-8: stored as float32)
  the UPDATE itself, over the entries the step touched (north_star's tolerance is on the update):
      poses' - poses over the free poses                <= 1e-5           (measured <= 1.3e-6; reference float32 8e-5 .. 6e-3)
      disparities' - disparities over the active tracks <= 1e-5           (measured <= 2.8e-7; reference float32 3e-6 .. 1.9e-3)
  state (poses', disparities')   <= 2e-7                                  (measured <= 2e-8: float32 rounding of the output)
With the float32 / mixed per-edge kernels forced (BT_FORCE kernel=k_stream / k_edge2, or prec=f32) the round-2 gates apply (5e-6 state,
4e-6 system, 3e-4 / 1e-4 update)."""
import os

import numpy as np
import pytest
import torch

import oracle
from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper
from gpu_util import F32_EDGE, TOL, HipProblem, rel, update_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
STATE_TOL = TOL["state"]
DX_TOL = TOL["dx"]
SYS_TOL = TOL["sys"]
UPD_POSE_TOL, UPD_DISP_TOL = TOL["upd_pose"], TOL["upd_disp"]


def tol(f64, f32):
    """A gate that depends on the precision of the per-edge maths (gpu_util.F32_EDGE)."""
    return f32 if F32_EDGE else f64


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


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
def test_reduced_system_and_update_vs_reference(name, tag, wkey, fixedp, so, loss, kw):
    d = load(name)
    fixedp = int(d["fixedp"]) if fixedp is None else fixedp
    o = HipProblem(d).raw_step(wkey, fixedp, so, loss, **kw)
    act = np.unique(d["kk"])
    disp_in = d["patches"][:, 2].astype(np.float32)
    if f"{tag}.f64.S" in d:
        Sref = d[f"{tag}.f64.S"]
        assert rel(np.tril(o["S_lower"]), np.tril(Sref)) < SYS_TOL
        assert rel(o["y"], d[f"{tag}.f64.y"]) < SYS_TOL
        assert rel(o["dX"].reshape(-1), d[f"{tag}.f64.dX"].reshape(-1)) < DX_TOL
        assert o["status"] == 0
        n = Sref.shape[0] // 6
        free = np.arange(fixedp, fixedp + n)
        poses_in = d["poses"].astype(np.float32)
        u_pose = update_err(o["poses_out"], d[f"{tag}.f64.poses_out"], poses_in, free)
        u_ref = update_err(d[f"{tag}.f32.poses_out"], d[f"{tag}.f64.poses_out"], poses_in, free)
        assert u_pose < UPD_POSE_TOL and (u_ref < 1e-4 or u_pose < u_ref), (u_pose, u_ref)
    u_disp = update_err(o["patches_out"][:, 2], d[f"{tag}.f64.patches_out"][:, 2], disp_in, act)
    u_ref = update_err(d[f"{tag}.f32.patches_out"][:, 2], d[f"{tag}.f64.patches_out"][:, 2], disp_in, act)
    assert u_disp < UPD_DISP_TOL and (u_ref < 1e-4 or u_disp < u_ref), (u_disp, u_ref)
    e_pose = rel(o["poses_out"], d[f"{tag}.f64.poses_out"])
    e_pat = rel(o["patches_out"], d[f"{tag}.f64.patches_out"])
    assert e_pose < STATE_TOL and e_pat < STATE_TOL, (e_pose, e_pat)
    # no worse than the reference's own float32 run
    assert e_pose <= max(2 * rel(d[f"{tag}.f32.poses_out"], d[f"{tag}.f64.poses_out"]), 1e-6)


@pytest.mark.parametrize("name,fixedp", [("c1", 1), ("c1_rough", 2), ("window_small", None)])
def test_api_dual_iterations(name, fixedp):
    """BA_rgbd_droid called as BATRACK.update() does (batrack.py:869-875), 2 dual iterations."""
    d = load(name)
    fixedp = int(d["fixedp"]) if fixedp is None else fixedp
    hp = HipProblem(d)
    Gs, pat = None, None
    for _ in range(2):
        Gs, pat = hp.api_step("weights_pose", fixedp, False, poses=Gs, patches=pat)
        Gs2, pat = hp.api_step("weights", fixedp, True, poses=Gs, patches=pat)
        assert Gs2 is Gs                                      # structure-only returns the same object
    torch.cuda.synchronize()
    assert tuple(pat.shape) == (1, d["patches"].shape[0], 3, 1, 1)
    assert rel(Gs.data[0].cpu().numpy(), d["dual2.f64.poses_out"]) < tol(1e-6, 2e-5)
    assert rel(pat[0, :, :, 0, 0].cpu().numpy(), d["dual2.f64.patches_out"]) < tol(1e-6, 2e-5)
    # inputs untouched (functional semantics, ba.py:332-339)
    assert np.array_equal(hp.poses[0].cpu().numpy(), d["poses"].astype(np.float32))
    assert np.array_equal(hp.patches[0, :, :, 0, 0].cpu().numpy(), d["patches"].astype(np.float32))


def test_patches_of_size_three_step_like_their_centres():
    """Patch size p = 3 (ba.py:228-230 projects the centre pixel, :307 takes the prior at pixel (0, 0), :332-334 adds dZ to the whole
    disparity plane): the step on [1, P, 3, 3, 3] patches with per-patch-constant disparity is the p = 1 step of the fixture — the
    reference's float64 result — at every pixel; a plane that is not constant is refused."""
    d = load("c1_rough")
    hp = HipProblem(d)
    p1 = hp.patches                                                           # [1, P, 3, 1, 1]
    off = torch.tensor([-1.0, 0.0, 1.0], device=p1.device)
    p3 = p1.repeat(1, 1, 1, 3, 3).clone()
    p3[:, :, 0] += off[None, None, None, :]                                    # x, y of the neighbouring pixels: never read by the step
    p3[:, :, 1] += off[None, None, :, None]
    Gs, pat = hp.api_step("weights_pose", 2, False, patches=p3, alpha=0.5, ep=100.0)      # (the settings of the fixture's ps_fp2 run)
    torch.cuda.synchronize()
    assert tuple(pat.shape) == (1, d["patches"].shape[0], 3, 3, 3)
    ref = d["ps_fp2.f64.patches_out"] if "ps_fp2.f64.patches_out" in d else None
    one = hp.api_step("weights_pose", 2, False, alpha=0.5, ep=100.0)
    assert torch.equal(pat[:, :, 2], one[1][:, :, 2].expand(-1, -1, 3, 3))     # every pixel of the plane: the centre's new disparity
    assert torch.equal(pat[:, :, :2], p3[:, :, :2])                            # x, y untouched
    assert rel(Gs.data.cpu().numpy(), one[0].data.cpu().numpy()) < 1e-6
    if ref is not None:
        assert rel(pat[0, :, :, 1, 1].cpu().numpy(), ref) < tol(1e-6, 2e-5)
    bad = p3.clone()
    bad[0, 0, 2, 0, 0] += 0.1
    with pytest.raises(NotImplementedError):
        hp.api_step("weights_pose", 2, False, patches=bad)


def test_strided_depth_prior_is_used_in_place():
    """The caller's prior is the view patches_local[:, :, mid, 2:] (batrack.py:866): stride S_local * 3 floats between
    patches.  Same result as a contiguous copy of it, and no copy is made (ABI field mono_stride)."""
    d = load("c1_rough")
    hp = HipProblem(d)
    P = hp.mono.shape[1]
    local = torch.zeros(1, P, 5, 3, device=hp.mono.device)
    local[:, :, 2, 2:] = hp.mono
    view = local[:, :, 2, 2:]
    assert not view.is_contiguous() and tuple(view.shape) == (1, P, 1)
    a = hp.api_step("weights_pose", 2, False)
    contiguous = hp.mono
    hp.mono = view
    b = hp.api_step("weights_pose", 2, False)
    hp.mono = contiguous
    torch.cuda.synchronize()
    # (two executions differ by the order of their float64 atomics: last-bit noise, not a different prior)
    assert rel(b[0].data.cpu().numpy(), a[0].data.cpu().numpy()) < 1e-6 and rel(b[1].cpu().numpy(), a[1].cpu().numpy()) < 1e-6
    hp.mono = torch.zeros_like(contiguous)
    c = hp.api_step("weights_pose", 2, False)                  # and the prior does matter
    hp.mono = contiguous
    assert rel(c[1].cpu().numpy(), a[1].cpu().numpy()) > 1e-4


def test_expanded_depth_prior_is_materialised():
    """A prior that is one element expanded over all patches (stride 0) holds p_tot values in one float of storage: it must
    not be walked with a stride (round-2 advisor finding) — same result as the contiguous tensor of that value."""
    d = load("c1_rough")
    hp = HipProblem(d)
    P = hp.mono.shape[1]
    keep = hp.mono
    hp.mono = torch.full((1, P, 1), 0.37, device=keep.device)
    a = hp.api_step("weights_pose", 2, False)
    hp.mono = torch.full((1, 1, 1), 0.37, device=keep.device).expand(1, P, 1)
    assert hp.mono.stride(1) == 0
    b = hp.api_step("weights_pose", 2, False)
    hp.mono = keep
    torch.cuda.synchronize()
    assert rel(b[0].data.cpu().numpy(), a[0].data.cpu().numpy()) < 1e-6 and rel(b[1].cpu().numpy(), a[1].cpu().numpy()) < 1e-6


def test_per_track_lmbda_tensor():
    """ba.py:299-300: `lmbda` may be a tensor shaped like C — one damping value per distinct track, ascending patch order."""
    from oracle import refseq
    d = load("c1_rough")
    hp = HipProblem(d)
    m = len(np.unique(d["kk"]))
    lm = np.random.default_rng(3).uniform(1e-4, 0.5, m)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float64))
    ref = refseq.ba_step(t(d["poses"]), t(d["patches"]), t(d["mono"]), t(d["intrinsics"]), t(d["targets3"]), t(d["weights_pose"]),
                         torch.as_tensor(d["ii"]), torch.as_tensor(d["jj"]), torch.as_tensor(d["kk"]),
                         [float(x) for x in d["bounds"]], fixedp=2, lmbda=t(lm.astype(np.float32)))
    Gs, pat = hp.api_step("weights_pose", 2, False, lmbda=torch.as_tensor(lm, dtype=torch.float32, device="cuda:0"))
    torch.cuda.synchronize()
    assert rel(Gs.data[0].cpu().numpy(), ref["poses_out"].numpy()) < STATE_TOL
    assert rel(pat[0, :, :, 0, 0].cpu().numpy(), ref["patches_out"].numpy()) < STATE_TOL
    plain = hp.api_step("weights_pose", 2, False)
    assert rel(pat[0, :, 2, 0, 0].cpu().numpy(), plain[1][0, :, 2, 0, 0].cpu().numpy()) > 1e-4       # and it is not the scalar result
    with pytest.raises(ValueError):
        hp.api_step("weights_pose", 2, False, lmbda=torch.ones(m + 1, device="cuda:0"))


def c3_inputs(seed=0, **kw):
    g = graphgen.make_config("C3", seed=seed, **kw)
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    return dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
                targets3=f(g.targets3), weights=f(g.weights), weights_pose=f(g.weights_pose),
                ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds))


def test_c3_full_size_vs_reference():
    """64 KF / 131,072 edges / 16,384 tracks (BASELINE.json configs[2])."""
    d, gold = c3_inputs(0), load("c3")
    hp = HipProblem(d)
    o = hp.raw_step("weights_pose", 1)
    assert o["status"] == 0
    assert rel(o["dX"].reshape(-1), gold["ps.f64.dX"].reshape(-1)) < DX_TOL
    assert rel(np.diag(o["S_lower"]), gold["ps.f64.S_diag"]) < tol(1e-10, 2e-6)
    assert rel(o["y"], gold["ps.f64.y"]) < tol(1e-10, 2e-6)
    free, act = np.arange(1, 64), np.unique(d["kk"])
    assert update_err(o["poses_out"], gold["ps.f64.poses_out"], d["poses"], free) < UPD_POSE_TOL
    assert update_err(o["patches_out"][:, 2], gold["ps.f64.disp_out"], d["patches"][:, 2], act) < UPD_DISP_TOL
    assert rel(o["poses_out"], gold["ps.f64.poses_out"]) < STATE_TOL
    assert rel(o["patches_out"][:, 2], gold["ps.f64.disp_out"]) < STATE_TOL
    Gs, pat = hp.api_step("weights_pose", 1, False)
    _, pat = hp.api_step("weights", 1, True, poses=Gs, patches=pat)
    assert rel(pat[0, :, 2, 0, 0].cpu().numpy(), gold["so.f64.disp_out"]) < tol(1e-6, 2e-5)


def test_shuffled_edge_order_gives_same_answer():
    """The API accepts any edge order (batrack appends blocks); result must not depend on it."""
    d = c3_inputs(1)
    rng = np.random.default_rng(7)
    p = rng.permutation(len(d["ii"]))
    ds = dict(d)
    for k in ("ii", "jj", "kk", "targets3", "weights", "weights_pose"):
        ds[k] = d[k][p]
    a = HipProblem(d).raw_step("weights_pose", 1)
    b = HipProblem(ds).raw_step("weights_pose", 1)
    assert rel(b["poses_out"], a["poses_out"]) < tol(1e-7, 2e-6)
    assert rel(b["patches_out"], a["patches_out"]) < tol(1e-7, 2e-6)


@pytest.mark.parametrize("M", [256, 576])
def test_real_shape_window_graph_vs_oracle(M):
    """Sliding-window graph laid out by the reference's edge rules: duplicates, 15 free poses.  M = 576: 36 tiles of 16 tracks
    per source frame, i.e. groups of same-camera tiles beyond the 32 that k_pair_finalize adds up in one go."""
    g, fixedp = graphgen.make_window_graph(n_frames=50, M=M, seed=4)
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
             targets3=f(g.targets3), weights=f(g.weights), weights_pose=f(g.weights_pose),
             ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds))
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"],
                         d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=fixedp, want_system=True)
    o = HipProblem(d).raw_step("weights_pose", fixedp)
    assert o["plan"].n == 15 and (F32_EDGE or o["plan"].edge_precision == 8)
    if o["plan"].jacobian_kernel == "k_etile":
        sizes = np.diff(o["plan"].array("sg_ptr"))
        assert sizes.max() <= 32 and ((sizes == 4).any() == (M == 576))        # (36 tiles of a frame = 32 + 4)
    assert rel(np.tril(o["S_lower"]), np.tril(ref["S"])) < tol(1e-10, 2e-6)
    assert rel(o["dX"].reshape(-1), ref["dX"].reshape(-1)) < DX_TOL
    assert update_err(o["poses_out"], ref["poses_out"], d["poses"], np.arange(fixedp, fixedp + 15)) < UPD_POSE_TOL
    assert update_err(o["patches_out"][:, 2], ref["patches_out"][:, 2], d["patches"][:, 2], np.unique(d["kk"])) < UPD_DISP_TOL
    assert rel(o["poses_out"], ref["poses_out"]) < STATE_TOL
    assert rel(o["patches_out"], ref["patches_out"]) < STATE_TOL


def test_convergence_full_ba_c3():
    """Full BA to convergence on the 64-KF graph: pose error to ground truth must drop
    and HIP must track the float64 oracle iterate by iterate."""
    g = graphgen.make_config("C3", seed=2)
    d = c3_inputs(2)
    hp = HipProblem(d)
    Gs, pat = None, None
    po, pa = d["poses"], d["patches"]
    for it in range(4):
        Gs, pat = hp.api_step("weights_pose", 1, False, poses=Gs, patches=pat)
        _, pat = hp.api_step("weights", 1, True, poses=Gs, patches=pat)
        r = oracle.ba_step(po, pa, d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"],
                           d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1)
        r = oracle.ba_step(r["poses_out"], r["patches_out"], d["mono"], d["intrinsics"], d["targets3"], d["weights"],
                           d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1, structure_only=True)
        po, pa = r["poses_out"], r["patches_out"]
    hip_pose = Gs.data[0].cpu().numpy()
    assert rel(hip_pose, po) < tol(2e-6, 5e-5)
    e0 = np.linalg.norm(d["poses"][:, :3] - g.poses_gt[:, :3])
    e1 = np.linalg.norm(hip_pose[:, :3] - g.poses_gt[:, :3])
    assert e1 < 0.5 * e0


def test_cholesky_failure_gives_zero_pose_update():
    """ba.py:9-13: a failed factorisation returns dX = 0 — the depths still move, the poses are only
    re-normalised.  Forced with a negative damping that makes the reduced system indefinite."""
    d = load("c1")
    kw = dict(ep=-1e9)
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"],
                         d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1, want_system=True, **kw)
    assert ref["failed"] and np.all(ref["dX"] == 0)
    o = HipProblem(d).raw_step("weights_pose", 1, **kw)
    assert o["status"] == 1                                   # BT_SOLVE_CHOL_FAILED
    assert np.all(o["dX"] == 0)
    assert rel(o["poses_out"], ref["poses_out"]) < 1e-6       # Exp(0) * G: quaternions re-normalised only
    assert rel(o["patches_out"], ref["patches_out"]) < 1e-5   # dZ = Q w' (ba.py:328 with dX = 0)


def test_status_word_after_structure_only_and_plan_reuse():
    """Plans are reused across the 2*ITER calls of an update(); alternating call kinds must not leak state."""
    d = load("c1_rough")
    hp = HipProblem(d)
    a1 = hp.api_step("weights_pose", 2, False)
    b1 = hp.api_step("weights", 2, True, poses=a1[0], patches=a1[1])
    a2 = hp.api_step("weights_pose", 2, False)                # same inputs as a1, after an SO step on the same plan
    torch.cuda.synchronize()
    assert rel(a2[0].data.cpu().numpy(), a1[0].data.cpu().numpy()) < 1e-6
    assert rel(a2[1].cpu().numpy(), a1[1].cpu().numpy()) < 1e-6
    assert b1[0] is a1[0]


def test_reduced_system_is_reproducible_step_after_step():
    """The accumulators ([S | y], the per-pair sums) are filled with fp64 atomics by all workgroups and
    cleared by their consumers for the next step: 40 consecutive reductions of the same inputs must
    give the same system up to the order of the atomics (measured 2e-11..5e-11 relative: S = B - E C^-1 E^T
    cancels two much larger sums); a lost or doubled contribution, or a stale accumulator, would show at 1e-3."""
    d = c3_inputs(0)
    hp = HipProblem(d)
    o = hp.raw_step("weights_pose", 1)
    st, plan = o["stepper"], o["plan"]
    P = hp.poses[0].contiguous(); pat = hp.patches.reshape(-1, 3).contiguous()
    Pout, pout = torch.empty_like(P), torch.empty_like(pat)
    tg = hp.t3[0]
    args = (P, pat, hp.mono.reshape(-1), hp.intr[0], tg, tg.stride(0), hp.w["weights_pose"][0].contiguous(),
            Pout, pout, hp.bounds, 1e-4, 10.0, 0.05, "huber", False)
    ref = None
    for it in range(40):
        st.step(*args, phase="reduce")
        torch.cuda.synchronize()
        sysv = st.system.cpu().numpy().copy()
        st.step(*args, phase="solve_update")
        torch.cuda.synchronize()
        if ref is None:
            ref = sysv
            D = 6 * plan.n
            assert rel(np.diag(sysv[:D * D].reshape(D, D)), load("c3")["ps.f64.S_diag"]) < tol(1e-10, 2e-6)
        else:
            assert rel(sysv, ref) < 1e-9, it


@pytest.mark.parametrize("seed,N,M,fixedp,far,groups", [(0, 10, 6, 1, 0.3, 1), (1, 17, 5, 2, 0.1, 1), (2, 24, 40, 1, 0.0, 1),
                                                 (3, 30, 30, 3, 0.5, 1), (4, 12, 80, 1, 1.0, 1), (5, 40, 20, 1, 0.05, 1),
                                                 (6, 64, 16, 1, 0.2, 1), (7, 33, 64, 0, 0.0, 4), (8, 25, 64, 1, 0.2, 3),
                                                 (9, 36, 64, 0, 0.3, 6)])
def test_random_covisibility_graphs_vs_oracle(seed, N, M, fixedp, far, groups):
    """Irregular sparsity (loop-closure-like edges, self edges, repeats, shuffled order): whichever solver
    variant and tile layout the plan picks, the step equals the float64 oracle's."""
    g = graphgen.make_random_graph(N, M, seed=seed, far_frac=far, groups=groups)
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
             targets3=f(g.targets3), weights=f(g.weights), weights_pose=f(g.weights_pose),
             ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds))
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"],
                         d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=fixedp, want_system=True)
    o = HipProblem(d).raw_step("weights_pose", fixedp)
    assert o["status"] == 0
    # (float32 per edge: the residuals feed the robust weights; with a few hundred edges there is less averaging than in
    #  the fixtures, so S gets 2e-5 and dX 1e-3 there)
    assert rel(np.tril(o["S_lower"]), np.tril(ref["S"])) < tol(1e-10, 2e-5)
    assert rel(o["dX"].reshape(-1), ref["dX"].reshape(-1)) < tol(DX_TOL, 1e-3)
    assert rel(o["poses_out"], ref["poses_out"]) < tol(STATE_TOL, 1e-5)
    assert rel(o["patches_out"], ref["patches_out"]) < tol(STATE_TOL, 1e-5)


def test_largest_supported_system_vs_oracle():
    """255 free poses (the ABI's limit): a 256-frame banded graph; the reduced system (1530 x 1530) is far too
    large for LDS as double, so this also exercises the solver variant picked for big systems."""
    g = graphgen.make_graph(256, 8, 4, seed=11)
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
             targets3=f(g.targets3), weights=f(g.weights), weights_pose=f(g.weights_pose),
             ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds))
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"],
                         d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1, want_system=True)
    o = HipProblem(d).raw_step("weights_pose", 1)
    assert o["plan"].n == 255 and o["status"] == 0
    assert rel(np.tril(o["S_lower"]), np.tril(ref["S"])) < tol(1e-10, 2e-5)
    # (float32 factor for a system this size, brought back inside the contract by one step of iterative refinement)
    assert rel(o["dX"].reshape(-1), ref["dX"].reshape(-1)) < DX_TOL
    assert rel(o["poses_out"], ref["poses_out"]) < STATE_TOL
    assert rel(o["patches_out"], ref["patches_out"]) < STATE_TOL


@pytest.mark.parametrize("frames", [300, 700])
def test_more_than_255_free_poses_vs_oracle(frames):
    """A band of 300 / 700 keyframes (299 / 699 free poses; ba.py:60-70 has no size clause): beyond the 255 poses the block-sparse
    solvers' tables hold, the plan is `wide` — a dense blocked Cholesky in double in the global workspace (ba_dense.hip).  Float64
    gates, through the C ABI and through BA_rgbd_droid; the packed exchange form (every lower block) round-trips."""
    g = graphgen.make_graph(frames, 16, 6, seed=21)
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
             targets3=f(g.targets3), weights=f(g.weights), weights_pose=f(g.weights_pose),
             ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds))
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"],
                         d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1, want_system=True)
    hp = HipProblem(d)
    o = hp.raw_step("weights_pose", 1)
    n = frames - 1
    assert o["plan"].n == n and o["plan"].nnz_blocks == n * (n + 1) // 2 and o["status"] == 0
    assert rel(np.tril(o["S_lower"]), np.tril(ref["S"])) < tol(1e-10, 2e-5) and rel(o["y"], ref["y"]) < tol(1e-10, 2e-5)
    assert rel(o["dX"].reshape(-1), ref["dX"].reshape(-1)) < DX_TOL
    assert update_err(o["poses_out"], ref["poses_out"], d["poses"], np.arange(1, frames)) < UPD_POSE_TOL
    assert update_err(o["patches_out"][:, 2], ref["patches_out"][:, 2], d["patches"][:, 2], np.unique(g.kk)) < UPD_DISP_TOL
    assert rel(o["poses_out"], ref["poses_out"]) < STATE_TOL and rel(o["patches_out"], ref["patches_out"]) < STATE_TOL
    # the reference's entry point, two chained dual iterations (batrack.py:869-875) against the oracle's
    P, X = hp.api_step("weights_pose", 1, False)
    assert rel(P.data[0].cpu().numpy().astype(np.float64), ref["poses_out"]) < STATE_TOL
    assert rel(X[0, :, :, 0, 0].cpu().numpy().astype(np.float64), ref["patches_out"]) < STATE_TOL
    # pack -> clear -> unpack restores the dense system exactly (the multi-GPU exchange form of a wide plan)
    st = o["stepper"]
    Pg = hp.poses[0].contiguous(); pat = hp.patches.reshape(-1, 3).contiguous()
    Pout, pout = torch.empty_like(Pg), torch.empty_like(pat)
    tg = hp.t3[0]
    args = (Pg, pat, hp.mono.reshape(-1), hp.intr[0], tg, tg.stride(0), hp.w["weights_pose"][0].contiguous(),
            Pout, pout, hp.bounds, 1e-4, 10.0, 0.05, "huber", False)
    st.step(*args, phase="reduce")
    dense = st.system.clone()
    st.step(*args, phase="pack")
    st.system.zero_()
    st.step(*args, phase="unpack")
    D = 6 * n
    low = np.tril(np.ones((D, D), bool))
    a, b = st.system[: D * D].reshape(D, D).cpu().numpy(), dense[: D * D].reshape(D, D).cpu().numpy()
    assert np.array_equal(a[low], b[low]) and torch.equal(st.system[D * D:D * D + D], dense[D * D:D * D + D])
    st.step(*args, phase="solve_update")
    torch.cuda.synchronize()
    assert rel(Pout.cpu().numpy().astype(np.float64), ref["poses_out"]) < STATE_TOL
    if frames > 300:
        return
    # the solver's failure semantics on the dense path (ba.py:9-13, :324-325), as the block-sparse solvers' tests have them
    good = dense                                               # ([S | y] as the reduce phase left it: a solve clears it for the next step)
    st.system.copy_(good)
    st.system[D * D + 3] = float("nan")                        # a NaN in y: the factorisation succeeds, dX is NaN, one retry
    st.step(*args, phase="solve_update")
    torch.cuda.synchronize()
    assert st.status() == 2 and bool(torch.isnan(st.dx).any())
    st.system.copy_(good)
    st.system[0] = -1e9                                        # a negative pivot (beyond the damping): failed factorisation, dX = 0
    st.step(*args, phase="solve_update")
    torch.cuda.synchronize()
    assert st.status() == 1 and bool((st.dx == 0).all())
    assert rel(Pout.cpu().numpy().astype(np.float64), d["poses"]) < 1e-6       # Exp(0) * G
    st.system.copy_(good)
    st.step(*args, phase="solve_update")
    torch.cuda.synchronize()
    assert st.status() == 0 and rel(Pout.cpu().numpy().astype(np.float64), ref["poses_out"]) < STATE_TOL


def test_packed_exchange_form_roundtrip():
    """bt_ba_pack / bt_ba_unpack (the multi-GPU exchange form of [S | y]): packing, clearing the dense system and
    unpacking restores it exactly; the packed buffer is the plan's non-zero blocks in factor order; the
    step completed from the unpacked system equals the plain step."""
    d = c3_inputs(0)
    hp = HipProblem(d)
    o = hp.raw_step("weights_pose", 1)
    st, plan = o["stepper"], o["plan"]
    P = hp.poses[0].contiguous(); pat = hp.patches.reshape(-1, 3).contiguous()
    Pout, pout = torch.empty_like(P), torch.empty_like(pat)
    tg = hp.t3[0]
    args = (P, pat, hp.mono.reshape(-1), hp.intr[0], tg, tg.stride(0), hp.w["weights_pose"][0].contiguous(),
            Pout, pout, hp.bounds, 1e-4, 10.0, 0.05, "huber", False)
    st.step(*args, phase="reduce")
    dense = st.system.clone()
    st.step(*args, phase="pack")
    packed = st.packed.clone()
    st.system.zero_()
    st.step(*args, phase="unpack")
    torch.cuda.synchronize()
    assert torch.equal(st.system, dense)
    A = plan.arrays()
    D = 6 * plan.n
    Sd = dense[:D * D].reshape(D, D).cpu().numpy()
    pk = packed.cpu().numpy()
    assert pk.shape[0] == plan.nnz_blocks * 36 + D < dense.numel() // 6
    for b in (0, 1, plan.nnz_blocks // 2, plan.nnz_blocks - 1):
        src = int(A["blk_src"][b]); rn, cn, tr = src >> 9, (src >> 1) & 255, src & 1
        blk = Sd[6*rn:6*rn + 6, 6*cn:6*cn + 6]
        want = np.tril(blk) if rn == cn else (blk.T if tr else blk)
        assert np.array_equal(pk[36*b:36*b + 36].reshape(6, 6), want)
    assert np.array_equal(pk[plan.nnz_blocks * 36:].reshape(plan.n, 6), dense[D * D:].cpu().numpy().reshape(plan.n, 6)[A["perm"]])
    st.step(*args, phase="solve_update")
    torch.cuda.synchronize()
    assert rel(Pout.cpu().numpy(), o["poses_out"]) < 1e-7 and rel(pout.cpu().numpy(), o["patches_out"]) < 1e-7


@pytest.mark.parametrize("name", ["C3", "band48", "C1"])
def test_solver_gives_the_same_answer_every_time(name):
    """The barrier-free solver orders its waves with LDS flags only; a missing wait would show up as a result
    that depends on timing.  The same reduced system solved 400 times must give the same dX bit for bit."""
    g = {"C3": lambda: graphgen.make_config("C3", seed=0), "band48": lambda: graphgen.make_graph(48, 16, 8, seed=3),
         "C1": lambda: graphgen.make_config("C1", seed=0)}[name]()
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
             targets3=f(g.targets3), weights=f(g.weights), weights_pose=f(g.weights_pose),
             ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds))
    hp = HipProblem(d)
    st = hp.raw_step("weights_pose", 1)["stepper"]
    P = hp.poses[0].contiguous(); pat = hp.patches.reshape(-1, 3).contiguous()
    Pout, pout = torch.empty_like(P), torch.empty_like(pat)
    tg = hp.t3[0]
    args = (P, pat, hp.mono.reshape(-1), hp.intr[0], tg, tg.stride(0), hp.w["weights_pose"][0].contiguous(),
            Pout, pout, hp.bounds, 1e-4, 10.0, 0.05, "huber", False)
    st.step(*args, phase="reduce")
    torch.cuda.synchronize()
    sys0 = st.system.clone()
    ref = None
    for it in range(400):
        st.system.copy_(sys0)
        st.step(*args, phase="solve_update")
        dx = st.dx.clone()
        if ref is None:
            ref = dx
        else:
            assert torch.equal(dx, ref), it
    assert st.status() == 0


@pytest.mark.parametrize("n_hubs", [3, 80])
def test_track_seen_by_many_cameras_vs_oracle(n_hubs):
    """Tracks observed from 40 frames each next to an ordinary banded graph.  A FEW of them (landmarks; at most kFewHubs = 64 per plan)
    sit in no tile — loose tracks, walked in double — so that the plan's tiles keep at most 32 cameras and its per-edge maths float64:
    the float64 gates.  MANY of them stay in tiles of 38+ cameras (large local E blocks, the Schur product spread over many 16x16
    tiles, the depth back-substitution's long camera lists), whose E does not fit LDS as double: float32 per edge, the gates of the
    reference's own precision."""
    g = graphgen.make_graph(48, 8, 4, seed=21)
    rng = np.random.default_rng(5)
    ii, jj, kk = [g.ii], [g.jj], [g.kk]
    hub_tracks = (3, 100, 200) if n_hubs == 3 else tuple(range(2, 2 + 4 * n_hubs, 4))
    for k in hub_tracks:                                      # hub tracks: their source frame to 40 other frames
        tgt = rng.choice(48, size=40, replace=False)
        ii.append(np.full(40, k // 8)); jj.append(tgt); kk.append(np.full(40, k))
    ii, jj, kk = (np.concatenate(a).astype(np.int64) for a in (ii, jj, kk))
    gt = g.patches.copy(); gt[:, 2] = g.disp_gt
    u, v, _ = graphgen.reproject(g.poses_gt, gt, g.intrinsics, ii, jj, kk)
    E = len(kk)
    t3 = np.stack([u + rng.normal(0, 0.5, E), v + rng.normal(0, 0.5, E), g.disp_gt[kk]], 1)
    w = rng.uniform(0.3, 1.0, (E, 2))
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
             targets3=f(t3), weights=f(w), weights_pose=f(w), ii=ii, jj=jj, kk=kk, bounds=np.asarray(g.bounds))
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"],
                         d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1, want_system=True)
    o = HipProblem(d).raw_step("weights_pose", 1)
    assert o["status"] == 0
    loc, kx = o["plan"].array("trk_loc"), o["plan"].array("kx")
    if n_hubs == 3:
        assert sorted(kx[loc < 0]) == sorted(hub_tracks) and o["plan"].max_tile_cams <= 32
        assert F32_EDGE or o["plan"].edge_precision == 8
        assert rel(np.tril(o["S_lower"]), np.tril(ref["S"])) < tol(1e-10, 2e-5) and rel(o["y"], ref["y"]) < tol(1e-10, 2e-5)
        assert rel(o["dX"].reshape(-1), ref["dX"].reshape(-1)) < DX_TOL
        assert update_err(o["poses_out"], ref["poses_out"], d["poses"], np.arange(1, 48)) < UPD_POSE_TOL
        assert update_err(o["patches_out"][:, 2], ref["patches_out"][:, 2], d["patches"][:, 2], np.unique(kk)) < UPD_DISP_TOL
        assert rel(o["poses_out"], ref["poses_out"]) < STATE_TOL and rel(o["patches_out"], ref["patches_out"]) < STATE_TOL
    else:
        assert (loc >= 0).all() and o["plan"].max_tile_cams >= 38
        # E of a tile with 38 cameras does not fit LDS as double: this plan's per-edge maths is float32 (bt_plan_edge_precision)
        assert o["plan"].edge_precision == 4
        assert rel(np.tril(o["S_lower"]), np.tril(ref["S"])) < 2e-5
        assert rel(o["poses_out"], ref["poses_out"]) < 5e-6             # a 47-pose dense system: float32 factor + one refinement step
        assert rel(o["patches_out"], ref["patches_out"]) < 5e-6


@pytest.mark.parametrize("frames,hubs", [(128, 100), (320, 150)])
def test_hub_tracks_seen_by_a_hundred_cameras_vs_oracle(frames, hubs):
    """Tracks observed from 100 / 150 frames (a landmark of a global adjustment; the reference's dense E [n, m, 6] has no camera
    limit, ba.py:268-292) next to an ordinary banded graph: more than 64 free cameras fit no tile — LOOSE tracks (ba_plan.cpp),
    stepped by ba_loose.hip; the second case also has more than 255 free poses (the dense solver).  Float64 gates on the system,
    dX and the state, both step kinds, through the C ABI and through BA_rgbd_droid; repeated observations and a self edge among
    the hub's edges."""
    g = graphgen.make_graph(frames, 8, 4, seed=21)
    rng = np.random.default_rng(5)
    ii, jj, kk = [g.ii], [g.jj], [g.kk]
    hub_tracks = (3, 100, 8 * (frames // 2) + 1)
    for k in hub_tracks:                                      # hub tracks: their source frame to `hubs` other frames
        src = k // 8
        tgt = rng.choice(frames, size=hubs, replace=False)
        tgt = np.concatenate([tgt, tgt[:5], [src]])           # five targets twice, the source frame itself once more
        ii.append(np.full(tgt.size, src)); jj.append(tgt); kk.append(np.full(tgt.size, k))
    ii, jj, kk = (np.concatenate(a).astype(np.int64) for a in (ii, jj, kk))
    p = rng.permutation(ii.size)                              # (the caller's order is not the planner's)
    ii, jj, kk = ii[p], jj[p], kk[p]
    gt = g.patches.copy(); gt[:, 2] = g.disp_gt
    u, v, _ = graphgen.reproject(g.poses_gt, gt, g.intrinsics, ii, jj, kk)
    E = len(kk)
    t3 = np.stack([u + rng.normal(0, 0.5, E), v + rng.normal(0, 0.5, E), g.disp_gt[kk]], 1)
    w = rng.uniform(0.3, 1.0, (E, 2))
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
             targets3=f(t3), weights=f(w), weights_pose=f(w), ii=ii, jj=jj, kk=kk, bounds=np.asarray(g.bounds))
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"],
                         d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1, want_system=True)
    hp = HipProblem(d)
    o = hp.raw_step("weights_pose", 1)
    assert o["status"] == 0 and o["plan"].n == frames - 1
    loc = o["plan"].array("trk_loc")
    kx = o["plan"].array("kx")
    assert sorted(kx[loc < 0]) == sorted(hub_tracks)          # the hubs sit in no tile
    assert F32_EDGE or o["plan"].edge_precision == 8
    assert rel(np.tril(o["S_lower"]), np.tril(ref["S"])) < tol(1e-10, 2e-5) and rel(o["y"], ref["y"]) < tol(1e-10, 2e-5)
    assert rel(o["dX"].reshape(-1), ref["dX"].reshape(-1)) < DX_TOL
    assert update_err(o["poses_out"], ref["poses_out"], d["poses"], np.arange(1, frames)) < UPD_POSE_TOL
    assert update_err(o["patches_out"][:, 2], ref["patches_out"][:, 2], d["patches"][:, 2], np.unique(kk)) < UPD_DISP_TOL
    assert rel(o["poses_out"], ref["poses_out"]) < STATE_TOL and rel(o["patches_out"], ref["patches_out"]) < STATE_TOL
    # structure-only, and both through the reference's entry point
    ref_so = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights"],
                            d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1, structure_only=True)
    o_so = hp.raw_step("weights", 1, True)
    assert update_err(o_so["patches_out"][:, 2], ref_so["patches_out"][:, 2], d["patches"][:, 2], np.unique(kk)) < UPD_DISP_TOL
    P, X = hp.api_step("weights_pose", 1, False)
    assert rel(P.data[0].cpu().numpy().astype(np.float64), ref["poses_out"]) < STATE_TOL
    assert rel(X[0, :, :, 0, 0].cpu().numpy().astype(np.float64), ref["patches_out"]) < STATE_TOL
    P2, X2 = hp.api_step("weights", 1, True)
    assert rel(X2[0, :, :, 0, 0].cpu().numpy().astype(np.float64), ref_so["patches_out"]) < STATE_TOL


def test_print_reports_the_mean_masked_residual(capsys):
    """PRINT=True (ba.py:244-245) prints the mean over the edges of |v r| — r masked by Z > 0.2, |r| < 250 and the bounds — before the
    robust weights; the oracle's per-edge residual is that masked r."""
    d = load("c1_rough")
    hp = HipProblem(d)
    from batrack_amd.backend.ba import BA_rgbd_droid
    from batrack_amd.backend.lietorch import SE3
    capsys.readouterr()
    BA_rgbd_droid(SE3(hp.poses), hp.patches, hp.mono, hp.intr, hp.t3[..., :2], hp.t3[..., 2:], hp.w["weights_pose"], 1e-4,
                  hp.ii, hp.jj, hp.kk, hp.bounds, ep=10.0, PRINT=True, fixedp=1, loss="huber", alpha=0.05)
    torch.cuda.synchronize()
    out = capsys.readouterr().out.strip().splitlines()
    e = oracle.edges(d["poses"], d["patches"], d["intrinsics"], d["targets3"], d["weights_pose"], d["ii"], d["jj"], d["kk"], d["bounds"])
    want = float(np.linalg.norm(e["r"], axis=1).mean())
    assert len(out) == 1 and abs(float(out[0]) - want) < 1e-4 * max(want, 1.0), (out, want)


def test_nan_in_solution_retries_with_larger_damping():
    """ba.py:324-325: a NaN in dX makes the reference solve once more with lm = 1e-3 and keep whatever that gives.
    The path cannot be reached through finite inputs (a NaN in the matrix fails the factorisation first,
    ba.py:9-13), so the right-hand side of the reduced system is poisoned between the two halves of the step."""
    d = load("c1")
    hp = HipProblem(d)
    plan = Plan(hp.ii, hp.jj, hp.kk, hp.poses.shape[1], hp.patches.shape[1], 1)
    st = Stepper(plan, "cuda:0")
    P = hp.poses[0].contiguous()
    pat = hp.patches.reshape(-1, 3).contiguous()
    Pout, pout = torch.empty_like(P), torch.empty_like(pat)
    tg = hp.t3[0]
    args = (P, pat, hp.mono.reshape(-1), hp.intr[0], tg, tg.stride(0), hp.w["weights_pose"][0].contiguous(),
            Pout, pout, hp.bounds, 1e-4, 10.0, 0.05, "huber", False)
    D = 6 * plan.n
    st.step(*args, phase="reduce")
    good = st.system.clone()
    st.step(*args, phase="solve_update")
    torch.cuda.synchronize()
    assert st.status() == 0 and bool(torch.isfinite(Pout).all())
    ref_pose = Pout.clone()
    # same system, y[3] = NaN: the factorisation succeeds, the solution is NaN, the retry cannot cure it
    st.system.copy_(good)
    st.system[D * D + 3] = float("nan")
    st.step(*args, phase="solve_update")
    torch.cuda.synchronize()
    assert st.status() == 2                                    # BT_SOLVE_RETRIED
    assert bool(torch.isnan(st.dx).any())
    assert bool(torch.isfinite(Pout[0]).all()) and bool(torch.isnan(Pout[1:plan.n + 1]).any())   # pose 0 is fixed
    # and the plan is still usable afterwards
    st.system.copy_(good)
    st.step(*args, phase="solve_update")
    torch.cuda.synchronize()
    assert st.status() == 0 and torch.equal(Pout, ref_pose)


@pytest.mark.parametrize("loss,status,n_pose_nan,n_patch_nan", [("huber", 2, 49, 256), ("cauchy", 1, 0, 1)])
def test_nan_target_poisons_the_state_as_in_the_reference(loss, status, n_pose_nan, n_patch_nan):
    """The reference masks an edge by MULTIPLYING with v = 0 (ba.py:233-251), so a NaN target survives as 0 * NaN.
    Run on the C1 graph with one NaN target (tests/golden/make_golden.py machinery, float64) the reference gives:
    huber / trivial — S finite, y NaN, two solves (ba.py:324-325), all 7 free poses NaN (49 numbers) and all 256
    active disparities NaN; cauchy — the weight itself is NaN, S is NaN, the factorisation fails (ba.py:9-13),
    dX = 0, no pose is touched and only that track's disparity is NaN.  The oracle and the kernels multiply too."""
    d = load("c1")
    e0 = int(np.flatnonzero(d["weights_pose"][:, 0] > 0)[5])
    t = d["targets3"].copy()
    t[e0, 0] = np.nan
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], t, d["weights_pose"],
                         d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=1, loss=loss)
    o = HipProblem(dict(d, targets3=t)).raw_step("weights_pose", 1, loss=loss)
    assert o["status"] == status and ref["failed"] == (status == 1)
    for name, n in (("poses_out", n_pose_nan), ("patches_out", n_patch_nan)):
        nr, nh = np.isnan(ref[name]), np.isnan(o[name])
        assert nr.sum() == n and np.array_equal(nr, nh), (name, nr.sum(), nh.sum())
        if (~nr).any():
            assert rel(o[name][~nr], ref[name][~nr]) < STATE_TOL
    if status == 1:
        assert np.all(o["dX"] == 0) and np.isnan(o["patches_out"][d["kk"][e0], 2])


def test_random_graph_sweep():
    """40 random co-visibility graphs of random size, density, forest structure, fixed prefix and call kind (the short
    form of tests/gpu_random_sweep.py): the state stays within max(1e-5, 3x what float32 arithmetic throughout — the
    reference's own precision — loses against float64 on that graph)."""
    rng = np.random.default_rng(77)
    worst = 0.0
    for t in range(40):
        N = int(rng.integers(3, 60)); M = int(rng.integers(2, 60))
        far = float(rng.choice([0.0, 0.05, 0.2, 0.5, 1.0])); groups = int(rng.choice([1, 1, 1, 2, 4]))
        fixedp = int(rng.integers(0, min(4, N - 1)))
        so = bool(rng.random() < 0.15)
        g = graphgen.make_random_graph(N, M, seed=500 + t, far_frac=far, groups=groups)
        f = lambda a: np.asarray(a, np.float32).astype(np.float64)
        d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics), targets3=f(g.targets3),
                 weights=f(g.weights), weights_pose=f(g.weights_pose), ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds))
        args = (d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"], d["ii"], d["jj"], d["kk"], d["bounds"])
        r64 = oracle.ba_step(*args, fixedp=fixedp, structure_only=so)
        r32 = oracle.ba_step(*args, fixedp=fixedp, structure_only=so, dtype=np.float32)
        o = HipProblem(d).raw_step("weights_pose", fixedp, so=so)
        for name in ("poses_out", "patches_out"):
            err, own = rel(o[name], r64[name]), rel(r32[name], r64[name])
            assert err < tol(STATE_TOL, max(1e-5, 3.0 * own)), (t, N, M, far, groups, fixedp, so, name, err, own)
            worst = max(worst, err)
        assert so or o["status"] == 0
    assert worst < tol(STATE_TOL, 1e-4)


def test_device_planned_window_plan_equals_the_host_planned_one():
    """bt_plan_create with DEVICE index tensors plans a sliding-window list on the device (per-track figures, radix sort, the
    pair-major table by kernels; the host lays out tracks, pairs, tiles and the reduced system without reading an edge); with
    host arrays the host analyses the edges.  Same tables either way: the two plans' steps agree to the last bits (the few
    float64 atomics of k_pair_finalize are the only order-dependent sums), and a list with target frames 40 away from the source
    frame (the host's analysis in round 3, the device's 128-bit mask since) gets the same plan."""
    from batrack_amd.plan import Plan, Stepper
    dev = "cuda:0"
    g, fixedp = graphgen.make_window_graph(n_frames=50, M=256, seed=4)
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    poses, patches, mono, intr, t3, w = f32(g.poses), f32(g.patches), f32(g.mono_disp), f32(g.intrinsics), f32(g.targets3), f32(g.weights_pose)
    outs = []
    import force as force_env
    ft = force_env.tokens()
    forced = "kernel" in ft                       # (the suites of test_gpu_jacobian_kernels.py)
    host_plan = ft.get("plan") == "host"
    for on_dev in (True, False):
        idx = [torch.as_tensor(a, device=dev) for a in (g.ii, g.jj, g.kk)] if on_dev else [np.asarray(a) for a in (g.ii, g.jj, g.kk)]
        plan = Plan(*idx, poses.shape[0], patches.shape[0], fixedp)
        if not forced:
            assert plan.jacobian_kernel == "k_etile" and plan.built_on_device == (on_dev and not host_plan)
        st = Stepper(plan, dev)
        Po, Xo = torch.empty_like(poses), torch.empty_like(patches)
        st.step(poses, patches, mono, intr, t3, 3, w, Po, Xo, list(g.bounds), 1e-4, 10.0, 0.05, "huber", False)
        torch.cuda.synchronize()
        outs.append((Po.cpu().numpy().astype(np.float64), Xo.cpu().numpy().astype(np.float64), plan.tiles, plan.m, plan.pairs))
    a, b = outs
    assert a[2:] == b[2:]
    assert rel(a[0], b[0]) < 1e-7 and rel(a[1], b[1]) < 1e-7          # (float32 outputs of float64 sums that differ in their last bits)
    # a window whose tracks reach 40 frames back: outside round 3's 64-frame mask around the source frame, inside round 4's 128
    g2, fp2 = graphgen.make_window_graph(n_frames=60, M=64, seed=5, window=40, removal=45)
    idx = [torch.as_tensor(a, device=dev) for a in (g2.ii, g2.jj, g2.kk)]
    p2 = Plan(*idx, g2.poses.shape[0], g2.patches.shape[0], fp2)
    ref = Plan(np.asarray(g2.ii), np.asarray(g2.jj), np.asarray(g2.kk), g2.poses.shape[0], g2.patches.shape[0], fp2, upload=False)
    assert (forced or host_plan or p2.built_on_device) and (p2.tiles, p2.m, p2.pairs, p2.n) == (ref.tiles, ref.m, ref.pairs, ref.n)
