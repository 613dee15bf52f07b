"""-m gpu: bt_plan_create_shifted — the plan of an edge list that is an earlier one with every frame / patch index moved up
(the caller's sliding window in steady state, batrack.py:189-212) is a device-side copy with shifted numbers.  It must behave
exactly like the plan bt_plan_create builds for the same list."""
import numpy as np
import pytest
import torch

from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def window(seed=4, n_frames=30, M=64):
    g, fixedp = graphgen.make_window_graph(n_frames=n_frames, M=M, seed=seed, n_buf=n_frames + 8)
    return g, fixedp


def step_with(plan, g, ii, jj, kk, poses, patches):
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=DEV)
    st = Stepper(plan, DEV)
    P, X = torch.empty_like(poses), torch.empty_like(patches)
    st.step(poses, patches, f32(g.mono_disp_shifted), f32(g.intrinsics), f32(g.targets3), 3, f32(g.weights_pose), P, X,
            list(g.bounds), 1e-4, 10.0, 0.05, "huber", False)
    torch.cuda.synchronize()
    return P.cpu().numpy(), X.cpu().numpy(), st.status()


@pytest.mark.parametrize("df,M", [(2, 64), (1, 64), (3, 40)])
def test_shifted_plan_equals_a_fresh_one(df, M):
    g, fixedp = window(M=M)
    dk = df * M
    T = lambda a: torch.as_tensor(np.asarray(a, np.int64), device=DEV)
    n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
    ii0, jj0, kk0 = T(g.ii), T(g.jj), T(g.kk)
    src = Plan(ii0, jj0, kk0, n_buf, p_tot, fixedp)
    ii1, jj1, kk1 = ii0 + df, jj0 + df, kk0 + dk
    sh = Plan.shifted(src, ii1, jj1, kk1, n_buf, p_tot, fixedp + df)
    assert sh is not None
    fresh = Plan(ii1, jj1, kk1, n_buf, p_tot, fixedp + df)
    assert sh.info == fresh.info
    assert sh.jacobian_kernel == fresh.jacobian_kernel
    # state shifted the same way: frame f -> f + df, patch p -> p + dk
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=DEV)
    poses = f32(np.roll(g.poses, df, axis=0)); patches = f32(np.roll(g.patches, dk, axis=0))
    g.mono_disp_shifted = np.roll(g.mono_disp, dk, axis=0)
    g.intrinsics = np.roll(g.intrinsics, df, axis=0)
    a = step_with(sh, g, ii1, jj1, kk1, poses, patches)
    b = step_with(fresh, g, ii1, jj1, kk1, poses, patches)
    assert a[2] == b[2] == 0
    rel = lambda x, y: np.linalg.norm(x.astype(np.float64) - y) / np.linalg.norm(y)
    assert rel(a[0], b[0]) < 1e-6 and rel(a[1], b[1]) < 1e-6               # (two executions: the float64 atomics' order)
    assert np.abs(a[0] - np.asarray(poses.cpu())).max() > 0               # and it is a real step
    # a clone can be the source of the next one
    sh2 = Plan.shifted(sh, ii1 + df, jj1 + df, kk1 + dk, n_buf, p_tot, fixedp + 2 * df)
    assert sh2 is not None and sh2.info["fixedp"] == fixedp + 2 * df


def test_one_of_several_candidate_sources_matches():
    """bt_plan_create_shifted_any: the list is compared with every candidate in one pass; the first that is a shifted copy (in
    the order given) is cloned, `which` names it; the clone steps before any host wait for its copies (its launches are ordered
    behind them) exactly like a fresh plan."""
    g, fixedp = window()
    T = lambda a: torch.as_tensor(np.asarray(a, np.int64), device=DEV)
    n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
    ii0, jj0, kk0 = T(g.ii), T(g.jj), T(g.kk)
    src = Plan(ii0, jj0, kk0, n_buf, p_tot, fixedp)
    perm = torch.randperm(ii0.numel(), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    other = Plan(ii0[perm], jj0[perm], kk0[perm], n_buf, p_tot, fixedp)          # the same size, another order: no shifted copy
    later = Plan(ii0 + 1, jj0 + 1, kk0 + 64, n_buf, p_tot, fixedp + 1)            # a shift by one frame of the same list
    ii2, jj2, kk2 = ii0 + 2, jj0 + 2, kk0 + 128
    pl, matched = Plan.shifted_any([other, src, later], ii2, jj2, kk2, n_buf, p_tot, fixedp + 2)
    assert pl is not None and matched is src
    pl2, matched2 = Plan.shifted_any([other, later, src], ii2, jj2, kk2, n_buf, p_tot, fixedp + 2)
    assert pl2 is not None and matched2 is later and pl2.info == pl.info
    assert Plan.shifted_any([other], ii2, jj2, kk2, n_buf, p_tot, fixedp + 2) == (None, None)
    # a candidate that was never uploaded is skipped, and the source is still named by identity (not by a position in a filtered list)
    host_only = Plan(g.ii, g.jj, g.kk, n_buf, p_tot, fixedp, upload=False)
    pl3, matched3 = Plan.shifted_any([host_only, other, src], ii2, jj2, kk2, n_buf, p_tot, fixedp + 2)
    assert pl3 is not None and matched3 is src
    fresh = Plan(ii2, jj2, kk2, n_buf, p_tot, fixedp + 2)
    assert pl.info == fresh.info
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=DEV)
    poses = f32(np.roll(g.poses, 2, axis=0)); patches = f32(np.roll(g.patches, 128, axis=0))
    g.mono_disp_shifted = np.roll(g.mono_disp, 128, axis=0)
    g.intrinsics = np.roll(g.intrinsics, 2, axis=0)
    # straight into a step, several times over fresh clones: the first launch of each is the one ordered behind the copies
    outs = []
    for _ in range(4):
        c, _w = Plan.shifted_any([src], ii2, jj2, kk2, n_buf, p_tot, fixedp + 2)
        outs.append(step_with(c, g, ii2, jj2, kk2, poses, patches))
    b = step_with(fresh, g, ii2, jj2, kk2, poses, patches)
    rel = lambda x, y: np.linalg.norm(x.astype(np.float64) - y) / np.linalg.norm(y)
    for a in outs:
        assert a[2] == b[2] == 0 and rel(a[0], b[0]) < 1e-6 and rel(a[1], b[1]) < 1e-6


def test_speculative_clone_steps_like_a_fresh_plan_and_a_wrong_guess_is_found_out():
    """bt_plan_create_shifted_spec: the clone for the shift fixedp implies, enqueued with no host wait; the verdict of the
    comparison arrives behind it.  Right guess: confirm() is True and the step equals a fresh plan's.  Wrong guess (the list is
    another one of the same length): the plan is still safe to step — every index it holds is inside the buffers — and
    confirm() says False."""
    g, fixedp = window()
    T = lambda a: torch.as_tensor(np.asarray(a, np.int64), device=DEV)
    n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
    M = p_tot // n_buf
    ii0, jj0, kk0 = T(g.ii), T(g.jj), T(g.kk)
    src = Plan(ii0, jj0, kk0, n_buf, p_tot, fixedp)
    ii2, jj2, kk2 = ii0 + 2, jj0 + 2, kk0 + 2 * M
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=DEV)
    poses = f32(np.roll(g.poses, 2, axis=0)); patches = f32(np.roll(g.patches, 2 * M, axis=0))
    g.mono_disp_shifted = np.roll(g.mono_disp, 2 * M, axis=0)
    g.intrinsics = np.roll(g.intrinsics, 2, axis=0)
    fresh = step_with(Plan(ii2, jj2, kk2, n_buf, p_tot, fixedp + 2), g, ii2, jj2, kk2, poses, patches)
    rel = lambda x, y: np.linalg.norm(x.astype(np.float64) - y) / np.linalg.norm(y)
    for _ in range(3):
        pl = Plan.shifted_spec(src, ii2, jj2, kk2, n_buf, p_tot, fixedp + 2)
        assert pl is not None and pl.speculative
        out = step_with(pl, g, ii2, jj2, kk2, poses, patches)              # stepped BEFORE the verdict is asked for
        assert pl.confirm() and not pl.speculative and pl.confirm()
        assert out[2] == fresh[2] == 0 and rel(out[0], fresh[0]) < 1e-6 and rel(out[1], fresh[1]) < 1e-6
    # a list of the same length that is NOT the shifted one (two edges swapped targets): found out, and harmless to have stepped
    jj_bad = jj2.clone()
    jj_bad[[5, 9]] = jj_bad[[9, 5]] + 0
    if bool((jj_bad == jj2).all()):
        jj_bad[5] = jj_bad[5] - 1 if int(jj_bad[5]) > 0 else jj_bad[5] + 1
    pl = Plan.shifted_spec(src, ii2, jj_bad, kk2, n_buf, p_tot, fixedp + 2)
    assert pl is not None
    step_with(pl, g, ii2, jj_bad, kk2, poses, patches)
    assert pl.confirm() is False
    # shifts that are not of the assumed form are not speculated on at all
    assert Plan.shifted_spec(src, ii0, jj0, kk0, n_buf, p_tot, fixedp) is None                 # fixedp did not move
    assert Plan.shifted_spec(src, ii2, jj2, kk2, n_buf, p_tot, fixedp + n_buf) is None          # beyond the pose buffer


def test_a_wrong_guess_inside_BA_rgbd_droid_is_repeated_on_a_proper_plan(monkeypatch):
    """Forced mis-speculation through the drop-in entry point: a cached plan of the right size and fixedp distance whose list is
    NOT the new list's shifted copy.  The call must come back with the result of a properly built plan."""
    from batrack_amd.backend import ba as hip_ba
    from batrack_amd.backend.lietorch import SE3
    g, fixedp = window(seed=9)
    T = lambda a: torch.as_tensor(np.asarray(a, np.int64), device=DEV)
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=DEV)
    n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
    M = p_tot // n_buf
    ii0, jj0, kk0 = T(g.ii), T(g.jj), T(g.kk)
    perm = torch.randperm(ii0.numel(), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    iiB, jjB, kkB = (ii0[perm] + 2).contiguous(), (jj0[perm] + 2).contiguous(), (kk0[perm] + 2 * M).contiguous()     # same edges, shifted AND reordered
    poses0, patches0 = f32(g.poses)[None], f32(g.patches)[None, :, :, None, None]
    poses2, patches2 = f32(np.roll(g.poses, 2, axis=0))[None], f32(np.roll(g.patches, 2 * M, axis=0))[None, :, :, None, None]
    mono0, mono2 = f32(g.mono_disp)[None, :, None], f32(np.roll(g.mono_disp, 2 * M, axis=0))[None, :, None]
    intr0, intr2 = f32(g.intrinsics)[None], f32(np.roll(g.intrinsics, 2, axis=0))[None]
    t3 = f32(g.targets3)[None]
    w = f32(g.weights_pose)[None]
    t3B, wB = t3[:, perm].contiguous(), w[:, perm].contiguous()
    call = lambda P, X, mo, K, tg, ww, a, b, c, fp: hip_ba.BA_rgbd_droid(SE3(P), X, mo, K, tg[..., :2], tg[..., 2:], ww, 1e-4, a, b, c, list(g.bounds),
                                                                          ep=10, fixedp=fp, structure_only=False, loss="huber", alpha=0.05)
    monkeypatch.setenv("BT_PLAN_SPECULATE", "0")
    hip_ba.clear_plan_cache()
    want = call(poses2, patches2, mono2, intr2, t3B, wB, iiB, jjB, kkB, fixedp + 2)
    torch.cuda.synchronize()
    want = (want[0].data.clone(), want[1].clone())
    monkeypatch.setenv("BT_PLAN_SPECULATE", "1")
    hip_ba.clear_plan_cache()
    call(poses0, patches0, mono0, intr0, t3, w, ii0, jj0, kk0, fixedp)        # the cached plan the guess will be made from
    hip_ba._LAST_SHIFT[0] = 2                                                   # "the shift that was right last time"
    tried = {"n": 0}
    real_spec = Plan.shifted_spec.__func__

    def spec(cls, *a, **k):
        r = real_spec(cls, *a, **k)
        tried["n"] += r is not None
        return r
    monkeypatch.setattr(Plan, "shifted_spec", classmethod(spec))
    got = call(poses2, patches2, mono2, intr2, t3B, wB, iiB, jjB, kkB, fixedp + 2)
    torch.cuda.synchronize()
    assert tried["n"] == 1 and hip_ba._LAST_SHIFT[0] is None                    # the guess was made, found wrong, and forgotten
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    assert rel(got[0].data, want[0]) < 1e-6 and rel(got[1], want[1]) < 1e-6
    hip_ba.clear_plan_cache()


def test_lists_that_are_no_shift_are_refused():
    g, fixedp = window()
    T = lambda a: torch.as_tensor(np.asarray(a, np.int64), device=DEV)
    n_buf, p_tot = g.poses.shape[0], g.patches.shape[0]
    ii0, jj0, kk0 = T(g.ii), T(g.jj), T(g.kk)
    src = Plan(ii0, jj0, kk0, n_buf, p_tot, fixedp)
    assert Plan.shifted(src, ii0, jj0, kk0, n_buf, p_tot, fixedp) is None                    # the same list
    assert Plan.shifted(src, ii0 + 1, jj0 + 1, kk0 + 64, n_buf, p_tot, fixedp) is None        # fixedp did not move along
    assert Plan.shifted(src, ii0 + 1, jj0 + 2, kk0 + 64, n_buf, p_tot, fixedp + 1) is None    # ii and jj by different amounts
    jj_bad = jj0 + 1
    jj_bad[7] += 1
    assert Plan.shifted(src, ii0 + 1, jj_bad, kk0 + 64, n_buf, p_tot, fixedp + 1) is None     # one edge differs
    assert Plan.shifted(src, ii0[:-1] + 1, jj0[:-1] + 1, kk0[:-1] + 64, n_buf, p_tot, fixedp + 1) is None   # another length
    with pytest.raises(RuntimeError):
        Plan.shifted(src, ii0 + 10000, jj0 + 10000, kk0, n_buf, p_tot, fixedp + 10000)       # out of range: an error, like bt_plan_create


def test_sequence_replay_uses_shifted_plans_and_gives_the_same_trajectory(monkeypatch):
    from batrack_amd.backend import ba as hip_ba
    from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BT_PLAN_SHIFT", mode)
        hip_ba.clear_plan_cache()
        made = {"shifted": 0, "built": 0}
        real_shifted, real_spec, real_init = Plan.shifted_any.__func__, Plan.shifted_spec.__func__, Plan.__init__

        def shifted(cls, *a, **k):
            r = real_shifted(cls, *a, **k)
            made["shifted"] += r[0] is not None
            return r

        def spec(cls, *a, **k):                                    # (steady state: the clone is made on the ASSUMPTION of the shift)
            r = real_spec(cls, *a, **k)
            made["shifted"] += r is not None
            return r
        monkeypatch.setattr(Plan, "shifted_spec", classmethod(spec))
        real_bind = Plan.bind

        def bind(self, *a, **k):                                   # (... or was made AHEAD, during the update() before, and only meets its list here)
            r = real_bind(self, *a, **k)
            made["shifted"] += bool(r)
            made["bound"] = made.get("bound", 0) + bool(r)
            return r
        monkeypatch.setattr(Plan, "bind", bind)

        def init(self, *a, **k):
            made["built"] += 1
            return real_init(self, *a, **k)
        monkeypatch.setattr(Plan, "shifted_any", classmethod(shifted))
        monkeypatch.setattr(Plan, "__init__", init)
        obs = SyntheticObservations(n_frames=60, M=64, seed=6)
        trk = WindowedBA(obs, hip_ba.BA_rgbd_droid, SlamConfig(PATCHES_PER_FRAME=64, BUFFER_SIZE=64), device=DEV)
        runs[mode] = (trk.run(), dict(made))
        monkeypatch.undo()
    hip_ba.clear_plan_cache()
    assert runs["0"][1]["shifted"] == 0
    assert runs["1"][1]["shifted"] >= 20 and runs["1"][1]["built"] <= runs["0"][1]["built"] - 20     # the steady state (from frame ~34) is served by shifts
    assert runs["1"][1].get("bound", 0) >= 15                      # most of them made ahead (Plan.preshift), bound by one comparison kernel
    assert np.abs(runs["1"][0] - runs["0"][0]).max() < 2e-5


def test_a_clone_made_ahead_of_its_list():
    """bt_plan_preshift / bt_plan_spec_bind: the clone for the next window is enqueued before its list exists; it cannot be stepped
    unbound, binds to the list it was made for (verdict good: same result as a plan built from scratch), says BT_NO_MATCH — through
    `confirm()` — for a list that is not the shifted one, and refuses a list of another size or fixedp without a kernel."""
    from batrack_amd import graphgen
    g = graphgen.make_graph(12, 64, 6, seed=4, n_buf=24)
    dev = DEV
    T = lambda a: torch.as_tensor(a, device=dev)
    n_buf, p_tot, fixedp = 24, 24 * 64, 2
    ii0, jj0, kk0 = T(g.ii), T(g.jj), T(g.kk)
    src = Plan(ii0, jj0, kk0, n_buf, p_tot, fixedp)
    pre = Plan.preshift(src, 3)
    assert pre is not None and pre.info["fixedp"] == fixedp + 3
    st = Stepper(pre, dev)
    poses = torch.zeros(n_buf, 7, device=dev); poses[:, 6] = 1
    f = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    poses[3:15] = f(g.poses[:12])
    pat = torch.zeros(p_tot, 3, device=dev); pat[3 * 64:15 * 64] = f(g.patches[:12 * 64]); pat[:, 2].clamp_(min=0.1)
    mono = pat[:, 2].clone()
    intr = f(g.intrinsics[:1]).repeat(n_buf, 1)
    t3, w = f(g.targets3), f(g.weights_pose)
    args = lambda P, X: (poses, pat, mono, intr, t3, 3, w, P, X, list(g.bounds), 1e-4, 10.0, 0.05, "huber", False)
    with pytest.raises(RuntimeError):
        st.step(*args(torch.empty_like(poses), torch.empty_like(pat)))          # unbound: BT_EINVAL
    ii1, jj1, kk1 = ii0 + 3, jj0 + 3, kk0 + 3 * 64
    assert not pre.bind(ii1[:-1].contiguous(), jj1[:-1].contiguous(), kk1[:-1].contiguous(), n_buf, p_tot, fixedp + 3)   # another size
    assert not pre.bind(ii1, jj1, kk1, n_buf, p_tot, fixedp + 2)                                                        # another fixedp
    assert pre.bind(ii1, jj1, kk1, n_buf, p_tot, fixedp + 3)
    P1, X1 = torch.empty_like(poses), torch.empty_like(pat)
    st.step(*args(P1, X1))
    assert pre.confirm()
    ref = Stepper(Plan(ii1, jj1, kk1, n_buf, p_tot, fixedp + 3), dev)
    P2, X2 = torch.empty_like(poses), torch.empty_like(pat)
    ref.step(*args(P2, X2))
    torch.cuda.synchronize()
    assert (P1 - P2).abs().max() < 1e-6 and (X1 - X2).abs().max() < 1e-6
    # the same with the GPU far behind — tens of milliseconds of the caller's earlier work in front of the comparison (in the pipeline:
    # the tracker pass): the verdict is waited for (a spin, then a yielding poll), not guessed
    pre_b = Plan.preshift(src, 3)
    st_b = Stepper(pre_b, dev)
    big = torch.randn(4096, 4096, device=dev)
    for _ in range(40):
        big = (big @ big).clamp_(-1, 1)
    assert pre_b.bind(ii1, jj1, kk1, n_buf, p_tot, fixedp + 3)
    P3, X3 = torch.empty_like(poses), torch.empty_like(pat)
    st_b.step(*args(P3, X3))
    assert pre_b.confirm()
    torch.cuda.synchronize()
    assert (P3 - P2).abs().max() < 1e-6 and (X3 - X2).abs().max() < 1e-6
    # a list that is NOT the shifted one (one edge points elsewhere): bound, stepped, and then told so
    pre2 = Plan.preshift(src, 3)
    jj_bad = jj1.clone(); jj_bad[5] = jj_bad[5] + 1
    assert pre2.bind(ii1, jj_bad, kk1, n_buf, p_tot, fixedp + 3)
    assert pre2.confirm() is False
    # never bound: no plan of anything
    pre3 = Plan.preshift(src, 3)
    assert pre3.confirm() is False
    assert Plan.preshift(src, 30) is None                                        # the shift does not fit the buffers
