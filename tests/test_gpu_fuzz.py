"""Randomised graphs against the oracle: shapes no hand-written case has — ragged tracks next to hubs, unused patches, pose slots
nobody observes, repeated and self edges, zero weights, targets off the image, any number of fixed poses, either loss, either step
kind, the caller's order shuffled.  Every case is one seed; a failure prints it.

As a script (`python tests/test_gpu_fuzz.py [first_seed] [count] [big]`) it walks as many seeds as asked for; `big` draws graphs of
>= 2048 tiles (the mixed-precision kernels) as well.  What the reference does with such lists: ba.py:236-334 — nothing there depends
on the shape of the graph, so nothing here may."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import oracle  # noqa: E402
from batrack_amd import graphgen  # noqa: E402

import force as _force  # noqa: E402  (tests/force.py)

pytestmark = pytest.mark.gpu
FORCED_F32 = _force.f32_edges()
# lmbda and alpha as the step receives them: float32 (include/batrack_ba.h: bt_ba_args; the reference adds its Python floats to float32
# tensors, ba.py:302-303, i.e. rounds them the same way).  The float64 oracle is handed THESE values: with the doubles 1e-4 / 0.05 a
# graph whose tracks lean on the depth prior shows alpha's rounding (1.5e-8) in [S | y] — 9.0e-9 / 6.4e-9 on seeds 30971 / 30470, to
# six digits what the oracle itself gives between the two alphas — and that is not an error of the step.
ABI_SCALARS = dict(lmbda=float(np.float32(1e-4)), alpha=float(np.float32(0.05)))


def draw(seed, big=False):
    """One random problem: (input dict, fixedp, structure_only, loss, description)."""
    rng = np.random.default_rng(seed)
    if big:
        N = int(rng.choice([24, 64, 96]))
        M = int(rng.choice([2048, 4096, 8192])) * 64 // N * 2
        deg_lo, deg_hi = int(rng.integers(1, 4)), int(rng.integers(4, 9))
    else:
        N = int(rng.choice([3, 4, 6, 9, 14, 24, 40, 64, 100, 180, 290, 420]))
        M = int(rng.choice([1, 2, 3, 7, 16, 40, 96]))
        while N * M > 24000:
            M = max(1, M // 2)
        deg_lo, deg_hi = int(rng.integers(0, 3)), int(rng.integers(3, 12))
        if rng.random() < 0.2:                                             # deep edge lists: a sliding window's tracks (the pair-major kernel)
            deg_lo, deg_hi = int(rng.integers(20, 30)), int(rng.integers(30, 70))
            while N * M * deg_hi > 400000:
                M = max(1, M // 2)
    n_buf = N + int(rng.choice([0, 0, 1, 9]))
    g = graphgen.make_graph(N, M, 1, seed=seed, n_buf=n_buf, pose_noise=float(rng.choice([0.003, 0.01, 0.03])),
                            disp_noise=float(rng.choice([0.02, 0.1, 0.25])))
    na = N * M
    deg = rng.integers(deg_lo, deg_hi + 1, na)
    if rng.random() < 0.5:
        deg[rng.random(na) < 0.15] = 0                                     # patches no edge names
    kk = np.repeat(np.arange(na, dtype=np.int64), deg)
    ii = kk // M
    E = kk.size
    width = int(rng.choice([1, 3, 6, 12]))
    near = ii + rng.integers(-width, width + 1, E)
    far = rng.integers(0, N, E)
    jj = np.where(rng.random(E) < float(rng.choice([0.0, 0.1, 0.4])), far, near)
    jj = np.clip(jj, 0, N - 1).astype(np.int64)
    if rng.random() < 0.3:                                                 # the last frames see nothing and are seen by nothing
        cut = N - int(rng.integers(1, max(2, N // 4)))
        keep = (ii < cut) & (jj < cut)
        ii, jj, kk = ii[keep], jj[keep], kk[keep]
    hubs = 0
    if not big and N >= 24 and rng.random() < 0.35:                        # tracks seen from most of the trajectory
        hubs = int(rng.integers(1, 4))
        for k in rng.choice(na, hubs, replace=False):
            cnt = int(rng.integers(N // 2, N + 1))
            tgt = rng.choice(N, cnt, replace=False)
            ii = np.concatenate([ii, np.full(cnt, k // M)]); jj = np.concatenate([jj, tgt]); kk = np.concatenate([kk, np.full(cnt, k)])
    if rng.random() < 0.5 and ii.size:                                     # some edges twice
        dup = rng.choice(ii.size, max(1, ii.size // 20))
        ii, jj, kk = (np.concatenate([a, a[dup]]) for a in (ii, jj, kk))
    if ii.size == 0:
        ii, jj, kk = np.array([0], np.int64), np.array([min(1, N - 1)], np.int64), np.array([0], np.int64)
    p = rng.permutation(ii.size)
    ii, jj, kk = ii[p].astype(np.int64), jj[p].astype(np.int64), kk[p].astype(np.int64)
    E = ii.size
    gt = g.patches.copy(); gt[:, 2] = g.disp_gt
    u, v, _ = graphgen.reproject(g.poses_gt, gt, g.intrinsics, ii, jj, kk)
    px = float(rng.choice([0.2, 0.5, 2.0]))
    t3 = np.stack([u + rng.normal(0, px, E), v + rng.normal(0, px, E), g.disp_gt[kk]], 1)
    wild = rng.random(E) < float(rng.choice([0.0, 0.02, 0.1]))            # outliers: some far beyond the robust threshold,
    t3[wild, :2] += rng.normal(0, 300.0, (int(wild.sum()), 2))             # some off the image
    w = rng.uniform(0.05, 1.0, (E, 2))
    w[rng.random(E) < float(rng.choice([0.0, 0.05]))] = 0.0
    dyn = rng.random(na) < float(rng.choice([0.0, 0.3, 0.7]))
    wp = w * (~dyn[kk])[:, None]
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics), targets3=f(t3),
             weights=f(w), weights_pose=f(wp), ii=ii, jj=jj, kk=kk, bounds=np.asarray(g.bounds, np.float64))
    n_all = int(max(ii.max(), jj.max())) + 1
    fixedp = int(rng.choice([1, 1, 1, 2, 3, max(1, n_all // 2), max(1, n_all - 1), n_all]))
    so = bool(rng.random() < 0.25)
    loss = str(rng.choice(["huber", "huber", "cauchy"]))
    wkey = str(rng.choice(["weights_pose", "weights"]))
    desc = (f"seed {seed}: N={N} M={M} n_buf={n_buf} E={E} deg {deg_lo}..{deg_hi} width {width} hubs {hubs} fixedp {fixedp} "
            f"{'so' if so else 'ps'} {loss} {wkey}")
    return d, fixedp, so, loss, wkey, desc


def check(seed, big=False):
    """Runs one seed; returns (description, dict of measured errors).  Raises AssertionError on a miss."""
    from gpu_util import HipProblem, rel, update_err
    d, fixedp, so, loss, wkey, desc = draw(seed, big)
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"],
                         d["bounds"], fixedp=fixedp, structure_only=so, loss=loss, want_system=True, **ABI_SCALARS)
    hp = HipProblem(d)
    o = hp.raw_step(wkey, fixedp, so=so, loss=loss)
    plan = o["plan"]
    f32 = plan.edge_precision != 8
    desc += f" | n={plan.n} tiles={plan.tiles} kernel {plan.jacobian_kernel} f{'32' if f32 else '64'}"
    errs = {}
    solved = (not so) and plan.n > 0 and "S" in ref
    # The oracle's own float32 run (the reference's precision) says how hard the case is: its error against the float64 run scales the
    # gates of a float32 edge pass on [S | y], and of every path on the state — the update of an ill-conditioned random system amplifies
    # the float32 rounding of dX by its condition number.
    ref32 = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"],
                           d["bounds"], fixedp=fixedp, structure_only=so, loss=loss, dtype=np.float32, want_system=True, **ABI_SCALARS)
    if solved:
        # No valid edge reaches a free pose, or only SELF edges do (ii == jj: Gij is the identity, Ji = -Jj, and the edge's blocks cancel
        # to zero in exact arithmetic — seed 32333: the float64 oracle keeps 2.9e-11 of rounding there, its float32 run 2e-25, the HIP
        # step, which knows a self edge's Ad to be exactly I, something else again): S is rounding noise against the size of the edges' own
        # blocks, and two roundings of zero have no relative error.
        eo = oracle.edges(d["poses"], d["patches"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"], d["bounds"], loss=loss)
        blk = float((eo["W"][:, :, None] * np.maximum(eo["Ji"] ** 2, eo["Jj"] ** 2)).max()) if d["ii"].size else 0.0
        if np.abs(ref["S"]).max() <= 1e-12 * blk or np.abs(ref["S"]).max() < 1e-20:
            solved = False
    if solved:
        errs["S"] = rel(np.tril(o["S_lower"]), np.tril(ref["S"])); errs["y"] = rel(o["y"], ref["y"])
        errs["ref32_S"] = rel(np.tril(ref32["S"]), np.tril(ref["S"])); errs["ref32_y"] = rel(ref32["y"], ref["y"])
        # (float64 per edge: 1e-9, or — where the Schur complement cancels most of B — a thousandth of what float32 loses there)
        # (a float32 edge pass FORCED onto graphs of a few tiles — BT_FORCE, measurement only — sums in other orders than the oracle's
        #  float32 run: up to three times its error on systems of a handful of edges)
        k32 = 10.0 if FORCED_F32 else 2.0
        # (2e-5: the gate of the hub-track case of tests/test_gpu_parity.py — tiles of 40 cameras sum their float32 rows in long chains)
        ok_S = max(2e-5, k32 * errs["ref32_S"]) if f32 else max(1e-9, 1e-3 * errs["ref32_S"])
        ok_y = max(2e-5, k32 * errs["ref32_y"]) if f32 else max(1e-9, 1e-3 * errs["ref32_y"])
        assert errs["S"] < ok_S and errs["y"] < ok_y, (desc, errs)
        assert (o["status"] != 0) == ref["failed"] or o["status"] in (0, 1), (desc, o["status"], ref["failed"])
    hard_p = rel(ref32["poses_out"], ref["poses_out"]); hard_d = rel(ref32["patches_out"], ref["patches_out"])
    errs["poses"] = rel(o["poses_out"], ref["poses_out"]); errs["patches"] = rel(o["patches_out"], ref["patches_out"])
    errs["ref32_poses"], errs["ref32_patches"] = hard_p, hard_d
    floor = 8e-6 if f32 else 3e-7
    assert np.isfinite(o["poses_out"]).all() and np.isfinite(o["patches_out"]).all(), desc
    k32 = 10.0 if FORCED_F32 else 2.0
    assert errs["poses"] < max(floor, k32 * hard_p) and errs["patches"] < max(floor, k32 * hard_d), (desc, errs)
    if solved and not ref["failed"]:
        errs["upd_pose"] = update_err(o["poses_out"], ref["poses_out"], d["poses"])
        errs["upd_disp"] = update_err(o["patches_out"][:, 2], ref["patches_out"][:, 2], d["patches"][:, 2])
    # The same plan and workspace again: the OTHER step kind, then the first kind once more — every step must leave the workspace as
    # it found it (accumulators, counters, per-tile records), whatever ran before.
    import torch
    st = o["stepper"]
    P = hp.poses[0].contiguous(); pat = hp.patches.reshape(-1, 3).contiguous(); tg = hp.t3[0]
    for so2 in ((not so) or plan.n == 0, so or plan.n == 0):
        pout = torch.empty_like(pat)
        Pout = P if so2 else torch.empty_like(P)
        st.step(P, pat, hp.mono.reshape(-1), hp.intr[0], tg, tg.stride(0), hp.w[wkey][0].contiguous(), Pout, pout, hp.bounds,
                1e-4, 10.0, 0.05, loss, so2)
        torch.cuda.synchronize()
        r2 = ref if so2 == (so or plan.n == 0) else oracle.ba_step(
            d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"], d["bounds"],
            fixedp=fixedp, structure_only=so2, loss=loss, **ABI_SCALARS)
        r32 = ref32 if so2 == (so or plan.n == 0) else oracle.ba_step(
            d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"], d["bounds"],
            fixedp=fixedp, structure_only=so2, loss=loss, dtype=np.float32, **ABI_SCALARS)
        hp2, hd2 = rel(r32["poses_out"], r2["poses_out"]), rel(r32["patches_out"], r2["patches_out"])
        ep2, ed2 = rel(Pout.cpu().numpy(), r2["poses_out"]), rel(pout.cpu().numpy(), r2["patches_out"])
        assert ep2 < max(floor, k32 * hp2) and ed2 < max(floor, k32 * hd2), (desc, "again, structure-only" if so2 else "again, poses", ep2, ed2, hp2, hd2)
        assert st.system.numel() == 0 or float(st.system.abs().max()) == 0.0, (desc, "accumulators not clear")
    return desc, errs


@pytest.mark.parametrize("seed", range(7000, 7060))
def test_random_graph_vs_oracle(seed):
    check(seed)


@pytest.mark.parametrize("seed", range(7500, 7504))
def test_random_large_graph_vs_oracle(seed):
    check(seed, big=True)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    bad = 0
    worst = {}
    for s in range(first, first + count):
        try:
            desc, errs = check(s, big)
            for k, v in errs.items():
                if not k.startswith("ref32") and v > worst.get(k, (0.0, ""))[0]:
                    worst[k] = (v, desc)
            print("ok  ", desc, " ".join(f"{k}={v:.2e}" for k, v in errs.items()), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", e, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("ERR ", s, type(e).__name__, e, flush=True)
    print(f"{count} seeds from {first}: {bad} failed")
    for k, (v, desc) in worst.items():
        print(f"worst {k}: {v:.3e}  ({desc})")
    sys.exit(1 if bad else 0)
