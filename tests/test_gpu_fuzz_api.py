"""The random graphs of tests/test_gpu_fuzz.py through the reference's ENTRY POINT, BA_rgbd_droid (ba.py:217-339), with the tensors laid
out every way its callers lay them out: the 2-D target as a stride-3 view of the [E, 3] buffer or on its own, the depth prior as a
column of a wider tensor, the weights as a view, patches of size 1 or 3, lmbda as a float, a 0-dim / 1-element tensor or one value per
track, random ep / alpha / bounds, and two chained calls (pose+structure, then structure-only on what it returned — the pattern of
batrack.py:871-880).  Checked against the C oracle (scalar lmbda) or the operator-sequence oracle (per-track lmbda).

As a script: python tests/test_gpu_fuzz_api.py [first_seed] [count]"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import oracle  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def check(seed):
    from batrack_amd.backend.ba import BA_rgbd_droid
    from batrack_amd.backend.lietorch import SE3
    from gpu_util import rel
    from oracle import refseq
    rng = np.random.default_rng(seed + 500000)
    d, fixedp, so, loss, wkey, desc = F.draw(seed)
    if d["ii"].size > 60000:                                   # (the operator-sequence oracle builds dense [n, m, 6] tensors)
        sub = rng.choice(d["ii"].size, 60000, replace=False)
        for k in ("ii", "jj", "kk", "targets3", "weights", "weights_pose"):
            d[k] = d[k][sub]
    E, P, N = d["ii"].size, d["patches"].shape[0], d["poses"].shape[0]
    n_all = int(max(d["ii"].max(), d["jj"].max())) + 1
    fixedp = min(fixedp, n_all)
    t32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=DEV)
    # --- the caller's layouts
    t3 = t32(d["targets3"])[None]                                                   # [1, E, 3]
    lay_t = int(rng.integers(0, 2))
    tg2 = t3[..., :2] if lay_t == 0 else t3[..., :2].contiguous()
    tgd = t3[..., 2:]
    lay_m = int(rng.integers(0, 2))
    mono = t32(d["mono"])[None, :, None] if lay_m == 0 else torch.stack([t32(d["mono"]), t32(d["mono"]) * 0 - 1], -1)[None][..., :1]
    lay_w = int(rng.integers(0, 2))
    w = t32(d[wkey])[None] if lay_w == 0 else torch.cat([t32(d[wkey]), t32(d[wkey])[:, :1] * 0 + 7], -1)[None][..., :2]
    psz = int(rng.choice([1, 1, 3]))
    patches = t32(d["patches"])[None, :, :, None, None].expand(1, P, 3, psz, psz).contiguous()
    m = len(np.unique(d["kk"]))
    lay_l = int(rng.integers(0, 4))
    lm_np = rng.uniform(1e-4, 0.3, m).astype(np.float32) if lay_l == 3 else np.float32(rng.choice([1e-4, 1e-3, 0.05]))
    lmbda = (float(lm_np) if lay_l == 0 else torch.tensor(float(lm_np), device=DEV) if lay_l == 1 else
             torch.tensor([float(lm_np)], device=DEV) if lay_l == 2 else torch.as_tensor(lm_np, device=DEV))
    ep, alpha = float(rng.choice([0.1, 10.0, 100.0])), float(rng.choice([0.05, 0.5]))
    b = np.asarray(d["bounds"], np.float64)
    if rng.random() < 0.5:
        b = b + np.array([8.0, 5.0, -11.0, -6.0])
    bounds = [float(v) for v in b]
    ii, jj, kk = (torch.as_tensor(d[k], device=DEV) for k in ("ii", "jj", "kk"))
    desc += f" | E {E} targets {'view' if lay_t == 0 else 'own'} prior {'own' if lay_m == 0 else 'column'} weights {'own' if lay_w == 0 else 'view'} p {psz} " \
            f"lmbda {['float', '0-dim', '[1]', 'per track'][lay_l]} ep {ep} alpha {alpha}"
    # --- HIP: pose+structure (or what the draw says), then structure-only with the other weights on the result
    Gs = SE3(t32(d["poses"])[None])
    G1, p1 = BA_rgbd_droid(Gs, patches, mono, t32(d["intrinsics"])[None], tg2, tgd, w, lmbda, ii, jj, kk, bounds, ep=ep, fixedp=fixedp,
                           structure_only=so, loss=loss, alpha=alpha)
    G2, p2 = BA_rgbd_droid(G1, p1, mono, t32(d["intrinsics"])[None], tg2, tgd, w, lmbda, ii, jj, kk, bounds, ep=ep, fixedp=fixedp,
                           structure_only=True, loss=loss, alpha=alpha)
    torch.cuda.synchronize()
    assert G2 is G1 and (not so or G1 is Gs), desc                                   # ba.py:337-339: the same object back on structure-only
    assert p2.shape == patches.shape, desc
    # --- oracle, the same two calls on float32 state in between
    f64 = lambda a: np.asarray(a, np.float32).astype(np.float64)

    def step(poses, pats, so_, dtype=np.float64):
        if lay_l == 3:
            td = torch.float64 if dtype == np.float64 else torch.float32
            t = lambda a: torch.as_tensor(np.asarray(a, dtype))
            r = refseq.ba_step(t(poses), t(pats), t(d["mono"]), t(d["intrinsics"]), t(d["targets3"]), t(d[wkey]),
                               torch.as_tensor(d["ii"]), torch.as_tensor(d["jj"]), torch.as_tensor(d["kk"]), bounds, fixedp=fixedp,
                               structure_only=so_, loss=loss, lmbda=torch.as_tensor(lm_np).to(td), ep=ep, alpha=alpha)
            return r["poses_out"].numpy().astype(np.float64), r["patches_out"].numpy().astype(np.float64)
        r = oracle.ba_step(poses, pats, d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"], b,
                           lmbda=float(lm_np), ep=ep, alpha=alpha, fixedp=fixedp, structure_only=so_, loss=loss, dtype=dtype)
        return np.asarray(r["poses_out"], np.float64), np.asarray(r["patches_out"], np.float64)

    out = {}
    for dt in (np.float64, np.float32):
        po, pa = step(d["poses"], d["patches"], so, dt)
        po2, pa2 = step(f64(po), f64(pa), True, dt)
        out[dt] = (f64(po), f64(pa2))
    (rp, rx), (hp_, hx) = out[np.float64], out[np.float32]
    hard_p, hard_x = rel(hp_, rp), rel(hx, rx)
    got_p = G2.data[0].cpu().numpy()
    got_x = p2[0, :, :, psz // 2, psz // 2].cpu().numpy()
    ep_, ex_ = rel(got_p, rp), rel(got_x, rx)
    desc += f" | poses {ep_:.2e} patches {ex_:.2e} (float32 oracle {hard_p:.2e} {hard_x:.2e})"
    assert np.isfinite(got_p).all() and np.isfinite(got_x).all(), desc
    assert ep_ < max(8e-6, 2 * hard_p) and ex_ < max(8e-6, 2 * hard_x), desc
    if psz > 1:                                                  # the whole plane moved with its centre (ba.py:332-334)
        assert bool((p2[:, :, 2] == p2[:, :, 2, :1, :1]).all()), desc
        assert bool((p2[:, :, :2] == patches[:, :, :2]).all()), desc
    return desc


@pytest.mark.parametrize("seed", range(7600, 7640))
def test_random_call_layouts_vs_oracle(seed):
    check(seed)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad = 0
    for s in range(first, first + count):
        try:
            print("ok  ", check(s), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", e, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            import traceback
            print("ERR ", s, type(e).__name__, e, traceback.format_exc().splitlines()[-4:], flush=True)
    print(f"{count} seeds from {first}: {bad} failed")
    sys.exit(1 if bad else 0)
