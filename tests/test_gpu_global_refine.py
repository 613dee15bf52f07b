"""-m gpu: the dense global-alignment losses on the HIP kernels against the reference's vectors
(tests/golden/ga_small.npz, produced by the unmodified refine_net.py) and against the float64 oracle on a larger case.
Tolerances: float32 kernels vs float64 reference 2e-6 on the scaled depth, 1e-5 on the losses (sums of ~1e4..1e6 float32
terms, float64 across workgroups); the float16 depth-residual variant 2e-3 (float16 has 11 bits)."""
import os

import numpy as np
import pytest
import torch

from oracle import ga_losses as ga

pytestmark = pytest.mark.gpu
D = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ga_small.npz")))


def build(d, half=False, **kw):
    """`scale_smoothness_weight` defaults to 0 here: the round-2 fixtures (ga_small.npz) hold forward()'s total without the
    smoothness term; the tests of the full total pass the reference's settings explicitly."""
    from batrack_amd.global_refine import RefineLosses
    t = lambda k: torch.as_tensor(np.asarray(d[k]), device="cuda:0")
    kw.setdefault("scale_smoothness_weight", 0.0)
    return RefineLosses(t("trajs_2d"), t("trajs_disp"), t("trajs_disp_mono"), t("trajs_vis"), t("trajs_static"), t("jj"),
                        t("intrinsics"), t("grid_query_frames"), t("trajs_scales"), t("frame_scales_"), t("frame_shifts"), t("pose"),
                        int(d["H"]), int(d["W"]), float(d["pw_break"]), half_disp=half, **kw)


def test_losses_match_the_reference_vectors():
    net = build(D)
    ms = net.get_frame_scaled_depth().cpu().numpy()
    assert np.linalg.norm(ms - D["f64.mono_scaled"]) / np.linalg.norm(D["f64.mono_scaled"]) < 2e-6
    l = net.losses().cpu().numpy()[:3]
    for got, key in zip(l, ("f64.loss_spatial", "f64.loss_rigid", "f64.loss_pts3d")):
        assert abs(got / D[key] - 1) < 1e-5, (key, got, D[key])
    assert abs(float(net.forward(0.5)) / D["f64.total_alpha05"] - 1) < 1e-5
    assert abs(float(net.forward(0.0)) / D["f64.loss_spatial"] - 1) < 1e-5


def make_case(T, N, S, seed):
    rng = np.random.default_rng(seed)
    H, W, gh, gw = 436, 1024, 4, 4
    jj = np.arange(T)[:, None] + np.arange(S)[None] - S // 2
    disp = rng.uniform(0.05, 1.5, (T, N, S))
    q = rng.standard_normal((T, 4)) * 0.03 + np.array([0, 0, 0, 1.0])
    d = dict(trajs_2d=np.stack([rng.uniform(0, W - 1, (T, N, S)), rng.uniform(0, H - 1, (T, N, S))], -1), trajs_disp=disp,
             trajs_disp_mono=disp * rng.uniform(0.8, 1.25, (T, 1, 1)) * (1 + 0.03 * rng.standard_normal((T, N, S))),
             trajs_vis=rng.uniform(0.5, 1.0, (T, N, S)), trajs_static=rng.uniform(0.2, 1.0, (T, N, S)), jj=jj.astype(np.int64),
             intrinsics=np.tile(np.array([500.0, 500.0, W / 2, H / 2]), (T, 1)),
             pose=np.concatenate([rng.standard_normal((T, 3)) * 0.1, q / np.linalg.norm(q, axis=1, keepdims=True)], 1),
             grid_query_frames=np.arange(0, T, 2).astype(np.int64), trajs_scales=rng.standard_normal((T, N, S)) * 0.2,
             frame_scales_=rng.standard_normal((T, gh, gw)), frame_shifts=np.zeros(T), H=np.int64(H), W=np.int64(W), pw_break=np.float64(20.0))
    return {k: (np.asarray(v, np.float32).astype(np.float64) if np.asarray(v).dtype == np.float64 and np.ndim(v) > 0 else v) for k, v in d.items()}


# (the pairwise kernels walk the tracks two by two: odd N, an odd number of such pairs — 301 -> 151 — and a handful of tracks)
@pytest.mark.parametrize("T,N,S", [(12, 300, 7), (6, 520, 5), (5, 301, 5), (4, 7, 3), (4, 2, 3), (3, 1030, 3)])
def test_larger_cases_vs_oracle(T, N, S):
    d = make_case(T, N, S, seed=T + N)
    net = build(d)
    l = net.losses().cpu().numpy()[:3]
    ms = ga.frame_scaled_depth(d)
    ref = (ga.spatial_loss(d, ms), ga.inter_frame_loss(d, ms), ga.pts_3d_loss(d, ms))
    for got, want in zip(l, ref):
        assert abs(got / want - 1) < 1e-5, (got, want)


def test_float16_depth_residuals():
    """BASELINE.json configs[4]: disparities stored in float16, the depth residual formed in float16."""
    d = make_case(10, 256, 7, seed=5)
    d16 = dict(d)
    for k in ("trajs_disp", "trajs_disp_mono"):
        d16[k] = d[k].astype(np.float16).astype(np.float64)            # the oracle sees exactly the float16 values
    net = build(d16, half=True)
    l = net.losses().cpu().numpy()
    ms = ga.frame_scaled_depth(d16)
    ref = (ga.spatial_loss(d16, ms), ga.inter_frame_loss(d16, ms), ga.pts_3d_loss(d16, ms))
    assert abs(l[0] / ref[0] - 1) < 2e-3                              # residual rounded to float16
    assert abs(l[1] / ref[1] - 1) < 1e-5 and abs(l[2] / ref[2] - 1) < 1e-5
    assert net.trajs_disp.dtype == torch.float16


def test_cpu_tensors_raise():
    from batrack_amd.global_refine import RefineLosses
    t = lambda k: torch.as_tensor(np.asarray(D[k]))
    with pytest.raises(RuntimeError):
        RefineLosses(t("trajs_2d"), t("trajs_disp"), t("trajs_disp_mono"), t("trajs_vis"), t("trajs_static"), t("jj"), t("intrinsics"),
                     t("grid_query_frames"), t("trajs_scales"), t("frame_scales_"), t("frame_shifts"), t("pose"), 96, 128)


def _gerr(got, ref):
    """Error of a gradient tensor relative to its largest entry (many entries are exactly zero: masked tracks, frames
    that are no query frame)."""
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("alpha,key", [(0.0, "a00"), (0.5, "a05")])
def test_gradients_match_the_reference_autograd(alpha, key):
    """bt_ga_backward against the gradients the reference's OWN autograd produced for forward() (fixture keys *.grad_*)."""
    net = build(D)
    g = net.backward(alpha)
    g_ts, g_fs = g["trajs_scales"], g["frame_scales_"]
    assert _gerr(g_ts.cpu().numpy(), D[f"f64.grad_trajs_scales_{key}"]) < 2e-5
    assert _gerr(g_fs.cpu().numpy(), D[f"f64.grad_frame_scales_{key}"]) < 2e-5
    # frames that are no query frame carry no gradient in trajs_scales, exactly
    notq = np.setdiff1d(np.arange(net.T), D["grid_query_frames"])
    assert float(g_ts[torch.as_tensor(notq, device=g_ts.device)].abs().max()) == 0.0


@pytest.mark.parametrize("T,N,S", [(12, 300, 7), (6, 520, 5), (5, 301, 5), (4, 7, 3), (4, 2, 3), (3, 1030, 3)])
def test_gradients_of_larger_cases_vs_torch_oracle(T, N, S):
    from oracle import ga_torch
    d = make_case(T, N, S, seed=T + N)
    net = build(d)
    g = net.backward(0.5)
    g_ts, g_fs = g["trajs_scales"], g["frame_scales_"]
    tot, _, _, r_ts, r_fs = ga_torch.total_and_grads(d, 0.5)
    assert abs(float(net.forward(0.5)) / tot - 1) < 1e-5
    # float32 sums of up to N signed unit vectors per track against float64 autograd
    assert _gerr(g_ts.cpu().numpy(), r_ts) < 5e-5 and _gerr(g_fs.cpu().numpy(), r_fs) < 2e-4, (_gerr(g_ts.cpu().numpy(), r_ts), _gerr(g_fs.cpu().numpy(), r_fs))


def test_adam_steps_through_the_autograd_node():
    """The reference's refinement loop (trainer.py:23-77) in miniature: Adam on the two parameters through
    `net.loss(alpha).backward()`; same loss trajectory as the float64 torch oracle under the same optimiser."""
    from oracle import ga_torch
    d = make_case(8, 200, 5, seed=3)
    net = build(d)
    net.trajs_scales.requires_grad_(True); net.frame_scales_.requires_grad_(True)
    opt = torch.optim.Adam([net.trajs_scales, net.frame_scales_], lr=0.05)
    ts = torch.as_tensor(d["trajs_scales"], dtype=torch.float64).requires_grad_(True)
    fs = torch.as_tensor(d["frame_scales_"], dtype=torch.float64).requires_grad_(True)
    ropt = torch.optim.Adam([ts, fs], lr=0.05)
    hip, ref = [], []
    for _ in range(8):
        opt.zero_grad()
        l = net.loss(0.5)
        l.backward()
        opt.step()
        hip.append(float(l.detach()))
        ropt.zero_grad()
        ms = ga_torch.frame_scaled_depth(d, fs)
        lr_ = ga_torch.spatial_loss(d, ts, ms) + 0.5 * ga_torch.inter_frame_loss(d, ms)
        lr_.backward()
        ropt.step()
        ref.append(float(lr_.detach()))
    assert hip[-1] < 0.98 * hip[0] and all(b < a for a, b in zip(hip, hip[1:]))      # the optimiser makes progress, every step
    assert max(abs(a / b - 1) for a, b in zip(hip, ref)) < 2e-3, (hip, ref)


# ------------------------------------------------------------------ the total the reference optimises (refine_net.py:274-392)
G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ga_total.npz")))
GD = {k: G[k] for k in G if "." not in k}
RUN_WEIGHTS = {"spatial_loss": 5.0, "inter_frame_loss": 0.3, "pts_3d_loss": 1.0, "cam_smooth_vec_loss": 1.0,
               "scale_smoothness_loss": 0.3}                      # run_global_refine.py:61-67


def _settings(name):
    """A: what run_global_refine.py runs (weights dict, intrinsics refined); B: loss_weight_dict=None at the constructor's
    defaults (alpha 0.5, scale_smoothness_weight 0.1)."""
    return (dict(loss_weight_dict=RUN_WEIGHTS, refine_intrinsics=True, alpha=0.5, scale_smoothness_weight=0.1) if name == "A" else
            dict(loss_weight_dict=None, refine_intrinsics=False, alpha=0.5, scale_smoothness_weight=0.1))


@pytest.mark.parametrize("name", ["A", "B"])
def test_total_and_every_gradient_match_the_reference(name):
    """bt_ga_forward / bt_ga_backward_total against the reference's unmodified RefineNet.forward and its own autograd
    (tests/golden/ga_total.npz): every term, the weighted total, and the gradient w.r.t. trajs_scales, frame_scales_, pose
    (pypose's convention) and K."""
    net = build(GD, **_settings(name))
    if name == "A":
        assert np.allclose(net.K.cpu().numpy(), G["f64.A.K_init"], rtol=1e-6)        # the lower median (torch.median), / K_scale
    l = net.losses("l1").cpu().numpy()
    for got, key in zip(l, ("spatial", "rigid", "pts3d", "cam_smooth", "scale_smooth_l1")):
        assert abs(got / float(G[f"f64.{name}.{key}"]) - 1) < 1e-5, (key, got)
    for mode in ("l1", "l2", "huber"):
        assert abs(float(net.scale_grid_smoothness_loss(mode)) / float(G[f"f64.{name}.scale_smooth_{mode}"]) - 1) < 1e-5
    assert abs(float(net.cam_smooth_vec_loss()) / float(G[f"f64.{name}.cam_smooth"]) - 1) < 1e-5
    assert abs(float(net.forward()) / float(G[f"f64.{name}.total"]) - 1) < 1e-5
    g = net.backward()
    for k, key in (("trajs_scales", "grad_trajs_scales"), ("frame_scales_", "grad_frame_scales"), ("pose", "grad_pose"), ("K", "grad_K")):
        ref = G[f"f64.{name}.{key}"]
        if np.abs(ref).max() == 0:
            assert float(g[k].abs().max()) == 0.0, k
        else:
            assert _gerr(g[k].cpu().numpy(), ref) < 5e-5, (name, k, _gerr(g[k].cpu().numpy(), ref))


def test_adam_trajectory_of_the_full_total_vs_torch_oracle():
    """20 iterations of the reference's loop (trainer.py:23-77: Adam, betas (0.9, 0.9), pose and K at lr 1e-2) on the total
    run_global_refine.py optimises, through `net.loss().backward()`; the float64 torch oracle runs the same optimiser."""
    from oracle import ga_torch
    d = make_case(8, 160, 5, seed=9)
    d["intrinsics"] = d["intrinsics"] * (1 + 0.01 * np.random.default_rng(1).standard_normal(d["intrinsics"].shape))
    net = build(d, **_settings("A"))
    for p in (net.trajs_scales, net.frame_scales_, net.pose, net.K):
        p.requires_grad_(True)
    groups = lambda a, b, c, e: [{"params": [a], "lr": 1e-2}, {"params": [b], "lr": 1e-2}, {"params": [c], "lr": 1e-2}, {"params": [e], "lr": 1e-2}]
    opt = torch.optim.Adam(groups(net.trajs_scales, net.frame_scales_, net.pose, net.K), lr=1e-2, betas=(0.9, 0.9))
    f64 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    ts, fs, ps = f64(d["trajs_scales"]), f64(d["frame_scales_"]), f64(d["pose"])
    K = f64(net.K.detach().cpu().numpy())
    rparams = [t.clone().requires_grad_(True) for t in (ts, fs, ps, K)]
    ropt = torch.optim.Adam(groups(*rparams), lr=1e-2, betas=(0.9, 0.9))
    w = [RUN_WEIGHTS[k] for k in ("spatial_loss", "inter_frame_loss", "pts_3d_loss", "cam_smooth_vec_loss", "scale_smoothness_loss")]
    hip, ref = [], []
    for _ in range(20):
        opt.zero_grad()
        l = net.loss()
        l.backward()
        opt.step()
        hip.append(float(l.detach()))
        r = ga_torch.full_total_and_grads(d, w, "l1", refine_intrinsics=True, K=rparams[3].detach().numpy(),
                                          trajs_scales=rparams[0].detach().numpy(), frame_scales_=rparams[1].detach().numpy(),
                                          pose=rparams[2].detach().numpy())
        ropt.zero_grad()
        for p_, k in zip(rparams, ("grad_trajs_scales", "grad_frame_scales", "grad_pose", "grad_K")):
            p_.grad = torch.as_tensor(r[k])
        ropt.step()
        ref.append(r["total"])
    assert hip[-1] < 0.9 * hip[0]
    assert max(abs(a / b - 1) for a, b in zip(hip, ref)) < 2e-3, (hip, ref)
    assert np.abs(net.pose.detach().cpu().numpy() - rparams[2].detach().numpy()).max() < 2e-3
    assert np.abs(net.K.detach().cpu().numpy() - rparams[3].detach().numpy()).max() < 2e-3 * np.abs(rparams[3].detach().numpy()).max()


def test_float16_depth_residual_backward():
    """half_disp=True: the spatial term's residual is formed in float16 in the backward as in the forward (the derivative
    goes straight through the rounding); against the float64 oracle on the float16-rounded disparities, 2e-3 as the
    forward (float16 has 11 bits: an entry of the clamped residual moves by up to 5e-4)."""
    from oracle import ga_torch
    d = make_case(8, 256, 5, seed=6)
    d16 = dict(d)
    for k in ("trajs_disp", "trajs_disp_mono"):
        d16[k] = d[k].astype(np.float16).astype(np.float64)
    net = build(d16, half=True, scale_smoothness_weight=0.1)
    g = net.backward()
    r = ga_torch.full_total_and_grads(d16, [1.0, 0.5, 0.0, 0.0, 0.1], "l1")
    assert abs(float(net.forward()) / r["total"] - 1) < 2e-3
    assert _gerr(g["trajs_scales"].cpu().numpy(), r["grad_trajs_scales"]) < 2e-3
    assert _gerr(g["frame_scales_"].cpu().numpy(), r["grad_frame_scales"]) < 2e-3


def test_repeated_query_frames_count_as_often_as_they_are_listed():
    """`loss[self.grid_query_frames.long()].mean()` (refine_net.py:223,265) counts a frame that is listed twice twice; the
    oracle indexes the same way.  Total and every gradient with the full weight set, against the float64 oracle."""
    from oracle import ga_torch
    d = dict(GD)
    d["grid_query_frames"] = np.array([0, 2, 2, 7, 2, 9, 7], np.int64)
    w = [RUN_WEIGHTS[k] for k in ("spatial_loss", "inter_frame_loss", "pts_3d_loss", "cam_smooth_vec_loss", "scale_smoothness_loss")]
    net = build(d, **_settings("A"))
    r = ga_torch.full_total_and_grads(d, w, "l1", refine_intrinsics=True)
    assert abs(float(net.forward()) / r["total"] - 1) < 1e-5
    l = net.losses().cpu().numpy()
    for i, k in enumerate(("spatial", "rigid", "pts3d", "cam_smooth", "scale_smooth")):
        assert abs(l[i] / r[k] - 1) < 1e-5, (k, l[i], r[k])
    g = net.backward()
    for key, name in (("trajs_scales", "grad_trajs_scales"), ("frame_scales_", "grad_frame_scales"), ("pose", "grad_pose"), ("K", "grad_K")):
        assert _gerr(g[key].cpu().numpy(), np.asarray(r[name])) < 5e-5, name
    with pytest.raises(ValueError):
        build(dict(GD, grid_query_frames=np.array([0, 99], np.int64)))          # out of range is still refused
