"""oracle/ga_init.py (the hand-off results.pkl -> global alignment) against what the reference's
unmodified RefineNet.__init__ / _init_from_ba derived from the same dictionary (tests/golden/ga_init.npz,
tests/golden/make_golden_ga_init.py)."""
import os

import numpy as np

from oracle import ga_init

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ga_init.npz"), allow_pickle=False))
RES = {k[3:]: v for k, v in G.items() if k.startswith("in.")}
RES.update(rgbs=None, dmaps_gt=None)


def same_rotation(p, q):
    """poses [T,7] equal up to the sign of the quaternion"""
    s = np.sign(np.sum(p[:, 3:] * q[:, 3:], axis=1, keepdims=True))
    return max(np.abs(p[:, :3] - q[:, :3]).max(), np.abs(p[:, 3:] * s - q[:, 3:]).max())


def test_derived_tensors_match_the_reference():
    for tag, align in (("plain", False), ("aligned", True)):
        o = ga_init.init_from_ba(RES, align_depth=align)
        assert (o["T"], o["N"], o["S_local"], o["H"], o["W"]) == tuple(G[f"{tag}.T_N_S_H_W"])
        assert np.array_equal(o["jj"], G[f"{tag}.jj"]) and np.array_equal(o["ii"], G[f"{tag}.ii"])
        assert np.array_equal(o["trajs_2d"], G[f"{tag}.trajs_2d"]) and np.array_equal(o["trajs_disp"], G[f"{tag}.trajs_disp"])
        assert np.abs(o["K_init"] - G[f"{tag}.K_init"]).max() < 1e-6                       # (the reference divides in float32)
        assert same_rotation(o["pose_init"], G[f"{tag}.pose_init"].astype(np.float64)) < 1e-6
        ref = G[f"{tag}.trajs_disp_mono"]
        assert ref.dtype == np.float64 and np.abs(o["trajs_disp_mono"] - ref).max() <= 1e-12 * np.abs(ref).max()
    # the fixture exercises what it is meant to: the clamp of the depth, out-of-image tracks, every branch of mat2SE3
    assert (G["plain.trajs_disp_mono"] == 100.0).any()
    t2 = G["plain.trajs_2d"]
    assert (t2[..., 0] < 0).any() and (t2[..., 0] > G["plain.T_N_S_H_W"][4] - 1).any()
    tr = np.trace(RES["cams_T_world"][:, :3, :3], axis1=1, axis2=2)
    assert (tr > 0).any() and (tr < 0).sum() >= 3
    assert np.abs(G["aligned.trajs_disp_mono"] - G["plain.trajs_disp_mono"]).max() > 1e-3


def test_total_from_the_derived_tensors_matches_the_reference_forward():
    """dictionary -> oracle init -> the float64 torch statement of forward() (oracle/ga_torch.py) = RefineNet.forward right
    after __init__ and after the seeded perturbation of the parameters, gradients included."""
    from oracle import ga_torch
    o = ga_init.init_from_ba(RES)
    w = list(G["weights"])
    base = dict(trajs_2d=o["trajs_2d"], trajs_disp=o["trajs_disp"], trajs_disp_mono=o["trajs_disp_mono"], trajs_vis=o["trajs_vis"],
                trajs_static=o["trajs_static"], jj=o["jj"], intrinsics=o["intrinsics_raw"], pose=G["plain.pose_init"].astype(np.float64),
                grid_query_frames=o["grid_query_frames"], frame_shifts=np.zeros(o["T"]), H=np.int64(o["H"]), W=np.int64(o["W"]), pw_break=np.float64(20.0))
    for tag in ("init", "pert"):
        d = dict(base, trajs_scales=G[f"{tag}.trajs_scales"], frame_scales_=G[f"{tag}.frame_scales_"])
        d = {k: (np.asarray(v, np.float64) if np.asarray(v).dtype.kind == "f" else v) for k, v in d.items()}
        r = ga_torch.full_total_and_grads(d, w, "l1", refine_intrinsics=True)
        assert abs(r["total"] / float(G[f"{tag}.total"]) - 1) < 2e-6, (tag, r["total"], float(G[f"{tag}.total"]))     # the reference ran in float32
        if tag == "pert":
            for got, name in ((r["grad_trajs_scales"], "grad_trajs_scales"), (r["grad_frame_scales"], "grad_frame_scales"), (r["grad_pose"], "grad_pose"), (r["grad_K"], "grad_K")):
                ref = G[f"pert.{name}"].astype(np.float64)
                assert np.abs(np.asarray(got) - ref).max() < 2e-5 * max(np.abs(ref).max(), 1e-30), name
