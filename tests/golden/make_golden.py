#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by running the
reference's UNMODIFIED BA path in this container.

Run from the repo root, in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

What is executed: /root/reference/main/backend/{ba.py, projective_ops.py,
lietorch/{groups,group_ops,broadcasting}.py} exactly as they lie there, imported
with two stand-in modules from tests/golden/refstubs/ for dependencies that are
absent from the image (torch_scatter; the Eigen/CUDA extension lietorch_backends).
Consequences, stated once here and in DESIGN.md: every line of ba.py and
projective_ops.py is pinned by these vectors; the SE3 primitive arithmetic and
scatter_sum are our restatement (pinned only by algebraic identities).

Nothing of the reference is written out: the .npz files hold inputs we generated
(batrack_amd.graphgen) and the numeric outputs of the calls.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "refstubs"), os.path.join(REF, "main"), REF, ROOT]

import backend.ba as ref_ba                      # noqa: E402  (reference, unmodified)
import backend.projective_ops as ref_pops        # noqa: E402
from backend.lietorch import SE3 as RefSE3       # noqa: E402

from batrack_amd import graphgen                 # noqa: E402

torch.set_num_threads(8)


def f32r(a):
    """Round to fp32-representable values (kept as float64)."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def as_inputs(g):
    """fp32-representable copies of every floating input of a Graph."""
    return dict(poses=f32r(g.poses), patches=f32r(g.patches), mono=f32r(g.mono_disp),
                intrinsics=f32r(g.intrinsics), targets3=f32r(g.targets3),
                weights=f32r(g.weights), weights_pose=f32r(g.weights_pose),
                ii=g.ii.astype(np.int64), jj=g.jj.astype(np.int64), kk=g.kk.astype(np.int64),
                bounds=np.asarray(g.bounds, dtype=np.float64))


class Capture:
    """Records the reduced system the reference hands to its solver, without
    touching the reference source: wraps backend.ba.block_solve."""

    def __init__(self):
        self.calls = []
        self._orig = ref_ba.block_solve

    def __enter__(self):
        def wrapped(A, B, ep=1.0, lm=1e-4):
            X = self._orig(A, B, ep=ep, lm=lm)
            b, n = A.shape[0], A.shape[1]
            self.calls.append(dict(
                S=A.permute(0, 1, 3, 2, 4).reshape(6 * n, 6 * n).detach().clone().numpy(),
                y=B.permute(0, 1, 3, 2, 4).reshape(6 * n).detach().clone().numpy(),
                dX=X.reshape(n, 6).detach().clone().numpy(), lm=lm))
            return X
        ref_ba.block_solve = wrapped
        return self

    def __exit__(self, *exc):
        ref_ba.block_solve = self._orig


def run_ref(inp, dtype, weights_key, fixedp, structure_only, loss="huber",
            lmbda=1e-4, ep=10.0, alpha=0.05, poses=None, patches=None):
    """One reference BA_rgbd_droid call, argument pattern of batrack.py:871-875."""
    td = dict(dtype=dtype)
    P = torch.as_tensor(inp["poses"] if poses is None else poses, **td)[None]
    pat = torch.as_tensor(inp["patches"] if patches is None else patches, **td)[None, :, :, None, None]
    mono = torch.as_tensor(inp["mono"], **td)[None, :, None]
    intr = torch.as_tensor(inp["intrinsics"], **td)[None]
    t3 = torch.as_tensor(inp["targets3"], **td)[None]
    w = torch.as_tensor(inp[weights_key], **td)[None]
    ii, jj, kk = (torch.as_tensor(inp[k]) for k in ("ii", "jj", "kk"))
    bounds = [float(v) for v in inp["bounds"]]
    with Capture() as cap:
        Gs, pout = ref_ba.BA_rgbd_droid(
            RefSE3(P), pat, mono, intr, t3[..., :2], t3[..., 2:], w, lmbda, ii, jj, kk,
            bounds, ep=ep, fixedp=fixedp, structure_only=structure_only, loss=loss, alpha=alpha)
    out = dict(poses_out=Gs.data[0].numpy().copy(), patches_out=pout[0, :, :, 0, 0].numpy().copy())
    if cap.calls:
        out.update({k: cap.calls[0][k] for k in ("S", "y", "dX")})
        out["n_solves"] = np.int64(len(cap.calls))
    return out


def run_transform(inp, dtype):
    td = dict(dtype=dtype)
    P = torch.as_tensor(inp["poses"], **td)[None]
    pat = torch.as_tensor(inp["patches"], **td)[None, :, :, None, None]
    intr = torch.as_tensor(inp["intrinsics"], **td)[None]
    ii, jj, kk = (torch.as_tensor(inp[k]) for k in ("ii", "jj", "kk"))
    coords, valid, (Ji, Jj, Jz) = ref_pops.transform(RefSE3(P), pat, intr, ii, jj, kk, jacobian=True)
    return dict(coords=coords[0, :, 0, 0].numpy(), valid=valid[0].numpy(),
                Ji=Ji[0].numpy(), Jj=Jj[0].numpy(), Jz=Jz[0, :, :, 0].numpy())


def dual_iterations(inp, dtype, fixedp, iters, loss="huber"):
    """BATRACK.update()'s loop (batrack.py:869-875): ITER x {pose+structure with
    weights_pose ; structure-only with weights}."""
    poses, patches = inp["poses"], inp["patches"]
    for _ in range(iters):
        o = run_ref(inp, dtype, "weights_pose", fixedp, False, loss, poses=poses, patches=patches)
        poses, patches = o["poses_out"], o["patches_out"]
        o = run_ref(inp, dtype, "weights", fixedp, True, loss, poses=poses, patches=patches)
        poses, patches = o["poses_out"], o["patches_out"]
    return dict(poses_out=poses, patches_out=patches)


def pack(prefix, d):
    return {f"{prefix}.{k}": v for k, v in d.items()}


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def main():
    f64, f32 = torch.float64, torch.float32

    # ---- C1: 8 KF / 2,048 edges, every intermediate -----------------------
    g = graphgen.make_config("C1", seed=0, n_buf=10)
    inp = as_inputs(g)
    arrs = dict(inp)
    arrs.update(pack("tf64", run_transform(inp, f64)))
    for tag, dt in (("f64", f64), ("f32", f32)):
        arrs.update(pack(f"ps_fp1.{tag}", run_ref(inp, dt, "weights_pose", 1, False)))
        arrs.update(pack(f"ps_fp3.{tag}", run_ref(inp, dt, "weights_pose", 3, False)))
        arrs.update(pack(f"so.{tag}", run_ref(inp, dt, "weights", 1, True)))
        arrs.update(pack(f"triv.{tag}", run_ref(inp, dt, "weights_pose", 1, False, loss="trivial")))
        arrs.update(pack(f"cauchy.{tag}", run_ref(inp, dt, "weights_pose", 1, False, loss="cauchy")))
        arrs.update(pack(f"dual2.{tag}", dual_iterations(inp, dt, 1, 2)))
        arrs.update(pack(f"allfixed.{tag}", run_ref(inp, dt, "weights_pose", 8, False)))
    save("c1", **arrs)

    # ---- C1 roughened: thresholds, masks, shuffled order, odd intrinsics --
    g = graphgen.roughen(graphgen.make_config("C1", seed=3, n_buf=12), seed=5)
    inp = as_inputs(g)
    arrs = dict(inp)
    arrs.update(pack("tf64", run_transform(inp, f64)))
    for tag, dt in (("f64", f64), ("f32", f32)):
        arrs.update(pack(f"ps_fp1.{tag}", run_ref(inp, dt, "weights_pose", 1, False)))
        arrs.update(pack(f"ps_fp2.{tag}", run_ref(inp, dt, "weights_pose", 2, False, alpha=0.5, ep=100.0)))
        arrs.update(pack(f"so.{tag}", run_ref(inp, dt, "weights", 1, True)))
        arrs.update(pack(f"dual2.{tag}", dual_iterations(inp, dt, 2, 2)))
    save("c1_rough", **arrs)

    # ---- small sliding-window graph (duplicates, fixed window) ------------
    g, fixedp = graphgen.make_window_graph(n_frames=24, M=24, seed=2, n_buf=26)
    inp = as_inputs(g)
    arrs = dict(inp)
    arrs["fixedp"] = np.int64(fixedp)
    for tag, dt in (("f64", f64), ("f32", f32)):
        arrs.update(pack(f"ps.{tag}", run_ref(inp, dt, "weights_pose", fixedp, False)))
        arrs.update(pack(f"so.{tag}", run_ref(inp, dt, "weights", fixedp, True)))
        arrs.update(pack(f"dual2.{tag}", dual_iterations(inp, dt, fixedp, 2)))
    save("window_small", **arrs)

    # ---- C3: 64 KF / 131,072 edges — outputs only (inputs come from seed) --
    g = graphgen.make_config("C3", seed=0)
    inp = as_inputs(g)
    arrs = {}
    for tag, dt in (("f64", f64), ("f32", f32)):
        o = run_ref(inp, dt, "weights_pose", 1, False)
        arrs.update(pack(f"ps.{tag}", dict(poses_out=o["poses_out"], disp_out=o["patches_out"][:, 2],
                                           dX=o["dX"], y=o["y"], S_diag=np.diag(o["S"]).copy())))
        o2 = run_ref(inp, dt, "weights", 1, True, poses=o["poses_out"], patches=o["patches_out"])
        arrs.update(pack(f"so.{tag}", dict(disp_out=o2["patches_out"][:, 2])))
    arrs["seed"] = np.int64(0)
    save("c3", **arrs)


def main_nan():
    """Non-finite input: the C1 graph with ONE NaN target (first coordinate of the 6th edge that carries pose
    weight).  The reference masks by multiplication, so the NaN survives as 0 * NaN (ba.py:233-251)."""
    g = graphgen.make_config("C1", seed=0, n_buf=10)
    inp = as_inputs(g)
    e0 = int(np.flatnonzero(inp["weights_pose"][:, 0] > 0)[5])
    t = inp["targets3"].copy()
    t[e0, 0] = np.nan
    arrs = dict(e0=np.int64(e0))
    for loss in ("huber", "cauchy", "trivial"):
        o = run_ref(dict(inp, targets3=t), torch.float64, "weights_pose", 1, False, loss=loss)
        arrs.update(pack(loss, dict(poses_out=o["poses_out"], patches_out=o["patches_out"], dX=o["dX"], n_solves=o["n_solves"])))
    save("c1_nan", **arrs)


if __name__ == "__main__":
    main_nan() if sys.argv[1:] == ["nan"] else (main(), main_nan())
