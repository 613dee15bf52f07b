"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU oracle for the BA hot path (restatement of /root/reference/main/backend/ba.py
and projective_ops.py; see ba_oracle_impl.h for the line-by-line citations).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; batrack_amd/ never does.

  oracle.build()                      compile oracle/_build/libba_oracle.so (gcc)
  oracle.ba_step(...)                 one BA_rgbd_droid call, float64 or float32
  oracle.edges(...)                   per-edge reprojection / Jacobians / masks
  oracle.refseq (module)              torch-CPU restatement keeping the reference's
                                      operator sequence (oracle/refseq.py; timed as cpu_baseline "refseq")
  oracle.se3_torch (module)           torch statement of the SE3 formulas + SE3Ref (checker of row f-1)
  oracle.patchify (module)            numpy restatement of altcorr.patchify (checker of row f-2)
  oracle.ga_losses (module)           torch statement of the global-refinement forward losses (row f-4)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libba_oracle.so")
_lib = None
LOSS = {"trivial": 0, "huber": 1, "cauchy": 2}


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("ba_oracle.c", "ba_oracle_impl.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _prep(dtype, poses, patches, mono, intrinsics, targets, weights, ii, jj, kk, bounds):
    dt = np.dtype(dtype)
    assert dt in (np.float64, np.float32)
    c = lambda a: np.ascontiguousarray(np.asarray(a), dtype=dt)
    i64 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.int64)
    targets = np.asarray(targets)
    tstride = targets.shape[-1]          # 2, or 3 for (u, v, disp) rows
    return (dt, c(poses).reshape(-1, 7), c(patches).reshape(-1, 3), c(mono).reshape(-1),
            c(intrinsics).reshape(-1, 4), c(targets).reshape(-1, tstride), tstride,
            c(weights).reshape(-1, 2), i64(ii), i64(jj), i64(kk), c(bounds).reshape(4))


def ba_step(poses, patches, mono, intrinsics, targets, weights, ii, jj, kk, bounds,
            lmbda=1e-4, ep=10.0, alpha=0.05, fixedp=1, structure_only=False,
            loss="huber", dtype=np.float64, want_system=False):
    """One reference-equivalent BA_rgbd_droid call on the CPU.

    targets may be [E,2] or [E,3] rows (u, v[, disp]) — the reference receives a
    stride-3 view (batrack.py:871).  Returns dict(poses_out [N,7], patches_out
    [P,3], failed, and S/y/dX when want_system and a solve happened)."""
    lib = _load()
    (dt, poses, patches, mono, intr, targets, tstride, weights, ii, jj, kk, bounds) = _prep(
        dtype, poses, patches, mono, intrinsics, targets, weights, ii, jj, kk, bounds)
    E, N, P = ii.shape[0], poses.shape[0], patches.shape[0]
    n_all = int(max(ii.max(), jj.max())) + 1 if E else 0
    n = max(n_all - int(fixedp), 0)
    poses_out = np.empty_like(poses)
    patches_out = np.empty_like(patches)
    S = np.zeros((6 * n, 6 * n), dt) if want_system else None
    y = np.zeros(6 * n, dt) if want_system else None
    dX = np.zeros(6 * n, dt) if want_system else None
    real = ctypes.c_double if dt == np.float64 else ctypes.c_float
    fn = getattr(lib, "oracle_ba_step_f64" if dt == np.float64 else "oracle_ba_step_f32")
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_void_p] + [ctypes.c_void_p] * 3 + \
        [ctypes.c_int64] * 3 + [ctypes.c_void_p, real, real, real, ctypes.c_int64, ctypes.c_int, ctypes.c_int] + \
        [ctypes.c_void_p] * 5
    rc = fn(_ptr(poses), _ptr(patches), _ptr(mono), _ptr(intr), _ptr(targets), tstride, _ptr(weights),
            _ptr(ii), _ptr(jj), _ptr(kk), E, N, P, _ptr(bounds), lmbda, ep, alpha, int(fixedp),
            int(bool(structure_only)), LOSS[loss], _ptr(poses_out), _ptr(patches_out), _ptr(S), _ptr(y), _ptr(dX))
    if rc < 0:
        raise ValueError("oracle_ba_step: bad arguments")
    out = dict(poses_out=poses_out, patches_out=patches_out, failed=bool(rc))
    if want_system and n > 0 and not structure_only:
        out.update(S=S, y=y, dX=dX.reshape(n, 6))
    return out


def edges(poses, patches, intrinsics, targets, weights, ii, jj, kk, bounds, loss="huber",
          dtype=np.float64):
    """Per-edge coords / valid / Ji / Jj / Jz / masked residual / final weights."""
    lib = _load()
    mono = np.zeros(np.asarray(patches).reshape(-1, 3).shape[0])
    (dt, poses, patches, _, intr, targets, tstride, weights, ii, jj, kk, bounds) = _prep(
        dtype, poses, patches, mono, intrinsics, targets, weights, ii, jj, kk, bounds)
    E = ii.shape[0]
    o = dict(coords=np.empty((E, 2), dt), valid=np.empty(E, dt), Ji=np.empty((E, 2, 6), dt),
             Jj=np.empty((E, 2, 6), dt), Jz=np.empty((E, 2), dt), r=np.empty((E, 2), dt),
             W=np.empty((E, 2), dt))
    fn = getattr(lib, "oracle_edges_f64" if dt == np.float64 else "oracle_edges_f32")
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 7
    fn(_ptr(poses), _ptr(patches), _ptr(intr), _ptr(targets), tstride, _ptr(weights), _ptr(ii), _ptr(jj),
       _ptr(kk), E, _ptr(bounds), LOSS[loss], *[_ptr(o[k]) for k in ("coords", "valid", "Ji", "Jj", "Jz", "r", "W")])
    return o
