"""SE3 arithmetic against an implementation neither this repo nor the reference wrote: scipy.

The arithmetic inside the reference's compiled `lietorch_backends` (lietorch/include/so3.h:31-190, se3.h:36-142) cannot be
pinned by the reference in this image (Eigen is absent, the module cannot be built; the reference holds no vectors for it —
SURVEY.md §8c).  What CAN be done is a cross-check against third-party code: `scipy.spatial.transform.Rotation`
(exp / log / compose / apply of the rotation part) and `scipy.linalg.expm` / `logm` on the 4x4 twist matrix (a generic
Pade matrix exponential: the translation part of Exp and Log, V(phi) tau, with no closed form of ours in it).

* CPU (always runs): `oracle/se3_torch.py`, the checker of the HIP kernels and of the golden fixtures' stand-in, float64,
  <= 1e-12.
* `-m gpu`: the HIP kernels of batrack_amd/csrc/se3_kernels.hip themselves (through lietorch_backends, float64) against
  the same scipy results, <= 1e-12 (log: 1e-9, as against the oracle) — no code of this repo on the expected side.
"""
import numpy as np
import pytest
import torch
from scipy.linalg import expm as sp_expm, logm as sp_logm
from scipy.spatial.transform import Rotation

from oracle import se3_torch as so

RNG = np.random.default_rng(7)
B = 200


def hat(a):
    """4x4 twist matrix of a = (tau, phi): [[phi]x, tau; 0, 0]"""
    tx, ty, tz, px, py, pz = a
    return np.array([[0, -pz, py, tx], [pz, 0, -px, ty], [-py, px, 0, tz], [0, 0, 0, 0]], dtype=np.float64)


def to_matrix(d):
    """[B,7] (t, q xyzw) -> [B,4,4] with the rotation by scipy"""
    d = np.asarray(d, np.float64)
    M = np.tile(np.eye(4), (d.shape[0], 1, 1))
    M[:, :3, :3] = Rotation.from_quat(d[:, 3:7]).as_matrix()
    M[:, :3, 3] = d[:, :3]
    return M


def tangents():
    a = 0.8 * RNG.standard_normal((B, 6))
    a[:8, 3:] *= 1e-9                                   # the series branch (theta < EPS)
    a[8:16, 3:] *= 3.0 / np.linalg.norm(a[8:16, 3:], axis=1, keepdims=True)   # large angles (3 rad)
    return a


def quat_close(q, qref, tol):
    s = np.sign((q * qref).sum(-1, keepdims=True))
    return np.abs(q - s * qref).max() < tol


# ---- the implementations under test: name -> callables on numpy float64 [B, dim] arrays
def cpu_impl():
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64)
    n = lambda x: x.numpy()
    return dict(exp=lambda a: n(so.exp(t(a))), log=lambda X: n(so.log(t(X))), inv=lambda X: n(so.inv(t(X))),
                mul=lambda X, Y: n(so.mul(t(X), t(Y))), act=lambda X, p: n(so.act(t(X), t(p))),
                adjT=lambda X, a: n(so.adjT(t(X), t(a))), log_tol=1e-9)


def gpu_impl():
    from batrack_amd.backend import lietorch_backends as lb
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64, device="cuda:0")
    n = lambda x: x.cpu().numpy()
    return dict(exp=lambda a: n(lb.expm(3, t(a))), log=lambda X: n(lb.logm(3, t(X))), inv=lambda X: n(lb.inv(3, t(X))),
                mul=lambda X, Y: n(lb.mul(3, t(X), t(Y))),
                act=lambda X, p: n(lb.act(3, t(X), t(p)) if p.shape[1] == 3 else lb.act4(3, t(X), t(p))),
                adjT=lambda X, a: n(lb.adjT(3, t(X), t(a))), log_tol=1e-9)


IMPLS = [pytest.param(cpu_impl, id="oracle_se3_torch"),
         pytest.param(gpu_impl, id="hip_kernels", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("impl", IMPLS)
def test_exp_matches_scipy_matrix_exponential(impl):
    f = impl()
    a = tangents()
    X = f["exp"](a)
    M = np.stack([sp_expm(hat(v)) for v in a])
    assert np.abs(to_matrix(X) - M).max() < 1e-12                     # se3.h:134-142, so3.h:153-190
    assert quat_close(X[:, 3:], Rotation.from_rotvec(a[:, 3:]).as_quat(), 1e-12)
    assert np.abs(np.linalg.norm(X[:, 3:], axis=1) - 1).max() < 1e-14


@pytest.mark.parametrize("impl", IMPLS)
def test_log_matches_scipy(impl):
    f = impl()
    a = 0.8 * RNG.standard_normal((B, 6))
    a[:8] *= 1e-7
    nrm = np.linalg.norm(a[:, 3:], axis=1, keepdims=True)
    a[:, 3:] *= np.minimum(1.0, 3.0 / np.maximum(nrm, 1e-300))        # angles below pi: the logarithm is then the vector itself
    R = Rotation.from_rotvec(a[:, 3:])
    M = np.stack([sp_expm(hat(v)) for v in a])
    X = np.concatenate([M[:, :3, 3], R.as_quat()], 1)
    got = f["log"](X)
    assert np.abs(got[:, 3:] - R.as_rotvec()).max() < f["log_tol"]    # so3.h:115-151
    L = np.stack([np.real(sp_logm(m)) for m in M[8:]])                # scipy's generic matrix logarithm (not for the tiny angles)
    assert np.abs(got[8:, :3] - L[:, :3, 3]).max() < 1e-8             # se3.h:104-132
    assert np.abs(got - a).max() < f["log_tol"]


@pytest.mark.parametrize("impl", IMPLS)
def test_compose_inverse_and_action_match_scipy(impl):
    f = impl()
    X, Y = f["exp"](tangents()), f["exp"](tangents())
    MX, MY = to_matrix(X), to_matrix(Y)
    assert np.abs(to_matrix(f["mul"](X, Y)) - MX @ MY).max() < 1e-12                       # se3.h:36-47
    q = (Rotation.from_quat(X[:, 3:]) * Rotation.from_quat(Y[:, 3:])).as_quat()
    assert quat_close(f["mul"](X, Y)[:, 3:], q, 1e-12)
    assert np.abs(to_matrix(f["inv"](X)) - np.linalg.inv(MX)).max() < 1e-12                # se3.h:49-51
    assert quat_close(f["inv"](X)[:, 3:], Rotation.from_quat(X[:, 3:]).inv().as_quat(), 1e-12)
    p3 = RNG.standard_normal((B, 3))
    assert np.abs(f["act"](X, p3) - (Rotation.from_quat(X[:, 3:]).apply(p3) + X[:, :3])).max() < 1e-12   # so3.h:55-60
    p4 = RNG.standard_normal((B, 4))
    assert np.abs(f["act"](X, p4) - np.einsum("bij,bj->bi", MX, p4)).max() < 1e-12        # se3.h:53-56


@pytest.mark.parametrize("impl", IMPLS)
def test_adjoint_transpose_matches_the_matrix_built_from_scipy_rotations(impl):
    f = impl()
    X = f["exp"](tangents())
    R = Rotation.from_quat(X[:, 3:]).as_matrix()
    t = X[:, :3]
    tx = np.zeros((B, 3, 3))
    tx[:, 0, 1], tx[:, 0, 2], tx[:, 1, 0], tx[:, 1, 2], tx[:, 2, 0], tx[:, 2, 1] = -t[:, 2], t[:, 1], t[:, 2], -t[:, 0], -t[:, 1], t[:, 0]
    Ad = np.zeros((B, 6, 6))
    Ad[:, :3, :3] = R; Ad[:, :3, 3:] = tx @ R; Ad[:, 3:, 3:] = R     # se3.h:58-67
    a = RNG.standard_normal((B, 6))
    assert np.abs(f["adjT"](X, a) - np.einsum("bji,bj->bi", Ad, a)).max() < 1e-12
    # and Ad is THE adjoint: X Exp(a) X^-1 == Exp(Ad a), checked with scipy's exponential only
    s = 0.3 * a
    lhs = to_matrix(X) @ np.stack([sp_expm(hat(v)) for v in s]) @ np.linalg.inv(to_matrix(X))
    rhs = np.stack([sp_expm(hat(v)) for v in np.einsum("bij,bj->bi", Ad, s)])
    assert np.abs(lhs - rhs).max() < 1e-12
