"""Stand-in for the reference's compiled `lietorch_backends` module (Eigen 3.4 +
CUDA, un-buildable here: SURVEY.md §8c), used ONLY by tests/golden/make_golden.py.

Only the SE3 (group_id == 3) forward ops the BA path reaches are implemented —
inv, mul, act4, adjT, expm — as our own pure-torch statement of the published
formulas the reference's headers implement (lietorch/include/so3.h:31-65,153-190,
se3.h:36-67,134-142, common.h:7): unit quaternion normalised on every load and
after every product, tangent order (tau, phi), EPS = 1e-6.  The arithmetic inside
these stand-ins is therefore NOT pinned by the reference binary, only by the
algebraic identities its run_tests.py checks (tests/test_se3_identities.py).
"""
import torch

EPS = 1e-6


def _need_se3(gid):
    if gid != 3:
        raise NotImplementedError("stub covers SE3 (group_id 3) only")


def _unit(q):
    return q / q.norm(dim=-1, keepdim=True)


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz], -1)


def _qrot(q, p):
    qv, w = q[..., :3], q[..., 3:]
    uv = torch.linalg.cross(qv, p)
    uv = uv + uv
    return p + w * uv + torch.linalg.cross(qv, uv)


def _qconj(q):
    return torch.cat([-q[..., :3], q[..., 3:]], -1)


def _split(X):
    return X[..., :3], _unit(X[..., 3:7])


def inv(gid, X):
    _need_se3(gid)
    t, q = _split(X)
    qi = _unit(_qconj(q))
    return torch.cat([-_qrot(qi, t), qi], -1)


def mul(gid, X, Y):
    _need_se3(gid)
    t1, q1 = _split(X)
    t2, q2 = _split(Y)
    return torch.cat([t1 + _qrot(q1, t2), _unit(_qmul(q1, q2))], -1)


def act4(gid, X, p):
    _need_se3(gid)
    t, q = _split(X)
    return torch.cat([_qrot(q, p[..., :3]) + t * p[..., 3:], p[..., 3:]], -1)


def adjT(gid, X, a):
    _need_se3(gid)
    t, q = _split(X)
    qi = _qconj(q)
    atau, aphi = a[..., :3], a[..., 3:]
    return torch.cat([_qrot(qi, atau),
                      _qrot(qi, torch.linalg.cross(atau, t) + aphi)], -1)


def expm(gid, a):
    _need_se3(gid)
    tau, phi = a[..., :3], a[..., 3:]
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = th2.sqrt()
    small = th < EPS
    ths = torch.where(small, torch.ones_like(th), th)
    th4 = th2 * th2
    imag = torch.where(small, 0.5 - th2 / 48.0 + th4 / 3840.0, torch.sin(0.5 * ths) / ths)
    real = torch.where(small, 1.0 - th2 / 8.0 + th4 / 384.0, torch.cos(0.5 * ths))
    q = _unit(torch.cat([imag * phi, real], -1))
    c1 = torch.where(small, 0.5 - th2 / 24.0, (1.0 - torch.cos(ths)) / (ths * ths))
    c2 = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - torch.sin(ths)) / (ths * ths * ths))
    pxt = torch.linalg.cross(phi, tau)
    t = tau + c1 * pxt + c2 * torch.linalg.cross(phi, pxt)
    return torch.cat([t, q], -1)


def _absent(*_a, **_k):
    raise NotImplementedError("not on the BA path; stub does not provide it")


# names group_ops.py binds at import time (group_ops.py:28-66)
expm_backward = logm = logm_backward = inv_backward = mul_backward = _absent
adj = adj_backward = adjT_backward = act = act_backward = act4_backward = _absent
Jinv = as_matrix = projector = _absent
