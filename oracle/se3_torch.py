"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Plain-torch statement of the SE3 group formulas, device- and dtype-agnostic (float64 on the CPU in the tests):
the checker for the element-wise HIP kernels of batrack_amd/csrc/se3_kernels.hip (row f-1 of SURVEY.md §8)
and the pose arithmetic of the CPU-side caller-loop tests.  Restates the published lietorch formulas
(/root/reference/main/backend/lietorch/include/so3.h:31-65,153-190, se3.h:36-67,124-142, common.h:7):
data [...,7] = (tx ty tz qx qy qz qw), unit quaternion renormalised on use, tangent = (tau, phi), EPS = 1e-6.

PARITY UNPINNED against the reference's compiled lietorch (Eigen is not in this image, so lietorch_cpu.cpp
cannot be built): pinned by the identities of the reference's own test script (lietorch/run_tests.py:16-52,
tests/test_se3_identities.py) and against the independent numpy stand-in of tests/golden/refstubs.

`SE3Ref` has the surface of batrack_amd.backend.lietorch.SE3 (same tensor-like methods, same operator names);
batrack_amd/ never imports this module.
"""
import torch

EPS = 1e-6


def _unit(q):
    return q / q.norm(dim=-1, keepdim=True)


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw,
                        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def _qrot(q, p):
    qv, p = torch.broadcast_tensors(q[..., :3], p)
    uv = 2.0 * torch.linalg.cross(qv, p)
    return p + q[..., 3:] * uv + torch.linalg.cross(qv, uv)


def _tq(d):
    return d[..., :3], _unit(d[..., 3:7])


def inv(d):
    t, q = _tq(d)
    qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
    return torch.cat([-_qrot(qi, t), qi], -1)


def mul(a, b):
    t1, q1 = _tq(a)
    t2, q2 = _tq(b)
    t1, t2 = torch.broadcast_tensors(t1, t2)
    q1, q2 = torch.broadcast_tensors(q1, q2)
    return torch.cat([t1 + _qrot(q1, t2), _unit(_qmul(q1, q2))], -1)


def act(d, p):
    t, q = _tq(d)
    if p.shape[-1] == 3:
        return _qrot(q, p) + t
    xyz = _qrot(q, p[..., :3]) + t * p[..., 3:]
    return torch.cat([xyz, p[..., 3:].expand(xyz.shape[:-1] + (1,))], -1)


def adjT(d, a):
    t, q = _tq(d)
    qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
    at, ap = a[..., :3], a[..., 3:]
    at, tb = torch.broadcast_tensors(at, t)
    return torch.cat([_qrot(qi, at), _qrot(qi, torch.linalg.cross(at, tb) + ap)], -1)


def exp(x):
    tau, phi = x[..., :3], x[..., 3:]
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = th2.sqrt()
    small = th < EPS
    ths = torch.where(small, torch.ones_like(th), th)
    imag = torch.where(small, 0.5 - th2 / 48.0 + th2 * th2 / 3840.0, torch.sin(0.5 * ths) / ths)
    real = torch.where(small, 1.0 - th2 / 8.0 + th2 * th2 / 384.0, torch.cos(0.5 * ths))
    q = _unit(torch.cat([imag * phi, real], -1))
    c1 = torch.where(small, 0.5 - th2 / 24.0, (1.0 - torch.cos(ths)) / (ths * ths))
    c2 = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - torch.sin(ths)) / (ths * ths * ths))
    pxt = torch.linalg.cross(phi, tau)
    return torch.cat([tau + c1 * pxt + c2 * torch.linalg.cross(phi, pxt), q], -1)


def log(d):
    t, q = _tq(d)
    qv, w = q[..., :3], q[..., 3:]
    n2 = (qv * qv).sum(-1, keepdim=True)
    n = n2.sqrt()
    small = n2 < EPS * EPS
    ns = torch.where(small, torch.ones_like(n), n)
    ws = torch.where(w.abs() < EPS, torch.full_like(w, EPS), w)
    k_small = 2.0 / w - (2.0 / 3.0) * n2 / (w * w * w)
    k_tiny_w = torch.where(w > 0, torch.pi / ns, -torch.pi / ns)
    k_reg = 2.0 * torch.atan(ns / ws) / ns
    k = torch.where(small, k_small, torch.where(w.abs() < EPS, k_tiny_w, k_reg))
    phi = k * qv
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = th2.sqrt()
    half = 0.5 * th
    tsmall = th < EPS
    hs = torch.where(tsmall, torch.ones_like(half), half)
    c2 = torch.where(tsmall, torch.full_like(th, 1.0 / 12.0),
                     (1.0 - 2.0 * hs * torch.cos(hs) / (2.0 * torch.sin(hs))) / torch.where(tsmall, torch.ones_like(th2), th2))
    pxt = torch.linalg.cross(phi, t)
    tau = t - 0.5 * pxt + c2 * torch.linalg.cross(phi, pxt)
    return torch.cat([tau, phi], -1)


class SE3Ref:
    """Same surface as batrack_amd.backend.lietorch.SE3, every group operation by the torch formulas above."""
    group_name = "SE3"
    group_id = 3
    manifold_dim = 6
    embedded_dim = 7

    def __init__(self, data):
        self.data = data

    shape = property(lambda self: self.data.shape[:-1])
    device = property(lambda self: self.data.device)
    dtype = property(lambda self: self.data.dtype)

    def vec(self):
        return self.data

    def __getitem__(self, index):
        return SE3Ref(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def detach(self):
        return SE3Ref(self.data.detach())

    def view(self, dims):
        return SE3Ref(self.data.view(tuple(dims) + (7,)))

    def to(self, *a, **k):
        return SE3Ref(self.data.to(*a, **k))

    def cpu(self):
        return SE3Ref(self.data.cpu())

    @classmethod
    def Identity(cls, *batch_shape, **kw):
        if len(batch_shape) == 1 and isinstance(batch_shape[0], (tuple, list)):
            batch_shape = tuple(batch_shape[0])
        d = torch.zeros(tuple(batch_shape) + (7,), **kw)
        d[..., 6] = 1.0
        return cls(d)

    @classmethod
    def InitFromVec(cls, data):
        return cls(data)

    def inv(self):
        return SE3Ref(inv(self.data))

    def mul(self, other):
        return SE3Ref(mul(self.data, other.data))

    def act(self, p):
        return act(self.data, p)

    def __mul__(self, other):
        return self.mul(other) if isinstance(other, SE3Ref) else self.act(other)

    def matrix(self):
        eye = torch.eye(4, dtype=self.dtype, device=self.device)
        eye = eye.view([1] * (self.data.dim() - 1) + [4, 4])
        return SE3Ref(self.data[..., None, :]).act(eye).transpose(-1, -2)

    def translation(self):
        p = torch.zeros(self.data.shape[:-1] + (4,), dtype=self.dtype, device=self.device)
        p[..., 3] = 1.0
        return self.act(p)

    def adjT(self, a):
        return adjT(self.data, a)

    @classmethod
    def exp(cls, x):
        return cls(exp(x))

    def log(self):
        return log(self.data)

    def retr(self, a):
        return SE3Ref.exp(a).mul(self)
