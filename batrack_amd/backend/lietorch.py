"""SE3 pose wrapper with the surface the BA caller uses
(/root/reference/main/backend/lietorch/groups.py:51-285, used at
/root/reference/main/batrack.py:864,883): `.data` [...,7] = (tx ty tz qx qy qz qw),
`.vec()`, indexing, `inv`, `*`, `exp`, `retr`, `matrix`, `act`, `adjT`, `log`.

On the GPU the group operations run the HIP kernels of batrack_amd/csrc/se3_kernels.hip
(through `lietorch_backends`, the counterpart of the reference's compiled module); on
CPU tensors they fall back to the plain-torch formulas below (used by CPU tests and
tooling only — the BA hot path consumes `.data` directly inside its own kernels).  Conventions follow the reference's
headers: unit quaternion renormalised on use, tangent = (tau, phi), EPS = 1e-6
(lietorch/include/so3.h:31-65,153-190, se3.h:36-67,124-142, common.h:7).
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


def _hip(x):
    """HIP element-wise kernels apply to GPU float32/float64 data."""
    return x.is_cuda and x.dtype in (torch.float32, torch.float64)


def _flat_pair(x, y):
    """Broadcast two [..., d] tensors over their batch dims and flatten (broadcasting.py:9-31)."""
    shape = torch.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    xf = x.expand(shape + x.shape[-1:]).reshape(-1, x.shape[-1]).contiguous()
    yf = y.expand(shape + y.shape[-1:]).reshape(-1, y.shape[-1]).contiguous()
    return xf, yf, shape


class SE3:
    group_name = "SE3"
    group_id = 3
    manifold_dim = 6
    embedded_dim = 7

    def __init__(self, data):
        self.data = data

    # ---- tensor-like surface
    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    def __repr__(self):
        return f"SE3: size={tuple(self.shape)}, device={self.device}, dtype={self.dtype}"

    def vec(self):
        return self.data

    def __getitem__(self, index):
        return SE3(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def detach(self):
        return SE3(self.data.detach())

    def view(self, dims):
        return SE3(self.data.view(tuple(dims) + (7,)))

    def to(self, *a, **k):
        return SE3(self.data.to(*a, **k))

    def cpu(self):
        return SE3(self.data.cpu())

    def cuda(self):
        return SE3(self.data.cuda())

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

    # ---- group operations
    def _tq(self):
        return self.data[..., :3], _unit(self.data[..., 3:7])

    def inv(self):
        if _hip(self.data):
            from . import lietorch_backends as lb
            return SE3(lb.inv(3, self.data.reshape(-1, 7).contiguous()).view(self.data.shape))
        t, q = self._tq()
        qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
        return SE3(torch.cat([-_qrot(qi, t), qi], -1))

    def mul(self, other):
        if _hip(self.data) and _hip(other.data):
            from . import lietorch_backends as lb
            xf, yf, shape = _flat_pair(self.data, other.data)
            return SE3(lb.mul(3, xf, yf).view(shape + (7,)))
        t1, q1 = self._tq()
        t2, q2 = other._tq()
        t1, t2 = torch.broadcast_tensors(t1, t2)
        q1, q2 = torch.broadcast_tensors(q1, q2)
        return SE3(torch.cat([t1 + _qrot(q1, t2), _unit(_qmul(q1, q2))], -1))

    def act(self, p):
        if _hip(self.data) and _hip(p) and p.dtype == self.data.dtype:
            from . import lietorch_backends as lb
            xf, pf, shape = _flat_pair(self.data, p)
            out = lb.act(3, xf, pf) if p.shape[-1] == 3 else lb.act4(3, xf, pf)
            return out.view(shape + (p.shape[-1],))
        t, q = self._tq()
        if p.shape[-1] == 3:
            return _qrot(q, p) + t
        xyz = _qrot(q, p[..., :3]) + t * p[..., 3:]
        return torch.cat([xyz, p[..., 3:].expand(xyz.shape[:-1] + (1,))], -1)

    def __mul__(self, other):
        if isinstance(other, SE3):
            return self.mul(other)
        return self.act(other)

    def matrix(self):
        eye = torch.eye(4, dtype=self.dtype, device=self.device)
        eye = eye.view([1] * (self.data.dim() - 1) + [4, 4])
        return SE3(self.data[..., None, :]).act(eye).transpose(-1, -2)

    def translation(self):
        p = torch.zeros(self.data.shape[:-1] + (4,), dtype=self.dtype, device=self.device)
        p[..., 3] = 1.0
        return self.act(p)

    def adjT(self, a):
        if _hip(self.data) and _hip(a) and a.dtype == self.data.dtype:
            from . import lietorch_backends as lb
            xf, af, shape = _flat_pair(self.data, a)
            return lb.adjT(3, xf, af).view(shape + (6,))
        t, q = self._tq()
        qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
        at, ap = a[..., :3], a[..., 3:]
        at, tb = torch.broadcast_tensors(at, t)
        return torch.cat([_qrot(qi, at), _qrot(qi, torch.linalg.cross(at, tb) + ap)], -1)

    @classmethod
    def exp(cls, x):
        if _hip(x):
            from . import lietorch_backends as lb
            return cls(lb.expm(3, x.reshape(-1, 6).contiguous()).view(x.shape[:-1] + (7,)))
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
        return cls(torch.cat([tau + c1 * pxt + c2 * torch.linalg.cross(phi, pxt), q], -1))

    def log(self):
        if _hip(self.data):
            from . import lietorch_backends as lb
            return lb.logm(3, self.data.reshape(-1, 7).contiguous()).view(self.data.shape[:-1] + (6,))
        t, q = self._tq()
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

    def retr(self, a):
        return SE3.exp(a).mul(self)


def stack(group_list, dim):
    return SE3(torch.stack([g.data for g in group_list], dim=dim))


def cat(group_list, dim):
    return SE3(torch.cat([g.data for g in group_list], dim=dim))
