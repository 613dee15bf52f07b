"""SE3 pose wrapper with the surface the BA caller uses
(/root/reference/main/backend/lietorch/groups.py:51-285, used at
/root/reference/main/batrack.py:864,883): `.data` [...,7] = (tx ty tz qx qy qz qw),
`.vec()`, indexing, `inv`, `*`, `exp`, `retr`, `matrix`, `act`, `adjT`, `log`.

Every group operation runs the HIP kernels of batrack_amd/csrc/se3_kernels.hip (through
`lietorch_backends`, the counterpart of the reference's compiled module) and therefore needs
float32/float64 data on the GPU: a CPU tensor raises, there is no host fallback in the product
(the torch statement of the formulas that the tests check these kernels against is part of
the test oracle, module se3_torch).  The BA hot path consumes `.data` directly inside its own kernels.
Conventions follow the reference's headers: unit quaternion renormalised on use, tangent =
(tau, phi), EPS = 1e-6 (lietorch/include/so3.h:31-65,153-190, se3.h:36-67,124-142, common.h:7).
"""
import torch

from . import lietorch_backends as lb


def _need_hip(x):
    """The element-wise kernels take GPU float32/float64 data; anything else is an error, not a slower path."""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.float64)):
        raise RuntimeError(f"batrack_amd SE3 operations run on the GPU in float32/float64 (got {x.device}, {x.dtype}); "
                           "there is no CPU implementation in this package")
    return x


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

    # ---- group operations (HIP kernels; lietorch_backends raises on CPU tensors)
    def inv(self):
        return SE3(lb.inv(3, _need_hip(self.data).reshape(-1, 7).contiguous()).view(self.data.shape))

    def mul(self, other):
        xf, yf, shape = _flat_pair(_need_hip(self.data), _need_hip(other.data))
        return SE3(lb.mul(3, xf, yf).view(shape + (7,)))

    def act(self, p):
        xf, pf, shape = _flat_pair(_need_hip(self.data), _need_hip(p).to(self.data.dtype))
        out = lb.act(3, xf, pf) if p.shape[-1] == 3 else lb.act4(3, xf, pf)
        return out.view(shape + (p.shape[-1],))

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
        xf, af, shape = _flat_pair(_need_hip(self.data), _need_hip(a).to(self.data.dtype))
        return lb.adjT(3, xf, af).view(shape + (6,))

    @classmethod
    def exp(cls, x):
        return cls(lb.expm(3, _need_hip(x).reshape(-1, 6).contiguous()).view(x.shape[:-1] + (7,)))

    def log(self):
        return lb.logm(3, _need_hip(self.data).reshape(-1, 7).contiguous()).view(self.data.shape[:-1] + (6,))

    def retr(self, a):
        return SE3.exp(a).mul(self)


def stack(group_list, dim):
    return SE3(torch.stack([g.data for g in group_list], dim=dim))


def cat(group_list, dim):
    return SE3(torch.cat([g.data for g in group_list], dim=dim))
