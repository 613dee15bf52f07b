"""Stand-in for the `pypose` package (requirements.txt:23 of the reference, absent from this image), restricted to what
/root/reference/main/global_refine/model/refine_net.py touches: an SE3 `LieTensor` ([..., 7] = tx ty tz qx qy qz qw, the
pypose layout) with `Inv`, `@` (composition, or action on [..., 3] points), `.tensor()`, plus `SE3`, `Parameter`,
`mat2SE3`.  The arithmetic is OUR restatement of the published SE3 formulas — what the golden vectors made through it pin
is refine_net.py, not pypose (stated in DESIGN.md)."""
import torch


def _qrot(q, p):
    qv, w = q[..., :3], q[..., 3:]
    uv = 2.0 * torch.linalg.cross(qv, p)
    return p + w * uv + torch.linalg.cross(qv, uv)


def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)


class LieTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, data):
        return torch.Tensor._make_subclass(cls, data.detach() if isinstance(data, torch.Tensor) else torch.as_tensor(data))

    def tensor(self):
        return self.as_subclass(torch.Tensor)

    def Inv(self):
        d = self.tensor()
        t, q = d[..., :3], d[..., 3:]
        qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
        return LieTensor(torch.cat([-_qrot(qi, t), qi], -1))

    def __matmul__(self, other):
        a = self.tensor()
        if isinstance(other, LieTensor):
            b = other.tensor()
            return LieTensor(torch.cat([a[..., :3] + _qrot(a[..., 3:], b[..., :3]), _qmul(a[..., 3:], b[..., 3:])], -1))
        dt = torch.promote_types(a.dtype, other.dtype)            # (refine_net.py:326 casts the points to float32)
        a, other = a.to(dt), other.to(dt)
        return _qrot(a[..., 3:], other) + a[..., :3]


def SE3(data):
    return LieTensor(data.tensor() if isinstance(data, LieTensor) else data)


def Parameter(x):
    return x


def mat2SE3(m):
    raise NotImplementedError("stand-in: construct poses as [.., 7] tensors")
