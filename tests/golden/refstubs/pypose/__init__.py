"""Stand-in for the `pypose` package (requirements.txt:23 of the reference, absent from this image), restricted to what
/root/reference/main/global_refine/model/refine_net.py touches: an SE3 `LieTensor` ([..., 7] = tx ty tz qx qy qz qw, the
pypose layout) with `Inv`, `@` (composition, or action on [..., 3] points), `.tensor()`, plus `SE3`, `Parameter`,
`mat2SE3`.  The arithmetic is OUR restatement of the published SE3 formulas — what the golden vectors made through it pin
is refine_net.py, not pypose (stated in DESIGN.md).

Gradients follow pypose's own convention as we read its source (pypose/lietensor/operation.py: SE3_Act, SE3_Mul, SE3_Inv
are torch.autograd.Functions whose backward returns the gradient of the LEFT perturbation Exp(delta) X, tangent order
(tau, phi), padded with one zero to the seven stored numbers):
    Act   q = X p :   g_X = (g, q x g, 0),  g_p = R^T g
    Mul   Z = X Y :   g_X = g_Z,  g_Y = g_Z Ad(X)            (row vector times the 6x6 adjoint [[R, [t]x R], [0, R]])
    Inv   Y = X^-1:   g_X = -g_Y Ad(Y)
`.tensor()` and indexing are plain views, so a loss that reads the stored numbers directly (cam_smooth_vec_loss,
refine_net.py:356-360) differentiates them as ordinary numbers.  **Parity unpinned** for this convention: pypose itself is
not here to check it."""
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


def _qinv(q):
    return torch.cat([-q[..., :3], q[..., 3:]], -1)


def _inv(d):
    qi = _qinv(d[..., 3:])
    return torch.cat([-_qrot(qi, d[..., :3]), qi], -1)


def _mul(a, b):
    return torch.cat([a[..., :3] + _qrot(a[..., 3:], b[..., :3]), _qmul(a[..., 3:], b[..., 3:])], -1)


def _row_times_adj(g, X):
    """g [..., 6] (row vector) times Ad(X), X [..., 7]:  (R^T g_tau, R^T (g_tau x t + g_phi))."""
    qi = _qinv(X[..., 3:])
    gt, gp = g[..., :3], g[..., 3:]
    return torch.cat([_qrot(qi, gt), _qrot(qi, torch.linalg.cross(gt, X[..., :3].expand_as(gt)) + gp)], -1)


def _pad(g6):
    return torch.cat([g6, torch.zeros_like(g6[..., :1])], -1)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, p):
        out = _qrot(X[..., 3:], p) + X[..., :3]
        ctx.save_for_backward(X, out)
        ctx.shapes = (X.shape, p.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        X, out = ctx.saved_tensors
        gX = _pad(torch.cat([g, torch.linalg.cross(out, g)], -1))
        gp = _qrot(_qinv(X[..., 3:]), g)
        return gX.sum_to_size(ctx.shapes[0]), gp.sum_to_size(ctx.shapes[1])


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Y):
        ctx.save_for_backward(X)
        ctx.shapes = (X.shape, Y.shape)
        return _mul(X, Y)

    @staticmethod
    def backward(ctx, g):
        (X,) = ctx.saved_tensors
        g6 = g[..., :6]
        return _pad(g6).sum_to_size(ctx.shapes[0]), _pad(_row_times_adj(g6, X)).sum_to_size(ctx.shapes[1])


class _Inv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        Y = _inv(X)
        ctx.save_for_backward(Y)
        return Y

    @staticmethod
    def backward(ctx, g):
        (Y,) = ctx.saved_tensors
        return _pad(-_row_times_adj(g[..., :6], Y))


class LieTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, data):
        return torch.as_tensor(data).as_subclass(cls)

    def tensor(self):
        return self.as_subclass(torch.Tensor)

    def Inv(self):
        return _Inv.apply(self.tensor()).as_subclass(LieTensor)

    def __matmul__(self, other):
        a = self.tensor()
        if isinstance(other, LieTensor):
            return _Mul.apply(a, other.tensor()).as_subclass(LieTensor)
        dt = torch.promote_types(a.dtype, other.dtype)            # (refine_net.py:326 casts the points to float32)
        return _Act.apply(a.to(dt), torch.as_tensor(other).as_subclass(torch.Tensor).to(dt))


def SE3(data):
    return torch.as_tensor(data).as_subclass(torch.Tensor).as_subclass(LieTensor)


def Parameter(x):
    """pp.Parameter: a LEAF LieTensor that requires grad (refine_net.py:45), so that `net.pose.grad` is filled."""
    return torch.Tensor._make_subclass(LieTensor, torch.as_tensor(x).as_subclass(torch.Tensor).detach().clone(), True)


def mat2SE3(m):
    """pp.mat2SE3 (refine_net.py:61): [..,4,4] (or [..,3,4]) matrices -> SE3 LieTensor [..,7] = (t, q_xyzw).  Stand-in: the
    quaternion by the trace / largest-diagonal-entry rule, normalised.  q and -q are the same pose; which of the two (and
    which branch near the switch-over) real pypose returns is NOT pinned by this stand-in."""
    m = torch.as_tensor(m)
    R, t = m[..., :3, :3], m[..., :3, 3]
    m00, m01, m02 = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    m10, m11, m12 = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    m20, m21, m22 = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    tr = m00 + m11 + m22

    def case(w, x, y, z, sq):
        return torch.stack([x, y, z, w], -1) / sq[..., None]
    s0 = torch.sqrt(torch.clamp(tr + 1.0, min=1e-30)) * 2.0
    q0 = case(0.25 * s0 * s0, m21 - m12, m02 - m20, m10 - m01, s0)
    s1 = torch.sqrt(torch.clamp(1.0 + m00 - m11 - m22, min=1e-30)) * 2.0
    q1 = case(m21 - m12, 0.25 * s1 * s1, m01 + m10, m02 + m20, s1)
    s2 = torch.sqrt(torch.clamp(1.0 + m11 - m00 - m22, min=1e-30)) * 2.0
    q2 = case(m02 - m20, m01 + m10, 0.25 * s2 * s2, m12 + m21, s2)
    s3 = torch.sqrt(torch.clamp(1.0 + m22 - m00 - m11, min=1e-30)) * 2.0
    q3 = case(m10 - m01, m02 + m20, m12 + m21, 0.25 * s3 * s3, s3)
    c1 = ((m00 > m11) & (m00 > m22))[..., None]
    c2 = (m11 > m22)[..., None]
    q = torch.where((tr > 0)[..., None], q0, torch.where(c1, q1, torch.where(c2, q2, q3)))
    q = q / q.norm(dim=-1, keepdim=True)
    return SE3(torch.cat([t, q], -1))
