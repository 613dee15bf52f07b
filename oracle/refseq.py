"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

`refseq`: a torch-CPU restatement of one BA_rgbd_droid call that keeps the REFERENCE'S
OPERATOR SEQUENCE (/root/reference/main/backend/ba.py:217-339), i.e. what the reference
executes on a CPU device: per-edge gathers, the SE3 primitives as separate element-wise
passes (projective_ops.py:54-100: inv, mul, act4, adjT, the 4x4 matrix), `Ja` / `Jp` stacks
and their batched products, the nine batched 6x2 . 2x6 block products (ba.py:253-266), twelve
scatter-adds (ba.py:279-292), the DENSE E [n, m, 6], the dense-GEMM Schur complement
(ba.py:321-322), `cholesky_ex` + `cholesky_solve` (ba.py:5-19), the dense back-substitution
(ba.py:328) and the full-buffer retraction (ba.py:332-337).

It is the CPU baseline SURVEY.md §8(d) / BASELINE.md §3 specify (bench.py `cpu_baseline`, kind
"refseq"), validated against tests/golden/*.npz in tests/test_oracle_golden.py.  The C oracle
(oracle/ba_oracle_impl.h) is the edge-major scalar port used as the parity checker; this module
exists for timing the reference's own algorithmic structure on the GPU box's host cores.

The SE3 arithmetic restates the published formulas (lietorch include/se3.h:36-86,134-142,
so3.h:31-65,153-190); `torch_scatter.scatter_sum` is `index_add_` on a zero tensor.
"""
import torch


# ---------------------------------------------------------------- SE3 primitives, [.., 7] = tx ty tz qx qy qz qw
def _qnorm(q):
    return q / q.norm(dim=-1, keepdim=True)                                       # so3.h:35-37


def _qmul(a, b):                                                                    # so3.h:31-33, 51-53
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return _qnorm(torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                               aw * by - ax * bz + ay * bw + az * bx,
                               aw * bz + ax * by - ay * bx + az * bw,
                               aw * bw - ax * bx - ay * by - az * bz], -1))


def _qrot(q, p):                                                                    # so3.h:55-60
    qv, w = q[..., :3], q[..., 3:]
    uv = 2.0 * torch.linalg.cross(qv, p)
    return p + w * uv + torch.linalg.cross(qv, uv)


def se3_inv(G):                                                                     # se3.h:36-40
    t, q = G[..., :3], _qnorm(G[..., 3:])
    qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
    return torch.cat([-_qrot(qi, t), qi], -1)


def se3_mul(A, B):                                                                  # se3.h:45-47
    ta, qa = A[..., :3], _qnorm(A[..., 3:])
    tb, qb = B[..., :3], _qnorm(B[..., 3:])
    return torch.cat([ta + _qrot(qa, tb), _qmul(qa, qb)], -1)


def se3_act4(G, X):                                                                 # se3.h:53-56
    t, q = G[..., :3], _qnorm(G[..., 3:])
    return torch.cat([_qrot(q, X[..., :3]) + t * X[..., 3:], X[..., 3:]], -1)


def se3_adjT(G, a):                                                                 # se3.h:58-67, 84-86: Ad(G)^T a
    t, q = G[..., :3], _qnorm(G[..., 3:])
    qi = torch.cat([-q[..., :3], q[..., 3:]], -1)
    tau, phi = a[..., :3], a[..., 3:]
    return torch.cat([_qrot(qi, tau), _qrot(qi, torch.linalg.cross(tau, t) + phi)], -1)


def se3_matrix(G):                                                                  # groups.py:180-184: act4 on the identity columns
    I = torch.eye(4, dtype=G.dtype).view(*([1] * (G.dim() - 1)), 4, 4)
    return se3_act4(G[..., None, :], I.expand(*G.shape[:-1], 4, 4)).transpose(-1, -2)


def se3_exp(xi):                                                                    # se3.h:134-142, so3.h:153-190
    tau, phi = xi[..., :3], xi[..., 3:]
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = th2.sqrt()
    small = th < 1e-6                                                               # common.h:7
    ths = torch.where(small, torch.ones_like(th), th)
    imag = torch.where(small, 0.5 - th2 / 48.0, torch.sin(0.5 * ths) / ths)
    real = torch.where(small, 1.0 - th2 / 8.0, torch.cos(0.5 * ths))
    q = _qnorm(torch.cat([imag * phi, real], -1))
    c1 = torch.where(small, 0.5 - th2 / 24.0, (1.0 - torch.cos(ths)) / (ths * ths))
    c2 = torch.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - torch.sin(ths)) / (ths * ths * ths))
    pxt = torch.linalg.cross(phi, tau)
    return torch.cat([tau + c1 * pxt + c2 * torch.linalg.cross(phi, pxt), q], -1)


# ---------------------------------------------------------------- helpers of ba.py
def _scatter_sum(src, index, dim_size):                                             # torch_scatter.scatter_sum along dim 1
    out = torch.zeros(src.shape[0], dim_size, *src.shape[2:], dtype=src.dtype)
    return out.index_add_(1, index, src)


def _scatter_mat(A, ii, jj, n, m):                                                  # ba.py:33-35
    v = (ii >= 0) & (jj >= 0) & (ii < n) & (jj < m)
    return _scatter_sum(A[:, v], ii[v] * m + jj[v], n * m)


def _scatter_vec(b, ii, n):                                                         # ba.py:37-39
    v = (ii >= 0) & (ii < n)
    return _scatter_sum(b[:, v], ii[v], n)


def _block_matmul(A, B):                                                            # ba.py:52-58
    b, n1, m1, p1, q1 = A.shape
    _, n2, m2, p2, q2 = B.shape
    A2 = A.permute(0, 1, 3, 2, 4).reshape(b, n1 * p1, m1 * q1)
    B2 = B.permute(0, 1, 3, 2, 4).reshape(b, n2 * p2, m2 * q2)
    return torch.matmul(A2, B2).reshape(b, n1, p1, m2, q2).permute(0, 1, 3, 2, 4)


def _block_solve(A, B, ep, lm):                                                     # ba.py:60-70, 5-19
    b, n1, m1, p1, q1 = A.shape
    _, n2, m2, p2, q2 = B.shape
    A2 = A.permute(0, 1, 3, 2, 4).reshape(b, n1 * p1, m1 * q1)
    B2 = B.permute(0, 1, 3, 2, 4).reshape(b, n2 * p2, m2 * q2)
    A2 = A2 + (ep + lm * A2) * torch.eye(n1 * p1, dtype=A2.dtype)
    U, info = torch.linalg.cholesky_ex(A2)
    failed = bool(torch.any(info))
    X = torch.zeros_like(B2) if failed else torch.cholesky_solve(B2, U)
    return X.reshape(b, n1, p1, m2, q2).permute(0, 1, 3, 2, 4), failed


def _kernel_weight(r, loss):                                                        # ba.py:81-100
    if loss == "trivial":
        return torch.ones_like(r)
    s = r * r
    if loss == "huber":
        w = torch.ones_like(r)
        w[s > 1] = 1 / torch.sqrt(s)[s > 1]
        return w
    if loss == "cauchy":
        return 1 / (1 + s)
    raise NotImplementedError(loss)


# ---------------------------------------------------------------- the step
def ba_step(poses, patches, mono, intrinsics, targets3, weights, ii, jj, kk, bounds, fixedp=1,
            structure_only=False, loss="huber", lmbda=1e-4, ep=10.0, alpha=0.05, want_system=False):
    """poses [N,7], patches [P,3], mono [P], intrinsics [N,4], targets3 [E,3], weights [E,2] (torch CPU tensors of
    one floating dtype), ii/jj/kk int64 [E].  Returns a dict like oracle.ba_step."""
    dt = poses.dtype
    E = ii.numel()
    n_all = int(max(ii.max().item(), jj.max().item())) + 1                          # ba.py:219
    Gs, pat, K = poses[None], patches[None], intrinsics[None]
    tg, w_in = targets3[None, :, :2], weights[None]
    # ---- projective_ops.transform(jacobian=True), projective_ops.py:54-100
    x, y, d = pat[:, kk].unbind(-1)
    fx, fy, cx, cy = K[:, ii].unbind(-1)
    X0 = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(d), d], -1)     # iproj :19-29
    Gij = se3_mul(Gs[:, jj], se3_inv(Gs[:, ii]))                                    # :61
    X1 = se3_act4(Gij, X0)                                                          # :66
    X, Y, Z, H = X1.unbind(-1)
    fxj, fyj, cxj, cyj = K[:, jj].unbind(-1)
    dinv = 1.0 / Z.clamp(min=1e-2)                                                  # proj :43-45
    coords = torch.stack([fxj * (dinv * X) + cxj, fyj * (dinv * Y) + cyj], -1)
    o = torch.zeros_like(H)
    dj = torch.zeros_like(Z)
    big = Z.abs() > 0.2
    dj[big] = 1.0 / Z[big]                                                          # :80-81
    Ja = torch.stack([H, o, o, o, Z, -Y,
                      o, H, o, -Z, o, X,
                      o, o, H, Y, -X, o,
                      o, o, o, o, o, o], -1).view(1, E, 4, 6)                         # :83-88
    Jp = torch.stack([fxj * dj, o, -fxj * X * dj * dj, o,
                      o, fyj * dj, -fyj * Y * dj * dj, o], -1).view(1, E, 2, 4)       # :90-93
    Jj = torch.matmul(Jp, Ja)                                                       # :95
    Ji = -se3_adjT(Gij[:, :, None], Jj)                                             # :96
    Jz = torch.matmul(Jp, se3_matrix(Gij)[..., :, 3:])                              # :98
    v = (Z > 0.2).to(dt)                                                            # :100
    # ---- residual, validity, robust weights, ba.py:228-251
    r = tg - coords
    v = v * (r.norm(dim=-1) < 250).to(dt)
    inb = (coords[..., 0] > bounds[0]) & (coords[..., 1] > bounds[1]) & (coords[..., 0] < bounds[2]) & (coords[..., 1] < bounds[3])
    v = v * inb.to(dt)
    wts = w_in * _kernel_weight(r, loss)
    r = (v[..., None] * r).unsqueeze(-1)
    wts = (v[..., None] * wts).unsqueeze(-1)
    # ---- per-edge blocks, ba.py:253-266
    wJiT, wJjT, wJzT = (wts * Ji).transpose(2, 3), (wts * Jj).transpose(2, 3), (wts * Jz).transpose(2, 3)
    Bii, Bij = torch.matmul(wJiT, Ji), torch.matmul(wJiT, Jj)
    Bji, Bjj = torch.matmul(wJjT, Ji), torch.matmul(wJjT, Jj)
    Eik, Ejk = torch.matmul(wJiT, Jz), torch.matmul(wJjT, Jz)
    vi, vj = torch.matmul(wJiT, r), torch.matmul(wJjT, r)
    # ---- assembly, ba.py:268-292
    n = n_all - fixedp
    i2, j2 = ii - fixedp, jj - fixedp
    kx, k2 = torch.unique(kk, return_inverse=True, sorted=True)
    m = kx.numel()
    b = 1
    nn = max(n, 0)
    B = (_scatter_mat(Bii, i2, i2, nn, nn) + _scatter_mat(Bij, i2, j2, nn, nn) +
         _scatter_mat(Bji, j2, i2, nn, nn) + _scatter_mat(Bjj, j2, j2, nn, nn)).view(b, nn, nn, 6, 6)
    Em = (_scatter_mat(Eik, i2, k2, nn, m) + _scatter_mat(Ejk, j2, k2, nn, m)).view(b, nn, m, 6, 1)
    C = _scatter_vec(torch.matmul(wJzT, Jz), k2, m)
    vv = (_scatter_vec(vi, i2, nn) + _scatter_vec(vj, j2, nn)).view(b, nn, 1, 6, 1)
    w = _scatter_vec(torch.matmul(wJzT, r), k2, m)
    # ---- depth prior, Q, ba.py:296-311
    mono_kx = mono[None, kx, None, None]
    pm = (mono_kx > 1e-2).to(dt)
    if torch.is_tensor(lmbda):
        lmbda = lmbda.reshape(*C.shape)                                             # ba.py:299-300
    Cadj = C + pm * alpha
    Cadj = Cadj + lmbda
    wadj = w - pm * alpha * (pat[:, kx, 2, None, None] - mono_kx)
    Q = 1.0 / Cadj
    EQ = Em * Q[:, None]
    out = {"failed": False, "n": nn, "m": m}
    so = structure_only or n <= 0
    if so:                                                                           # ba.py:316-317
        dZ = (Q * wadj).view(b, -1)
        dX = None
    else:                                                                            # ba.py:320-330
        S = B - _block_matmul(EQ, Em.permute(0, 2, 1, 4, 3))
        yv = vv - _block_matmul(EQ, wadj.unsqueeze(2))
        dX, failed = _block_solve(S, yv, ep, 1e-4)
        if torch.isnan(dX).any():
            dX, failed = _block_solve(S, yv, ep, 1e-3)
        out["failed"] = failed
        dZ = (Q * (wadj - _block_matmul(Em.permute(0, 2, 1, 4, 3), dX).squeeze(-1))).view(b, -1)
        dX = dX.view(b, -1, 6)
        if want_system:
            out["S"] = S.permute(0, 1, 3, 2, 4).reshape(6 * nn, 6 * nn).clone()
            out["y"] = yv.reshape(-1).clone()
        out["dX"] = dX[0].clone()
    # ---- retraction, ba.py:332-337 (the whole buffers)
    disp = pat[..., 2] + _scatter_sum(dZ, kx, pat.shape[1])
    disp = disp.clamp(min=1e-3, max=10.0)
    out["patches_out"] = torch.stack([pat[0, :, 0], pat[0, :, 1], disp[0]], -1)
    if so:
        out["poses_out"] = poses
    else:
        delta = _scatter_sum(dX, fixedp + torch.arange(nn), Gs.shape[1])
        out["poses_out"] = se3_mul(se3_exp(delta), Gs)[0]
    return out
