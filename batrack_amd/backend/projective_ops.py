"""Projective geometry helpers with the reference's names and argument meaning
(/root/reference/main/backend/projective_ops.py:19-175), used by the caller outside the BA
step: reprojection for map filtering (batrack.py:337), point clouds (:823-850,891), flow
magnitude for keyframing (:1017), back-projection helpers (:440-518).

Row f-3 of SURVEY.md §8.  Tensor layout follows the caller: poses SE3 [1,N,7], patches
[1,P,3,p,p] = (x, y, inverse depth), intrinsics [1,N,4] = (fx fy cx cy), index vectors
ii/jj/kk [E].  The SE3 group operations are those of the pose object handed in (the HIP kernels of
se3_kernels.hip for batrack_amd's SE3); the pinhole arithmetic around them is plain tensor code, as in the reference.  The Jacobian variant
is kept for API completeness and for tests against the golden vectors — inside the BA step the
same quantities are produced by k_tile without being materialised.
"""
import ctypes

import torch

MIN_DEPTH = 0.2


def _fused_reproject(poses, patches, intrinsics, ii, jj, kk, depth, valid, tonly):
    """One launch of bt_reproject (include/batrack_projective.h) for float32 data on the GPU with batch 1."""
    from .. import _lib
    L = _lib.lib()
    P = poses.data[0].contiguous()
    pat = patches[0].contiguous()
    K = intrinsics[0].contiguous()
    idx = [t.contiguous() for t in (ii, jj, kk)]
    E, ph, pw = idx[0].numel(), pat.shape[-2], pat.shape[-1]
    no = 3 if depth else 2
    coords = torch.empty(1, E, ph, pw, no, dtype=torch.float32, device=pat.device)
    val = torch.empty(1, E, ph, pw, dtype=torch.float32, device=pat.device) if valid else None
    st = torch.cuda.current_stream(pat.device).cuda_stream
    _lib.check(L.bt_reproject(P.data_ptr(), P.shape[0], pat.data_ptr(), pat.shape[0], ph * pw, K.data_ptr(),
                              idx[0].data_ptr(), idx[1].data_ptr(), idx[2].data_ptr(), E,
                              (1 if depth else 0) | (2 if tonly else 0), coords.data_ptr(),
                              val.data_ptr() if valid else None, st), "bt_reproject")
    return (coords, val) if valid else coords


def _can_fuse(poses, patches, intrinsics, ii, jj, kk):
    d = poses.data
    return (d.is_cuda and d.dtype == torch.float32 and patches.dtype == torch.float32 and intrinsics.dtype == torch.float32
            and d.dim() == 3 and d.shape[0] == 1 and patches.dim() == 5 and patches.shape[0] == 1 and patches.shape[2] == 3
            and intrinsics.dim() == 3 and intrinsics.shape[0] == 1 and intrinsics.shape[1] == d.shape[1]
            and all(t.dtype == torch.int64 and t.is_cuda and t.dim() == 1 for t in (ii, jj, kk)))


def coords_grid(ht, wd, **kwargs):
    ys, xs = torch.meshgrid(torch.arange(ht).to(**kwargs).float(), torch.arange(wd).to(**kwargs).float(), indexing="ij")
    return torch.stack([xs, ys], dim=-1)


def _split_intrinsics(intrinsics):
    """[..., 4] -> four tensors broadcastable over the patch window dims [..., 1, 1]."""
    k = intrinsics[..., None, None, :]
    return k[..., 0], k[..., 1], k[..., 2], k[..., 3]


def iproj(patches, intrinsics):
    """Pixel + inverse depth -> homogeneous point (X/Z, Y/Z, 1, 1/Z)      (projective_ops.py:19-29)."""
    x, y, d = patches[:, :, 0], patches[:, :, 1], patches[:, :, 2]
    fx, fy, cx, cy = _split_intrinsics(intrinsics)
    return torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(d), d], dim=-1)


def proj(X, intrinsics, depth=False):
    """Homogeneous point -> pixel (and projected inverse depth); Z clamped at 1e-2   (:32-52)."""
    fx, fy, cx, cy = _split_intrinsics(intrinsics)
    inv_z = 1.0 / X[..., 2].clamp(min=1e-2)
    u = fx * (inv_z * X[..., 0]) + cx
    v = fy * (inv_z * X[..., 1]) + cy
    if depth:
        return torch.stack([u, v, inv_z * X[..., 3]], dim=-1)
    return torch.stack([u, v], dim=-1)


def transform(poses, patches, intrinsics, ii, jj, kk, depth=False, valid=False, jacobian=False, tonly=False, fused=True):
    """Reproject patches kk from frame ii into frame jj                                    (:54-105).

    jacobian=True additionally returns the validity mask Z > 0.2 and (Ji, Jj, Jz), the derivatives of
    the patch-centre pixel w.r.t. a left perturbation of pose i, pose j, and the inverse depth.

    Without Jacobians, float32 data on the GPU takes the fused kernel (one launch instead of ~25); `fused=False`
    forces the composed tensor operations (tests compare the two)."""
    if not jacobian and fused and _can_fuse(poses, patches, intrinsics, ii, jj, kk):
        return _fused_reproject(poses, patches, intrinsics, ii, jj, kk, depth, valid, tonly)
    X0 = iproj(patches[:, kk], intrinsics[:, ii])
    Gij = poses[:, jj] * poses[:, ii].inv()
    if tonly:                                                  # translation-only motion (flow_mag)
        data = Gij.data.clone()
        data[..., 3:] = torch.as_tensor([0.0, 0.0, 0.0, 1.0], dtype=data.dtype, device=data.device)
        Gij = type(Gij)(data)
    X1 = Gij[:, :, None, None] * X0
    x1 = proj(X1, intrinsics[:, jj], depth)

    if jacobian:
        c = X1.shape[2] // 2
        X, Y, Z, H = X1[..., c, c, :].unbind(dim=-1)
        fx, fy = intrinsics[:, jj, 0], intrinsics[:, jj, 1]
        dz = torch.where(Z.abs() > MIN_DEPTH, 1.0 / Z, torch.zeros_like(Z))
        zero = torch.zeros_like(Z)
        # d(pixel)/d(point): 2x3 (the homogeneous 4th column is zero)
        Jp = torch.stack([torch.stack([fx * dz, zero, -fx * X * dz * dz], -1),
                          torch.stack([zero, fy * dz, -fy * Y * dz * dz], -1)], -2)            # [1,E,2,3]
        # d(point)/d(xi_j): [H*I | -[X]x]
        Ja = torch.stack([torch.stack([H, zero, zero, zero, Z, -Y], -1),
                          torch.stack([zero, H, zero, -Z, zero, X], -1),
                          torch.stack([zero, zero, H, Y, -X, zero], -1)], -2)                  # [1,E,3,6]
        Jj = Jp @ Ja
        Ji = -Gij[:, :, None].adjT(Jj)
        t_ij = Gij.data[..., :3]
        Jz = (Jp @ t_ij[..., None])                                                            # [1,E,2,1]
        return x1, (Z > MIN_DEPTH).to(x1.dtype), (Ji, Jj, Jz)

    if valid:
        return x1, (X1[..., 2] > MIN_DEPTH).to(x1.dtype)
    return x1


def point_cloud(poses, patches, intrinsics, ix):
    """World-frame homogeneous points of the patches                                        (:107-109)."""
    return poses[:, ix, None, None].inv() * iproj(patches, intrinsics[:, ix])


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta=0.3):
    """Blend of full-motion and translation-only flow magnitude, for keyframe selection     (:112-122)."""
    c0 = transform(poses, patches, intrinsics, ii, ii, kk)
    c1 = transform(poses, patches, intrinsics, ii, jj, kk, tonly=False)
    c2 = transform(poses, patches, intrinsics, ii, jj, kk, tonly=True)
    return beta * (c1 - c0).norm(dim=-1) + (1 - beta) * (c2 - c0).norm(dim=-1)


def back_proj(xy, xy_depth, intrinsics, cams_c2w=None):
    """Pixels [B,N,2] + depth [B,N,1] -> homogeneous points [B,N,4], optionally to world      (:129-149)."""
    fx, fy, cx, cy = (intrinsics[:, k, None] for k in range(4))
    D = xy_depth[..., 0]
    P = torch.stack([(xy[..., 0] - cx) / fx * D, (xy[..., 1] - cy) / fy * D, D, torch.ones_like(D)], dim=2)
    if cams_c2w is not None:
        P = (cams_c2w.float() @ P.transpose(1, 2)).transpose(1, 2)
    return P


def proj_to_frames(P, intrinsics, cams_w2c):
    """World points [B,N,4] into S cameras [B,S,4,4] -> pixels [B,S,N,2]                       (:151-175)."""
    Pc = (cams_w2c.float() @ P[:, None].transpose(2, 3)).transpose(2, 3)        # [B,S,N,4]
    fx, fy, cx, cy = (intrinsics[..., [k]] for k in range(4))
    inv = 1.0 / Pc[..., 2]
    return torch.stack([fx * (Pc[..., 0] * inv) + cx, fy * (Pc[..., 1] * inv) + cy], dim=-1)
