"""HIP counterpart of the reference's compiled module `lietorch_backends` for the SE3 group
(forward operations), same call convention:

    expm logm inv mul adj adjT act act4 as_matrix (group_id, *tensors) -> Tensor
        /root/reference/main/backend/lietorch/src/lietorch.cpp:286-316
        bound by /root/reference/main/backend/lietorch/group_ops.py:28-66

Inputs: contiguous 2-D `[B, dim]` float32/float64 tensors on the ROCm device (the reference's
TORCH_CHECKs, lietorch.cpp:7,19); output freshly allocated.  group_id must be 3 (SE3): SO3, RxSO3,
Sim3 and every `*_backward` are outside the BA-Track inference path and raise.
Kernels: batrack_amd/csrc/se3_kernels.hip through include/batrack_se3.h.  No CPU fallback.
"""
import torch

from .. import _lib

SE3_ID = 3
_DT = {torch.float32: 0, torch.float64: 1}


def _check(gid, *ts):
    if gid != SE3_ID:
        raise NotImplementedError("batrack_amd.lietorch_backends implements the SE3 group (group_id 3) only")
    dt = ts[0].dtype
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("lietorch_backends: tensors must be on the GPU (no CPU fallback)")
        if not t.is_contiguous() or t.dim() != 2:
            raise RuntimeError("lietorch_backends: inputs must be contiguous [B, dim] tensors")   # lietorch.cpp:7,19
        if t.dtype != dt or dt not in _DT:
            raise TypeError("lietorch_backends: float32 or float64, all inputs alike")
        if t.shape[0] != ts[0].shape[0]:
            raise ValueError("lietorch_backends: batch sizes differ (broadcast before the call, broadcasting.py:9-31)")
    return _DT[dt]


def _run(name, out_dim, gid, *ts, dims):
    dt = _check(gid, *ts)
    for t, d in zip(ts, dims):
        if t.shape[1] != d:
            raise ValueError(f"lietorch_backends.{name}: expected last dim {d}, got {t.shape[1]}")
    B = ts[0].shape[0]
    out = torch.empty((B, out_dim), dtype=ts[0].dtype, device=ts[0].device)
    fn = getattr(_lib.lib(), "bt_se3_" + name)
    st = torch.cuda.current_stream(ts[0].device).cuda_stream
    _lib.check(fn(*[t.data_ptr() for t in ts], out.data_ptr(), B, dt, st), "bt_se3_" + name)
    return out


def expm(gid, a):
    return _run("exp", 7, gid, a, dims=(6,))


def logm(gid, X):
    return _run("log", 6, gid, X, dims=(7,))


def inv(gid, X):
    return _run("inv", 7, gid, X, dims=(7,))


def mul(gid, X, Y):
    return _run("mul", 7, gid, X, Y, dims=(7, 7))


def act(gid, X, p):
    return _run("act", 3, gid, X, p, dims=(7, 3))


def act4(gid, X, p):
    return _run("act4", 4, gid, X, p, dims=(7, 4))


def adj(gid, X, a):
    return _run("adj", 6, gid, X, a, dims=(7, 6))


def adjT(gid, X, a):
    return _run("adjT", 6, gid, X, a, dims=(7, 6))


def as_matrix(gid, X):
    return _run("matrix", 16, gid, X, dims=(7,)).view(-1, 4, 4)


def _absent(*_a, **_k):
    raise NotImplementedError("backward / projector / Jinv are outside the BA-Track inference path (SURVEY.md §2 row 4)")


expm_backward = logm_backward = inv_backward = mul_backward = adj_backward = adjT_backward = _absent
act_backward = act4_backward = Jinv = projector = _absent
