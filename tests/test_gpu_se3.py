"""-m gpu: the element-wise SE3 HIP kernels (include/batrack_se3.h, row f-1 of SURVEY.md §8)
against the float64 torch formulas of oracle/se3_torch.py on the CPU, plus the identities of the
reference's own test script (lietorch/run_tests.py:16-52) evaluated on the device."""
import numpy as np
import pytest
import torch

from batrack_amd.backend import lietorch_backends as lb
from batrack_amd.backend.lietorch import SE3
from oracle.se3_torch import SE3Ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.manual_seed(1)


def cpu64(x):
    return x.detach().cpu().double()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float64, 1e-12)])
def test_ops_match_cpu_float64(dtype, tol):
    B = 4099
    a = (0.7 * torch.randn(B, 6, dtype=torch.float64)).to(dtype)
    a[:5] *= 1e-8                                            # small-angle branch
    X = SE3Ref.exp(cpu64(a))
    Y = SE3Ref.exp(0.5 * torch.randn(B, 6, dtype=torch.float64))
    p4 = torch.randn(B, 4, dtype=torch.float64)
    v6 = torch.randn(B, 6, dtype=torch.float64)
    g = lambda t: t.to(dtype).to(DEV).contiguous()
    Xg, Yg = g(X.data), g(Y.data)
    Xr, Yr = SE3Ref(cpu64(Xg)), SE3Ref(cpu64(Yg))                 # reference on the same rounded inputs
    rel = lambda got, ref: float((cpu64(got) - ref).norm() / ref.norm())
    assert rel(lb.expm(3, g(a)), SE3Ref.exp(cpu64(g(a))).data) < tol
    assert rel(lb.inv(3, Xg), Xr.inv().data) < tol
    assert rel(lb.mul(3, Xg, Yg), (Xr * Yr).data) < tol
    assert rel(lb.act4(3, Xg, g(p4)), Xr.act(cpu64(g(p4)))) < tol
    assert rel(lb.act(3, Xg, g(p4[:, :3])), Xr.act(cpu64(g(p4[:, :3])))) < tol
    assert rel(lb.adjT(3, Xg, g(v6)), Xr.adjT(cpu64(g(v6)))) < tol
    assert rel(lb.as_matrix(3, Xg), Xr.matrix()) < tol
    assert rel(lb.logm(3, Xg), Xr.log()) < (5e-5 if dtype == torch.float32 else 1e-9)
    # adj is the transpose-adjoint's adjoint: <Ad a, b> == <a, Ad^T b>
    b6 = torch.randn(B, 6, dtype=torch.float64)
    lhs = (cpu64(lb.adj(3, Xg, g(v6))) * cpu64(g(b6))).sum(-1)
    rhs = (cpu64(g(v6)) * cpu64(lb.adjT(3, Xg, g(b6)))).sum(-1)
    assert float((lhs - rhs).abs().max()) < (2e-4 if dtype == torch.float32 else 1e-10)


def test_reference_identities_on_device():
    B = 1000
    a = 0.5 * torch.randn(B, 6, dtype=torch.float64, device=DEV)
    X = SE3.exp(a)                                                        # HIP path (GPU tensor)
    assert torch.allclose(X.log(), a, atol=1e-8)                          # Log(Exp(a)) == a
    I = (X * X.inv()).data
    ref = torch.zeros_like(I); ref[:, 6] = 1
    assert torch.allclose(I, ref, atol=1e-8)                              # X X^-1 == identity
    b = 0.3 * torch.randn(B, 6, dtype=torch.float64, device=DEV)
    lhs = (X * SE3.exp(b)).data
    rhs = (SE3.exp(lb.adj(3, X.data, b)) * X).data                        # X Exp(b) == Exp(Ad_X b) X
    sign = torch.sign((lhs[:, 3:] * rhs[:, 3:]).sum(-1, keepdim=True))
    assert torch.allclose(lhs[:, :3], rhs[:, :3], atol=1e-8) and torch.allclose(lhs[:, 3:], sign * rhs[:, 3:], atol=1e-8)
    p = torch.randn(B, 4, dtype=torch.float64, device=DEV)
    assert torch.allclose(X.act(p), torch.einsum("bij,bj->bi", lb.as_matrix(3, X.data), p), atol=1e-8)


def test_wrapper_broadcasts_like_the_reference():
    """poses[:, jj] * poses[:, ii].inv(), Gij[:, :, None, None] * X0 (projective_ops.py:61-66)."""
    P = SE3.exp(0.3 * torch.randn(1, 12, 6, device=DEV))
    ii = torch.randint(0, 12, (40,), device=DEV); jj = torch.randint(0, 12, (40,), device=DEV)
    Gij = P[:, jj] * P[:, ii].inv()
    X0 = torch.randn(1, 40, 3, 3, 4, device=DEV)
    X1 = Gij[:, :, None, None] * X0
    assert tuple(X1.shape) == (1, 40, 3, 3, 4)
    Pc = SE3Ref(P.data.cpu().double())
    Gc = Pc[:, jj.cpu()] * Pc[:, ii.cpu()].inv()
    X1c = Gc[:, :, None, None] * X0.cpu().double()
    assert float((X1.cpu().double() - X1c).abs().max()) < 2e-5
    M = Gij.matrix()
    assert tuple(M.shape) == (1, 40, 4, 4)


def test_rejects_other_groups_and_cpu_tensors():
    x = torch.zeros(4, 7, device=DEV); x[:, 6] = 1
    with pytest.raises(NotImplementedError):
        lb.inv(4, x)                                         # Sim3
    with pytest.raises(RuntimeError):
        lb.inv(3, x.cpu())
    with pytest.raises(NotImplementedError):
        lb.inv_backward(3, x, x)
    with pytest.raises(RuntimeError):
        SE3(x.cpu()).inv()                                   # the wrapper has no host path either
    with pytest.raises(RuntimeError):
        SE3.exp(torch.zeros(2, 6))
