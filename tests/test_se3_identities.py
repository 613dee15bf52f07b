"""The algebraic identities the reference's own lietorch test script checks
(/root/reference/main/backend/lietorch/run_tests.py:16-52: exp/log, X X^-1, adjoint
commutation, act == matrix @ p) restated for our SE3 wrapper and for the stand-in
used to import the reference when generating golden vectors."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle.se3_torch import SE3Ref as SE3

torch.manual_seed(0)
HERE = os.path.dirname(os.path.abspath(__file__))


def rand_se3(n, scale=1.0):
    return SE3.exp(scale * torch.randn(n, 6, dtype=torch.float64))


def test_exp_log_roundtrip():
    a = 0.5 * torch.randn(64, 6, dtype=torch.float64)
    assert torch.allclose(SE3.exp(a).log(), a, atol=1e-8)
    tiny = 1e-9 * torch.randn(8, 6, dtype=torch.float64)
    assert torch.allclose(SE3.exp(tiny).log(), tiny, atol=1e-12)


def test_inverse_is_identity():
    X = rand_se3(32)
    I = (X * X.inv()).data
    ref = torch.zeros_like(I); ref[:, 6] = 1
    assert torch.allclose(I, ref, atol=1e-8)


def test_adjoint_commutation():
    """X Exp(a) == Exp(Ad_X a) X, with Ad applied through adjT on basis vectors."""
    X = rand_se3(16)
    a = 0.3 * torch.randn(16, 6, dtype=torch.float64)
    eye = torch.eye(6, dtype=torch.float64)
    AdT = torch.stack([X.adjT(eye[k].expand(16, 6)) for k in range(6)], dim=1)   # rows: Ad^T e_k  -> [16, 6(k), 6]
    Ad_a = torch.einsum("bkj,bk->bj", AdT.transpose(1, 2), a)                     # Ad a
    lhs = (X * SE3.exp(a)).data
    rhs = (SE3.exp(Ad_a) * X).data
    sign = torch.sign((lhs[:, 3:] * rhs[:, 3:]).sum(-1, keepdim=True))
    assert torch.allclose(lhs[:, :3], rhs[:, :3], atol=1e-8)
    assert torch.allclose(lhs[:, 3:], sign * rhs[:, 3:], atol=1e-8)


def test_act_matches_matrix():
    X = rand_se3(16)
    p = torch.randn(16, 4, dtype=torch.float64)
    assert torch.allclose(X.act(p), torch.einsum("bij,bj->bi", X.matrix(), p), atol=1e-8)
    p3 = torch.randn(16, 3, dtype=torch.float64)
    hom = torch.cat([p3, torch.ones(16, 1, dtype=torch.float64)], -1)
    assert torch.allclose(X.act(p3), X.act(hom)[:, :3], atol=1e-10)


def test_retraction_is_left_multiplication():
    X = rand_se3(8)
    a = 0.1 * torch.randn(8, 6, dtype=torch.float64)
    assert torch.allclose(X.retr(a).data, (SE3.exp(a) * X).data, atol=1e-12)


def test_reference_import_stub_agrees_with_wrapper():
    """tests/golden/refstubs/lietorch_backends.py (used ONLY to import the reference) and our
    SE3 wrapper are two statements of the same published formulas; they must agree."""
    spec = importlib.util.spec_from_file_location("lb_stub", os.path.join(HERE, "golden", "refstubs", "lietorch_backends.py"))
    stub = importlib.util.module_from_spec(spec); spec.loader.exec_module(stub)
    X, Y = rand_se3(32), rand_se3(32)
    a = torch.randn(32, 6, dtype=torch.float64)
    p = torch.randn(32, 4, dtype=torch.float64)
    assert torch.allclose(stub.inv(3, X.data), X.inv().data, atol=1e-12)
    assert torch.allclose(stub.mul(3, X.data, Y.data), (X * Y).data, atol=1e-12)
    assert torch.allclose(stub.act4(3, X.data, p), X.act(p), atol=1e-12)
    assert torch.allclose(stub.adjT(3, X.data, a), X.adjT(a), atol=1e-12)
    assert torch.allclose(stub.expm(3, 0.3 * a), SE3.exp(0.3 * a).data, atol=1e-12)
