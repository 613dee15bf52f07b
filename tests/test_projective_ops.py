"""batrack_amd.backend.projective_ops (row f-3) against the golden vectors of the reference's
transform(jacobian=True) and against the oracle's per-edge quantities — CPU, float64."""
import os

import numpy as np
import pytest
import torch

import oracle
from batrack_amd.backend import projective_ops as pops
from oracle.se3_torch import SE3Ref as SE3     # CPU float64 pose arithmetic: the torch formulas, not the product's HIP kernels

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def load(name):
    d = dict(np.load(os.path.join(GOLD, name + ".npz")))
    T = lambda a: torch.as_tensor(a, dtype=torch.float64)
    poses = SE3(T(d["poses"])[None])
    patches = T(d["patches"])[None, :, :, None, None]
    intr = T(d["intrinsics"])[None]
    ii, jj, kk = (torch.as_tensor(d[k]) for k in ("ii", "jj", "kk"))
    return d, poses, patches, intr, ii, jj, kk


@pytest.mark.parametrize("name", ["c1", "c1_rough"])
def test_transform_jacobian_matches_reference(name):
    d, poses, patches, intr, ii, jj, kk = load(name)
    coords, valid, (Ji, Jj, Jz) = pops.transform(poses, patches, intr, ii, jj, kk, jacobian=True)
    assert rel(coords[0, :, 0, 0], d["tf64.coords"]) < 1e-12
    assert np.array_equal(valid[0].numpy(), d["tf64.valid"])
    assert rel(Ji[0], d["tf64.Ji"]) < 1e-11
    assert rel(Jj[0], d["tf64.Jj"]) < 1e-12
    assert rel(Jz[0, :, :, 0], d["tf64.Jz"]) < 1e-12


def test_transform_variants_and_point_cloud():
    d, poses, patches, intr, ii, jj, kk = load("c1")
    x2 = pops.transform(poses, patches, intr, ii, jj, kk)
    x3, v = pops.transform(poses, patches, intr, ii, jj, kk, depth=True, valid=True)
    assert x2.shape[-1] == 2 and x3.shape[-1] == 3 and torch.allclose(x2, x3[..., :2])
    o = oracle.edges(d["poses"], d["patches"], d["intrinsics"], d["targets3"], d["weights"], d["ii"], d["jj"], d["kk"], d["bounds"])
    assert rel(x2[0, :, 0, 0], o["coords"]) < 1e-12
    # self reprojection is the identity; translation-only motion keeps the rotation out
    same = pops.transform(poses, patches, intr, ii, ii, kk)
    assert torch.allclose(same[0, :, 0, 0], patches[0, kk, :2, 0, 0], atol=1e-9)
    fm = pops.flow_mag(poses, patches, intr, ii, jj, kk)
    assert fm.shape[:2] == (1, len(ii)) and bool((fm >= 0).all())
    # point cloud: world point reprojects onto its own pixel
    ix = (torch.arange(patches.shape[1]) // 32).clamp(max=poses.data.shape[1] - 1)     # C1: 32 patches per frame
    pc = pops.point_cloud(poses, patches, intr, ix)
    back = poses[:, ix, None, None] * pc
    uv = pops.proj(back, intr[:, ix])
    m = patches[0, :, 2, 0, 0] > 0
    assert torch.allclose(uv[0, m, 0, 0], patches[0, m, :2, 0, 0], atol=1e-8)


def test_back_proj_roundtrip():
    B, N = 2, 50
    g = torch.Generator().manual_seed(0)
    K = torch.tensor([[500.0, 510.0, 320.0, 240.0]]).repeat(B, 1)
    xy = torch.rand(B, N, 2, generator=g) * 400 + 50
    depth = torch.rand(B, N, 1, generator=g) * 5 + 1
    T = SE3.exp(0.2 * torch.randn(B, 6, generator=g)).matrix().float()
    Pw = pops.back_proj(xy, depth, K, torch.linalg.inv(T))
    uv = pops.proj_to_frames(Pw, K[:, None], T[:, None])
    assert torch.allclose(uv[:, 0], xy, atol=1e-2)
