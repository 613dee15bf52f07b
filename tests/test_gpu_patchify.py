"""-m gpu: altcorr.patchify (row f-2) against a numpy restatement of the reference's gather
(correlation_kernel.cu:16-47) and blend (correlation.py:49-68).  The reference's extension
cannot be built here (CUDA), so this row is pinned by the restatement only.  Bit-exact."""
import numpy as np
import pytest
import torch

from batrack_amd.backend.altcorr import patchify

pytestmark = pytest.mark.gpu


def ref_patchify(net, coords, R, mode="bilinear"):
    B, C, H, W = net.shape
    M = coords.shape[1]
    D = 2 * R + 2
    pat = np.zeros((B, M, C, D, D), np.float32)
    fl = np.floor(coords).astype(np.int64)
    for b in range(B):
        for m in range(M):
            for a in range(D):
                for e in range(D):
                    i, j = fl[b, m, 1] + a - R, fl[b, m, 0] + e - R
                    if 0 <= i < H and 0 <= j < W:
                        pat[b, m, :, a, e] = net[b, :, i, j]
    if mode != "bilinear":
        return pat
    off = (coords - np.floor(coords)).astype(np.float32)
    dx, dy = off[..., 0][:, :, None, None, None], off[..., 1][:, :, None, None, None]
    d = 2 * R + 1
    one = np.float32(1)
    x00 = ((one - dy) * (one - dx)) * pat[..., :d, :d]
    x01 = ((one - dy) * dx) * pat[..., :d, 1:]
    x10 = (dy * (one - dx)) * pat[..., 1:, :d]
    x11 = (dy * dx) * pat[..., 1:, 1:]
    return x00 + x01 + x10 + x11


@pytest.mark.parametrize("R,mode", [(0, "bilinear"), (1, "bilinear"), (0, "nearest"), (2, "nearest")])
def test_patchify_matches_restatement(R, mode):
    rng = np.random.default_rng(R + len(mode))
    B, C, H, W, M = 2, 3, 37, 53, 200
    net = rng.standard_normal((B, C, H, W)).astype(np.float32)
    coords = np.stack([rng.uniform(-3, W + 3, (B, M)), rng.uniform(-3, H + 3, (B, M))], -1).astype(np.float32)
    coords[0, :4] = [[0, 0], [W - 1, H - 1], [10.0, 20.0], [W - 0.5, H - 0.5]]      # borders, integer coordinates
    out = patchify(torch.as_tensor(net).cuda(), torch.as_tensor(coords).cuda(), R, mode=mode).cpu().numpy()
    ref = ref_patchify(net, coords, R, mode)
    assert out.shape == ref.shape
    assert np.array_equal(out, ref)


def test_patchify_caller_shapes():
    """As used by the caller: colour at (coords + 0.5) with radius 0, depth per point (batrack.py:323,438)."""
    img = torch.rand(1, 3, 48, 64, device="cuda") * 255
    coords = torch.rand(1, 100, 2, device="cuda") * torch.tensor([63.0, 47.0], device="cuda")
    clr = patchify(img, coords + 0.5, 0).view(1, -1, 3)
    assert clr.shape == (1, 100, 3)
    depth = torch.rand(1, 1, 48, 64, device="cuda") + 1
    dpt = patchify(depth, coords, 0).reshape(1, 100, 1)
    assert bool((dpt >= 1).all()) and bool((dpt <= 2).all())
    with pytest.raises(RuntimeError):
        patchify(img.cpu(), coords.cpu(), 0)
