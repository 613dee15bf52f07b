"""-m gpu: altcorr.patchify (row f-2) against the vectors the reference's own Python produced
(tests/golden/patchify.npz — blend by correlation.py:51-68 itself, gather by a stand-in for the CUDA
extension, which cannot be built here) and against oracle/patchify.py on larger inputs.  Bit-exact."""
import os

import numpy as np
import pytest
import torch

from batrack_amd.backend.altcorr import patchify
from oracle import patchify as op

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,mode", [(0, "bilinear"), (1, "bilinear"), (0, "nearest"), (2, "nearest")])
def test_patchify_matches_restatement(R, mode):
    rng = np.random.default_rng(R + len(mode))
    B, C, H, W, M = 2, 3, 37, 53, 200
    net = rng.standard_normal((B, C, H, W)).astype(np.float32)
    coords = np.stack([rng.uniform(-3, W + 3, (B, M)), rng.uniform(-3, H + 3, (B, M))], -1).astype(np.float32)
    coords[0, :4] = [[0, 0], [W - 1, H - 1], [10.0, 20.0], [W - 0.5, H - 0.5]]      # borders, integer coordinates
    out = patchify(torch.as_tensor(net).cuda(), torch.as_tensor(coords).cuda(), R, mode=mode).cpu().numpy()
    ref = op.patchify(net, coords, R, mode)
    assert out.shape == ref.shape
    assert np.array_equal(out, ref)


def test_patchify_matches_reference_vectors():
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "patchify.npz")))
    for n in range(5):
        R, mode = int(g[f"case{n}.R"]), str(g[f"case{n}.mode"])
        out = patchify(torch.as_tensor(g[f"case{n}.net"]).cuda(), torch.as_tensor(g[f"case{n}.coords"]).cuda(), R, mode=mode)
        assert np.array_equal(out.cpu().numpy(), g[f"case{n}.out"]), (n, R, mode)
    clr = patchify(torch.as_tensor(g["caller.img"]).cuda(), torch.as_tensor(g["caller.coords"]).cuda() + 0.5, 0).view(1, -1, 3)
    assert np.array_equal(clr.cpu().numpy(), g["caller.clr"])


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
