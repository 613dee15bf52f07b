"""Stand-in for the reference's compiled `cuda_corr` module (CUDA; cannot be built in this image), used ONLY by
tests/golden/make_golden_patchify.py so that /root/reference/main/backend/altcorr/correlation.py imports and its own
Python (PatchLayer + the bilinear blend, correlation.py:33-68) runs unmodified.  `patchify_forward` restates the gather
of correlation_kernel.cu:16-47: a (2R+2)x(2R+2) window at floor(coords) - R, zeros outside the image.  Forward only."""
import numpy as np
import torch


def patchify_forward(net, coords, radius):
    B, C, H, W = net.shape
    M = coords.shape[1]
    D = 2 * radius + 2
    out = torch.zeros(B, M, C, D, D, dtype=net.dtype)
    fl = torch.floor(coords).to(torch.int64)
    for b in range(B):
        for m in range(M):
            x0, y0 = int(fl[b, m, 0]) - radius, int(fl[b, m, 1]) - radius
            for a in range(D):
                i = y0 + a
                if not 0 <= i < H:
                    continue
                for e in range(D):
                    j = x0 + e
                    if 0 <= j < W:
                        out[b, m, :, a, e] = net[b, :, i, j]
    return [out]


def _absent(*a, **k):
    raise NotImplementedError("cuda_corr stand-in: only patchify_forward exists")


forward = backward = patchify_backward = _absent
