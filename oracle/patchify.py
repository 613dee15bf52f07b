"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

numpy restatement of the reference's `altcorr.patchify` (row f-2 of SURVEY.md §8): the gather of
/root/reference/main/backend/altcorr/correlation_kernel.cu:16-47 — a (2R+2)x(2R+2) window whose top-left corner is
floor(coords) - R, zero outside the image — and the bilinear blend of correlation.py:51-68 (float32, products and sum in
the reference's order).  Pinned by tests/golden/patchify.npz, which the reference's own Python produced here with the
CUDA gather replaced by a stand-in (tests/golden/make_golden_patchify.py): the blend is pinned to the reference's code,
the gather to our reading of the kernel — PARITY UNPINNED for the gather itself (the CUDA extension cannot be built)."""
import numpy as np


def gather(net, coords, R):
    B, C, H, W = net.shape
    M = coords.shape[1]
    D = 2 * R + 2
    fl = np.floor(coords).astype(np.int64)
    ii = fl[:, :, 1, None] - R + np.arange(D)[None, None]              # [B,M,D] rows
    jj = fl[:, :, 0, None] - R + np.arange(D)[None, None]              # [B,M,D] columns
    ok = ((ii >= 0) & (ii < H))[:, :, :, None] & ((jj >= 0) & (jj < W))[:, :, None, :]
    ic, jc = np.clip(ii, 0, H - 1), np.clip(jj, 0, W - 1)
    b = np.arange(B)[:, None, None, None]
    pat = net[b, :, ic[:, :, :, None], jc[:, :, None, :]]              # [B,M,D,D,C]
    pat = np.where(ok[..., None], pat, np.float32(0)).astype(np.float32)
    return np.ascontiguousarray(np.moveaxis(pat, -1, 2))               # [B,M,C,D,D]


def patchify(net, coords, R, mode="bilinear"):
    net, coords = np.asarray(net, np.float32), np.asarray(coords, np.float32)
    pat = gather(net, coords, R)
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
