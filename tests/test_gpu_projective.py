"""bt_reproject (include/batrack_projective.h, row f-3): the fused reprojection against the reference's golden
coordinates (float64 run of its transform), against the oracle's per-edge coordinates, and against the composed
tensor operations it replaces — patch sizes 1 and 3, depth / valid / translation-only variants, bad indices."""
import ctypes
import os

import numpy as np
import pytest
import torch

import oracle
from batrack_amd import _lib, graphgen
from batrack_amd.backend import projective_ops as pops
from batrack_amd.backend.lietorch import SE3

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def gpu_inputs(poses, patches, intr, ii, jj, kk, p=1, seed=0):
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=DEV)
    pat = f32(patches)[None, :, :, None, None]
    if p > 1:                                          # a p x p window around the centre, as patchify makes it
        rng = np.random.default_rng(seed)
        off = torch.as_tensor(rng.normal(0, 1.5, (1, pat.shape[1], 3, p, p)).astype(np.float32), device=DEV)
        off[:, :, 2] = 0.0
        off[:, :, :, p // 2, p // 2] = 0.0
        pat = pat + off
    return (SE3(f32(poses)[None]), pat.contiguous(), f32(intr)[None],
            *(torch.as_tensor(np.asarray(a, np.int64), device=DEV) for a in (ii, jj, kk)))


@pytest.mark.parametrize("name", ["c1", "c1_rough", "window_small"])
def test_fused_matches_reference_coordinates(name):
    d = dict(np.load(os.path.join(GOLD, name + ".npz")))
    G, pat, K, ii, jj, kk = gpu_inputs(d["poses"], d["patches"], d["intrinsics"], d["ii"], d["jj"], d["kk"])
    x, v = pops.transform(G, pat, K, ii, jj, kk, valid=True)
    assert x.shape == (1, ii.numel(), 1, 1, 2) and v.shape == (1, ii.numel(), 1, 1)
    o = oracle.edges(d["poses"], d["patches"], d["intrinsics"], d["targets3"], d["weights"], d["ii"], d["jj"], d["kk"], d["bounds"])
    got = x[0, :, 0, 0].cpu().numpy().astype(np.float64)
    # float32 against float64: relative to the pixel magnitude; points near the camera plane (rough graphs) are
    # amplified by 1/Z and compared where the reference itself is tame
    def close(ref):
        tame = np.isfinite(ref).all(1) & (np.abs(ref).max(1) < 1e4)
        assert tame.mean() > 0.9
        return (np.abs(got - ref) / (100.0 + np.abs(ref)))[tame].max()         # 2e-3 px at the image centre scale
    if "tf64.coords" in d:
        assert close(d["tf64.coords"]) < 2e-5
        flips = v[0, :, 0, 0].cpu().numpy() != d["tf64.valid"].astype(np.float32)
        assert flips.mean() < 1e-3
    assert close(o["coords"]) < 2e-5


@pytest.mark.parametrize("p", [1, 3])
@pytest.mark.parametrize("depth,tonly", [(False, False), (True, False), (False, True), (True, True)])
def test_fused_equals_composed_operations(p, depth, tonly):
    g = graphgen.make_random_graph(20, 64, seed=5)
    G, pat, K, ii, jj, kk = gpu_inputs(g.poses, g.patches, g.intrinsics, g.ii, g.jj, g.kk, p=p)
    a, va = pops.transform(G, pat, K, ii, jj, kk, depth=depth, valid=True, tonly=tonly)
    b, vb = pops.transform(G, pat, K, ii, jj, kk, depth=depth, valid=True, tonly=tonly, fused=False)
    assert a.shape == b.shape == (1, ii.numel(), p, p, 3 if depth else 2) and va.shape == vb.shape
    # points close to the camera plane amplify rounding by 1/Z: compare where the composed result is tame
    tame = (b[..., :2].abs().amax(-1) < 1e4)
    assert tame.float().mean() > 0.9
    err = ((a - b).abs() / (100.0 + b.abs()))[tame]
    assert float(err.max()) < 2e-5, float(err.max())
    assert float((va != vb).float().mean()) < 1e-3            # Z within rounding of 0.2 may flip
    # non-contiguous index views and a strided patch tensor are accepted
    a2 = pops.transform(G, pat.expand(1, -1, -1, -1, -1), K, ii[::2], jj[::2], kk[::2], depth=depth, tonly=tonly)
    assert torch.equal(a2, a[:, ::2])


def test_flow_mag_and_self_reprojection():
    g = graphgen.make_config("C1", seed=0)
    G, pat, K, ii, jj, kk = gpu_inputs(g.poses, g.patches, g.intrinsics, g.ii, g.jj, g.kk)
    same = pops.transform(G, pat, K, ii, ii, kk)
    assert float((same[0, :, 0, 0] - pat[0, kk, :2, 0, 0]).abs().max()) < 1e-3
    f = pops.flow_mag(G, pat, K, ii, jj, kk, beta=0.5)
    c0 = pops.transform(G, pat, K, ii, ii, kk, fused=False)
    c1 = pops.transform(G, pat, K, ii, jj, kk, fused=False)
    c2 = pops.transform(G, pat, K, ii, jj, kk, tonly=True, fused=False)
    ref = 0.5 * (c1 - c0).norm(dim=-1) + 0.5 * (c2 - c0).norm(dim=-1)
    assert float((f - ref).abs().max()) < 1e-2 and f.shape == ref.shape


def test_abi_rejects_and_flags():
    L = _lib.lib()
    g = graphgen.make_config("C1", seed=0)
    G, pat, K, ii, jj, kk = gpu_inputs(g.poses, g.patches, g.intrinsics, g.ii, g.jj, g.kk)
    E = ii.numel()
    out = torch.zeros(E, 2, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    args = lambda **k: [k.get("poses", G.data.data_ptr()), G.data.shape[1], pat.data_ptr(), pat.shape[1], k.get("pe", 1), K.data_ptr(),
                        ii.data_ptr(), jj.data_ptr(), kk.data_ptr(), k.get("E", E), k.get("mode", 0), out.data_ptr(), None, st]
    assert L.bt_reproject(*args(E=-1)) == _lib.BT_EINVAL
    assert L.bt_reproject(*args(pe=0)) == _lib.BT_EINVAL
    assert L.bt_reproject(*args(mode=7)) == _lib.BT_EINVAL
    assert L.bt_reproject(*args(poses=None)) == _lib.BT_EINVAL
    assert L.bt_reproject(*args(E=0)) == _lib.BT_OK
    bad = jj.clone(); bad[3] = 10 ** 6; bad[5] = -1
    x, v = pops.transform(G, pat, K, ii, bad, kk, valid=True)
    torch.cuda.synchronize()
    assert bool(torch.isnan(x[0, 3]).all()) and bool(torch.isnan(x[0, 5]).all()) and float(v[0, 3].sum() + v[0, 5].sum()) == 0.0
    good = torch.ones(E, dtype=torch.bool, device=DEV); good[3] = good[5] = False
    assert not bool(torch.isnan(x[0, good]).any())
