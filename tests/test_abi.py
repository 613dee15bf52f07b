"""The C-ABI library loads without a GPU and exports every symbol include/batrack_ba.h
declares; argument validation returns codes instead of crashing (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from batrack_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in sorted(f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")):     # every header of the boundary
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(bt_[A-Za-z_0-9]+)\s*\(", src))
    return sorted(names)


def test_header_symbols_are_exported():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 24 and "bt_se3_adjT" in names and "bt_reproject" in names
    for n in names:
        assert hasattr(L, n), f"{n} declared in batrack_ba.h but not exported"


def test_version_and_arch():
    L = _lib.lib()
    assert L.bt_version() >= 100
    assert L.bt_target_arch() == b"gfx950"


def test_plan_info_struct_matches_header():
    src = open(os.path.join(ROOT, "include", "batrack_ba.h")).read()
    body = re.search(r"typedef struct \{(.*?)\} bt_plan_info;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"int64_t\s+([A-Za-z_0-9]+);", body)
    assert fields == [n for n, _ in _lib.PlanInfo._fields_]
    body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} bt_ba_args;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        for part in stmt.split(","):
            m = re.search(r"([A-Za-z_0-9]+)\s*(?:\[4\])?\s*$", part.strip())
            if m and part.strip():
                names.append(m.group(1))
    assert names == [n for n, _ in _lib.BaArgs._fields_]


def test_error_codes_not_crashes():
    L = _lib.lib()
    h = ctypes.c_void_p()
    ii = np.array([0, 1], np.int64)
    # null output pointer / negative sizes / out-of-range indices
    assert L.bt_plan_create(ii.ctypes.data, ii.ctypes.data, ii.ctypes.data, 2, 4, 8, 1, 0, 0, 0, 0, 0, None) == _lib.BT_EINVAL
    assert L.bt_plan_create(ii.ctypes.data, ii.ctypes.data, ii.ctypes.data, -1, 4, 8, 1, 0, 0, 0, 0, 0, ctypes.byref(h)) == _lib.BT_EINVAL
    bad = np.array([0, 9], np.int64)
    assert L.bt_plan_create(ii.ctypes.data, bad.ctypes.data, ii.ctypes.data, 2, 4, 8, 1, 0, 0, 0, 0, 0, ctypes.byref(h)) == _lib.BT_EINVAL
    assert L.bt_plan_create(ii.ctypes.data, ii.ctypes.data, ii.ctypes.data, 2, 4, 8, 1, 0, 0, 0, 0, 0, ctypes.byref(h)) == _lib.BT_OK
    # a host-only plan refuses to launch (no device arrays): code, not a crash
    args = _lib.BaArgs()
    assert L.bt_ba_step(h, ctypes.byref(args), ctypes.c_void_p(1), None) == _lib.BT_EINVAL
    assert L.bt_ba_step(None, ctypes.byref(args), None, None) == _lib.BT_EINVAL
    assert L.bt_plan_workspace_bytes(h) > 0
    L.bt_plan_destroy(h)
    L.bt_plan_destroy(None)


def test_unsupported_graphs_are_reported():
    L = _lib.lib()
    h = ctypes.c_void_p()
    # (round 6: 299 free poses and a track seen by 79 free cameras are plans now — the dense solver, the loose tracks; what is
    #  left is more free poses than the dense solver's right-hand side has room for in LDS)
    for n, rc in ((300, _lib.BT_OK), (80, _lib.BT_OK), (2100, _lib.BT_EUNSUPPORTED)):
        ii = np.zeros(n, np.int64); jj = np.arange(n, dtype=np.int64); kk = np.zeros(n, np.int64)
        # (host-only plans: no upload without a GPU)
        assert L.bt_plan_create(ii.ctypes.data, jj.ctypes.data, kk.ctypes.data, n, n, 4, 1, 0, 0, 0, 0, 0, ctypes.byref(h)) == rc, n
        if rc == _lib.BT_OK:
            L.bt_plan_destroy(h)


def test_track_with_two_source_frames_is_rejected():
    L = _lib.lib()
    h = ctypes.c_void_p()
    ii = np.array([0, 1], np.int64); jj = np.array([1, 2], np.int64); kk = np.array([3, 3], np.int64)
    assert L.bt_plan_create(ii.ctypes.data, jj.ctypes.data, kk.ctypes.data, 2, 4, 8, 1, 0, 0, 0, 0, 0, ctypes.byref(h)) == _lib.BT_EUNSUPPORTED


def test_product_path_has_no_cpu_fallback():
    """BA_rgbd_droid must refuse CPU tensors instead of silently computing elsewhere."""
    import torch
    from batrack_amd.backend.ba import BA_rgbd_droid
    from batrack_amd.backend.lietorch import SE3
    P = torch.zeros(1, 4, 7); P[..., 6] = 1
    with pytest.raises(RuntimeError, match="GPU"):
        BA_rgbd_droid(SE3(P), torch.zeros(1, 8, 3, 1, 1), torch.zeros(1, 8, 1), torch.ones(1, 4, 4),
                      torch.zeros(1, 2, 2), None, torch.ones(1, 2, 2), 1e-4, torch.tensor([0, 1]), torch.tensor([1, 2]),
                      torch.tensor([0, 1]), [0, 0, 10, 10])


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "batrack_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, f


def test_host_thread_guard():
    """hostenv: the CPU pool is capped at the cgroup quota (a wider pool freezes the launching thread)."""
    import torch
    from batrack_amd import hostenv
    q = hostenv.cpu_quota()
    assert 1 <= q <= (os.cpu_count() or 1)
    assert hostenv.limit_host_threads() <= q
    before = torch.get_num_threads()
    assert hostenv.limit_host_threads(cap=10 ** 6) == before      # never raises the count


def test_destroyed_plans_are_recycled():
    """bt_plan_destroy keeps host arrays for the next bt_plan_create; a recycled plan is indistinguishable."""
    from batrack_amd import graphgen
    from batrack_amd.plan import Plan, PLAN_ARRAYS
    g1 = graphgen.make_config("C1", seed=0)
    g2 = graphgen.make_random_graph(12, 40, seed=3)
    ref = {}
    for name, g in (("a", g1), ("b", g2)):
        pl = Plan(g.ii, g.jj, g.kk, g.poses.shape[0], g.patches.shape[0], 1, upload=False)
        ref[name] = (dict(pl.info), pl.arrays())
        pl.close()
        _lib.lib().bt_plan_pool_trim()                     # nothing kept: the next plan is a new object
    for name, g in (("a", g1), ("b", g2), ("a", g1)):          # big -> small -> big through the same recycled object
        pl = Plan(g.ii, g.jj, g.kk, g.poses.shape[0], g.patches.shape[0], 1, upload=False)
        info, arrs = ref[name]
        assert dict(pl.info) == info
        got = pl.arrays()
        assert set(got) == set(arrs) and all(np.array_equal(got[k], arrs[k]) for k in PLAN_ARRAYS)
        pl.close()
    # an invalid edge list goes back to the pool too and does not poison the next plan
    bad = g1.jj.copy(); bad[0] = 10 ** 6
    with pytest.raises(Exception):
        Plan(g1.ii, bad, g1.kk, g1.poses.shape[0], g1.patches.shape[0], 1, upload=False)
    pl = Plan(g1.ii, g1.jj, g1.kk, g1.poses.shape[0], g1.patches.shape[0], 1, upload=False)
    assert all(np.array_equal(pl.arrays()[k], ref["a"][1][k]) for k in PLAN_ARRAYS)
    pl.close()
