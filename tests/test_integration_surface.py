"""The reference-named import surface (integration/): batrack.py's own import lines resolve to batrack_amd and the call
signatures equal the ones recorded from the reference (tests/golden/signatures.json, made by make_signatures.py)."""
import inspect
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIG = json.load(open(os.path.join(ROOT, "tests", "golden", "signatures.json")))


def _run(code):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "integration"), ROOT]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_batrack_import_lines_resolve_and_signatures_match():
    lines = "\n".join(SIG["batrack_imports"])                 # exactly /root/reference/main/batrack.py:8-11
    out = _run(lines + """
import inspect, json
import backend.projective_ops as bare_pops            # the second root, ba.py:3
assert bare_pops is pops
print(json.dumps({
    "BA_rgbd_droid": str(inspect.signature(BA_rgbd_droid)),
    "pops": {n: str(inspect.signature(getattr(pops, n))) for n in %r},
    "stack": str(inspect.signature(lietorch.stack)), "cat": str(inspect.signature(lietorch.cat)),
    "names": [n for n in %r if hasattr(lietorch, n)],
    "se3": [n for n in %r if hasattr(SE3, n) or n in ("data", "shape", "device")],
    "patchify": str(inspect.signature(altcorr.patchify)),
    "where": BA_rgbd_droid.__module__}))
""" % (sorted(SIG["projective_ops"]), SIG["lietorch"]["module_names"], SIG["lietorch"]["SE3_used"]))
    got = json.loads(out.strip().splitlines()[-1])
    assert got["where"] == "batrack_amd.backend.ba"
    assert got["BA_rgbd_droid"] == SIG["ba"]["BA_rgbd_droid"]
    for n, s in SIG["projective_ops"].items():
        ref = inspect.signature(eval("lambda " + s[1:-1] + ": 0"))
        mine = inspect.signature(eval("lambda " + got["pops"][n][1:-1] + ": 0"))
        # same leading parameters, names and defaults (ours may append keyword-only switches such as fused=True)
        rp, mp = list(ref.parameters.values()), list(mine.parameters.values())
        assert [(p.name, p.default) for p in mp[:len(rp)]] == [(p.name, p.default) for p in rp], n
        assert all(p.default is not inspect.Parameter.empty for p in mp[len(rp):]), n
    assert got["stack"] == SIG["lietorch"]["stack"] and got["cat"] == SIG["lietorch"]["cat"]
    assert got["names"] == SIG["lietorch"]["module_names"]
    assert got["se3"] == SIG["lietorch"]["SE3_used"]
    assert got["patchify"] == "(" + ", ".join(SIG["altcorr"]["patchify_call"]) + ")"


def test_absent_groups_raise_on_use_only():
    _run("""
from main.backend.lietorch import SO3, Sim3
from main.backend import altcorr
for f in (SO3, Sim3, altcorr.corr):
    try:
        f()
    except NotImplementedError:
        continue
    raise SystemExit("expected NotImplementedError")
""")


@pytest.mark.gpu
def test_windowed_ba_through_the_reference_names():
    """The replayed caller loop (batrack_amd/sequence.py, BATRACK.update's call pattern) driven through
    `main.backend.ba.BA_rgbd_droid` / `main.backend.lietorch.SE3`: same trajectory as through batrack_amd's own names."""
    out = _run("""
import sys, json
sys.path.insert(0, %r)
import numpy as np, torch
from main.backend.ba import BA_rgbd_droid
from main.backend.lietorch import SE3
import batrack_amd.backend.ba as own
from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA
res = []
for fn in (BA_rgbd_droid, own.BA_rgbd_droid):
    obs = SyntheticObservations(n_frames=24, M=32, seed=3)
    w = WindowedBA(obs, fn, SlamConfig(PATCHES_PER_FRAME=32, BUFFER_SIZE=25), device="cuda:0")
    res.append(w.run())
assert isinstance(SE3.Identity(1, device="cuda:0") * SE3.Identity(1, device="cuda:0"), SE3)
print(json.dumps({"diff": float(np.abs(res[0] - res[1]).max())}))
""" % os.path.join(ROOT, "tests"))
    # (the same function behind both names: the runs differ only by the order of the float64 atomics of two executions)
    assert json.loads(out.strip().splitlines()[-1])["diff"] < 1e-5
