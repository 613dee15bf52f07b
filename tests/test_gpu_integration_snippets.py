"""-m gpu: the two binding snippets of INTEGRATION.md §2 (torch.ops and ctypes), executed as written in the document, give
the step the package's own Stepper gives."""
import os
import re

import numpy as np
import pytest
import torch

from batrack_amd import graphgen
from batrack_amd.plan import Plan, Stepper

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def snippets():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. The binding itself"):text.index("## 3. What maps to what")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 2 and "torch.ops.batrack_hip.ba_step" in blocks[0] and "L.bt_ba_step" in blocks[1]
    return blocks


@pytest.mark.parametrize("which", [0, 1])
def test_documented_binding_runs_and_matches_the_package(which):
    g = graphgen.make_graph(12, 64, 6, seed=4)
    dev = "cuda:0"
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
    env = dict(poses=f32(g.poses), patches=f32(g.patches), mono=f32(g.mono_disp), intrinsics=f32(g.intrinsics), targets_3d=f32(g.targets3),
               weights=f32(g.weights_pose), ii=torch.as_tensor(g.ii, device=dev), jj=torch.as_tensor(g.jj, device=dev), kk=torch.as_tensor(g.kk, device=dev),
               N_buf=g.poses.shape[0], P_tot=g.patches.shape[0], fixedp=1, wd=float(g.bounds[2]), ht=float(g.bounds[3]))
    env["poses_out"], env["patches_out"] = torch.empty_like(env["poses"]), torch.empty_like(env["patches"])
    cwd = os.getcwd()
    os.chdir(ROOT)                                   # the snippets name the libraries relative to the repository root
    try:
        exec(compile(snippets()[which], f"INTEGRATION.md#2[{which}]", "exec"), env)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    if which == 1:
        assert env["rc"] == 0
        env["L"].bt_plan_destroy(env["plan"])
    st = Stepper(Plan(env["ii"], env["jj"], env["kk"], env["N_buf"], env["P_tot"], 1), dev)
    P, X = torch.empty_like(env["poses"]), torch.empty_like(env["patches"])
    st.step(env["poses"], env["patches"], env["mono"], env["intrinsics"], env["targets_3d"], 3, env["weights"], P, X,
            [0.0, 0.0, env["wd"], env["ht"]], 1e-4, 10.0, 0.05, "huber", False)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(env["poses_out"], P) < 1e-6 and rel(env["patches_out"], X) < 1e-6      # (two executions: f64 atomics in a different order)
    assert float((env["poses_out"] - env["poses"]).abs().max()) > 0                   # and it did move the poses
