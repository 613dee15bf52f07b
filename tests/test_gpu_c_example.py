"""-m gpu: the C ABI from a plain HIP host program (examples/c_abi_step.cpp: no Python, no torch in the process): compiled
here with hipcc against include/batrack_ba.h and libbatrack_ba.so, run on a generated problem, compared with the oracle."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import oracle
from batrack_amd import _lib, graphgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_plain_c_program_steps_through_the_abi(tmp_path):
    _lib.lib()                                                    # makes sure the library is built
    exe = str(tmp_path / "c_abi_step")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_step.cpp"),
                    "-L" + libdir, "-lbatrack_ba", "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True, timeout=600)
    g = graphgen.make_graph(12, 64, 6, seed=9)
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, np.float32))
    E, N, P = len(g.ii), g.poses.shape[0], g.patches.shape[0]
    prob = str(tmp_path / "problem.bin")
    with open(prob, "wb") as f:
        f.write(struct.pack("<4q", E, N, P, 1))
        f.write(f32(g.bounds).tobytes())
        for a in (g.ii, g.jj, g.kk):
            f.write(np.ascontiguousarray(a, np.int64).tobytes())
        for a in (g.poses, g.patches, g.mono_disp, g.intrinsics, g.targets3, g.weights_pose):
            f.write(f32(a).tobytes())
    res = str(tmp_path / "result.bin")
    r = subprocess.run([exe, prob, res], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Jacobian kernel 0" in r.stdout and "solver status 0" in r.stdout, r.stdout
    raw = open(res, "rb").read()
    status = struct.unpack("<i", raw[:4])[0]
    out = np.frombuffer(raw[4:], np.float32)
    poses_out, patches_out = out[:7 * N].reshape(N, 7), out[7 * N:].reshape(P, 3)
    d = lambda a: f32(a).astype(np.float64)
    ref = oracle.ba_step(d(g.poses), d(g.patches), d(g.mono_disp), d(g.intrinsics), d(g.targets3), d(g.weights_pose), g.ii, g.jj, g.kk,
                         np.asarray(g.bounds, np.float64), fixedp=1)
    rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)
    assert status == 0 and rel(poses_out, ref["poses_out"]) < 5e-6 and rel(patches_out, ref["patches_out"]) < 5e-6
