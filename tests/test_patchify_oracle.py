"""Row f-2 on the CPU: oracle/patchify.py against the vectors the reference's own Python produced
(tests/golden/patchify.npz, generator tests/golden/make_golden_patchify.py).  Bit-exact."""
import os

import numpy as np

from oracle import patchify as op

GOLD = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "patchify.npz")))
N_CASES = sum(1 for k in GOLD if k.endswith(".R"))


def test_oracle_reproduces_the_reference_vectors():
    assert N_CASES == 5
    for n in range(N_CASES):
        R, mode = int(GOLD[f"case{n}.R"]), str(GOLD[f"case{n}.mode"])
        got = op.patchify(GOLD[f"case{n}.net"], GOLD[f"case{n}.coords"], R, mode)
        assert got.dtype == np.float32 and got.shape == GOLD[f"case{n}.out"].shape
        assert np.array_equal(got, GOLD[f"case{n}.out"]), (n, R, mode)


def test_caller_use_colour_lookup():
    clr = op.patchify(GOLD["caller.img"], GOLD["caller.coords"] + np.float32(0.5), 0).reshape(1, -1, 3)
    assert np.array_equal(clr, GOLD["caller.clr"])


def test_window_outside_the_image_is_zero():
    net = np.ones((1, 2, 5, 6), np.float32)
    pat = op.patchify(net, np.array([[[-10.0, 2.0], [2.0, 99.0], [5.0, 4.0]]], np.float32), 1, mode="nearest")
    assert not pat[0, 0].any() and not pat[0, 1].any()
    assert pat[0, 2, :, :2, :2].all() and not pat[0, 2, :, 2:, :].any() and not pat[0, 2, :, :, 2:].any()
