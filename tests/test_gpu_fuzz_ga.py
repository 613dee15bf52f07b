"""Random shapes through the dense global-alignment losses and their gradients: frames, tracks per frame
(odd, tiny, not a multiple of anything the kernels block by), window spans, query-frame subsets, the loss mix — against the float64
numpy statement of the losses and the torch autograd of it (oracle/ga_losses.py, oracle/ga_torch.py).

As a script: python tests/test_gpu_fuzz_ga.py [first_seed] [count]"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def check(seed):
    import test_gpu_global_refine as G
    from oracle import ga_losses as ga
    from oracle import ga_torch
    rng = np.random.default_rng(seed)
    T = int(rng.integers(2, 17))
    N = int(rng.choice([1, 2, 3, 5, 8, 31, 64, 65, 127, 200, 513, 777, 1200]))
    S = int(rng.choice([3, 5, 7, 9]))
    alpha = float(rng.choice([0.0, 0.3, 0.5, 1.0]))
    d = G.make_case(T, N, S, seed=seed)
    nq = int(rng.integers(1, T + 1))
    d["grid_query_frames"] = np.sort(rng.choice(T, nq, replace=bool(rng.random() < 0.2))).astype(np.int64)
    desc = f"seed {seed}: T={T} N={N} S={S} alpha={alpha} query frames {d['grid_query_frames'].tolist()}"
    net = G.build(d)
    l = net.losses().cpu().numpy()[:3]
    ms = ga.frame_scaled_depth(d)
    ref = (ga.spatial_loss(d, ms), ga.inter_frame_loss(d, ms), ga.pts_3d_loss(d, ms))
    for got, want, name in zip(l, ref, ("spatial", "rigid", "pts3d")):
        assert abs(got - want) <= 1e-5 * abs(want) + 1e-9, (desc, name, got, want)
    g = net.backward(alpha)
    tot, _, _, r_ts, r_fs = ga_torch.total_and_grads(d, alpha)
    assert abs(float(net.forward(alpha)) - tot) <= 1e-5 * abs(tot) + 1e-9, (desc, float(net.forward(alpha)), tot)
    def gerr(got, want):                          # (one track per frame has no pair: those gradients are exactly zero on both sides)
        got = np.asarray(got, np.float64)
        assert np.isfinite(got).all(), desc
        return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))
    e_ts, e_fs = gerr(g["trajs_scales"].cpu().numpy(), r_ts), gerr(g["frame_scales_"].cpu().numpy(), r_fs)
    assert e_ts < 1e-4 and e_fs < 4e-4, (desc, e_ts, e_fs)
    return desc + f" | grad errors {e_ts:.1e} {e_fs:.1e}"


@pytest.mark.parametrize("seed", range(8800, 8816))
def test_random_shapes_vs_oracle(seed):
    check(seed)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad = 0
    for s in range(first, first + count):
        try:
            print("ok  ", check(s), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", e, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            import traceback
            print("ERR ", s, type(e).__name__, e, traceback.format_exc().splitlines()[-3:], flush=True)
    print(f"{count} seeds from {first}: {bad} failed")
    sys.exit(1 if bad else 0)
