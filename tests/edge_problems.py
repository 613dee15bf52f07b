"""Degenerate BA problems shared by the CPU (plan emulator) and GPU edge-case tests."""
import numpy as np

import oracle


def problem(ii, jj, kk, n_buf, p_tot, seed=0, target_shift=0.0):
    rng = np.random.default_rng(seed)
    K = np.tile(np.array([500.0, 500.0, 320.0, 240.0]), (n_buf, 1))
    poses = np.zeros((n_buf, 7)); poses[:, 6] = 1.0
    poses[:, 0] = 0.05 * np.arange(n_buf)
    q = 0.01 * rng.standard_normal((n_buf, 4)) + np.array([0, 0, 0, 1.0])
    poses[:, 3:] = q / np.linalg.norm(q, axis=1, keepdims=True)
    patches = np.stack([rng.uniform(100, 540, p_tot), rng.uniform(80, 400, p_tot), rng.uniform(0.2, 1.0, p_tot)], 1)
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    d = dict(poses=f(poses), patches=f(patches), mono=f(patches[:, 2] * 1.05), intrinsics=f(K), ii=np.asarray(ii, np.int64), jj=np.asarray(jj, np.int64),
             kk=np.asarray(kk, np.int64), bounds=np.array([0.0, 0.0, 640.0, 480.0]))
    E = len(d["ii"])
    # targets: the reprojection of the current state plus noise (so that the residuals are small and valid) [+ a shift]
    e = oracle.edges(d["poses"], d["patches"], d["intrinsics"], np.zeros((E, 3)), np.ones((E, 2)), d["ii"], d["jj"], d["kk"], d["bounds"])
    t3 = np.zeros((E, 3)); t3[:, :2] = e["coords"] + rng.normal(0, 0.5, (E, 2)) + target_shift
    d["targets3"] = f(t3)
    d["weights"] = d["weights_pose"] = f(rng.uniform(0.5, 1.0, (E, 2)))
    return d


CASES = {
    "one_edge": (lambda: problem([0], [1], [3], n_buf=2, p_tot=8), 1),
    "one_track": (lambda: problem([0] * 5, [1, 2, 3, 4, 5], [7] * 5, n_buf=6, p_tot=16), 1),
    "two_frames": (lambda: problem([0] * 20 + [1] * 20, [1] * 20 + [0] * 20, list(range(20)) + list(range(32, 52)), n_buf=2, p_tot=64), 1),
    "all_fixed": (lambda: problem([0] * 10 + [1] * 10, [1] * 10 + [2] * 10, list(range(10)) + list(range(16, 26)), n_buf=3, p_tot=32), 3),
    "all_masked": (lambda: problem([0] * 12, [1] * 6 + [2] * 6, list(range(6)) * 2, n_buf=3, p_tot=8, target_shift=300.0), 1),
    "self_edges": (lambda: problem([1] * 8, [1] * 8, list(range(8)), n_buf=3, p_tot=8), 1),
}
