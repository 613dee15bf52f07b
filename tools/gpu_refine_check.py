import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import numpy as np, torch, oracle
from batrack_amd import graphgen
from gpu_util import HipProblem, rel, update_err
f = lambda a: np.asarray(a, np.float32).astype(np.float64)
def run(g, fixedp, name):
    d = dict(poses=f(g.poses), patches=f(g.patches), mono=f(g.mono_disp), intrinsics=f(g.intrinsics),
             targets3=f(g.targets3), weights=f(g.weights), weights_pose=f(g.weights_pose), ii=g.ii, jj=g.jj, kk=g.kk, bounds=np.asarray(g.bounds))
    ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d["weights_pose"], d["ii"], d["jj"], d["kk"], d["bounds"], fixedp=fixedp, want_system=True)
    o = HipProblem(d).raw_step("weights_pose", fixedp)
    n = o["plan"].n
    print(f"{name}: n={n} nnzb={o['plan'].nnz_blocks} status={o['status']} dX {rel(o['dX'].reshape(-1), ref['dX'].reshape(-1)):.2e} poses {rel(o['poses_out'], ref['poses_out']):.2e} "
          f"patches {rel(o['patches_out'], ref['patches_out']):.2e} upd pose {update_err(o['poses_out'], ref['poses_out'], d['poses'], np.arange(fixedp, fixedp+n)):.2e}", flush=True)
run(graphgen.make_graph(256, 8, 4, seed=11), 1, "band255")
run(graphgen.make_graph(99, 16, 8, seed=3), 0, "band99")
run(graphgen.make_random_graph(47, 30, seed=5, far_frac=1.0), 1, "dense47")
