"""The random graphs of tests/test_gpu_fuzz.py through the TRACK-SHARDED step: `world` ranks share cuda:0, every rank plans its own
range of the list (the device planner's sliced analysis where it applies, the host's otherwise), reduces its tracks, exchanges the
packed system (both exchange paths, alternating by seed) and solves; the disparities are gathered.  Rank 0 compares with the oracle.

As a script: python tests/test_gpu_fuzz_sharded.py [first_seed] [count] [world] [big]"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, seeds, out, big=False):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        import test_gpu_fuzz as F
        from batrack_amd.parallel import ShardedBA
        from gpu_util import rel
        dev = torch.device("cuda:0")
        lines = []
        for seed in seeds:
            d, fixedp, so, loss, wkey, desc = F.draw(seed, big)
            T = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev)
            poses, patches, mono, intr, t3, w = (T(d[k]) for k in ("poses", "patches", "mono", "intrinsics", "targets3", wkey))
            ii, jj, kk = (torch.as_tensor(d[k], device=dev) for k in ("ii", "jj", "kk"))
            exchange = "ipc" if seed % 2 else "rccl"
            try:
                eng = ShardedBA(ii, jj, kk, poses.shape[0], patches.shape[0], fixedp, dev, exchange=exchange)
                Pn, Xn = (poses if so else torch.empty_like(poses)), torch.empty_like(patches)
                eng.step(poses, patches, mono, intr, t3, t3.stride(0), w, Pn, Xn, [float(v) for v in d["bounds"]], 1e-4, 10.0, 0.05, loss, so)
                full = eng.gather_patches(Xn)
                torch.cuda.synchronize()
                status, xstatus = eng.stepper.status(), eng.exchange_status()
                f32 = eng.plan.edge_precision != 8
                kern = eng.plan.jacobian_kernel
                eng.close()
            except Exception as e:  # noqa: BLE001
                lines.append(f"ERR  rank {rank} {desc}: {type(e).__name__} {e}")
                dist.barrier()
                continue
            if rank == 0:
                ref = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"],
                                     d["bounds"], fixedp=fixedp, structure_only=so, loss=loss, **F.ABI_SCALARS)
                ref32 = oracle.ba_step(d["poses"], d["patches"], d["mono"], d["intrinsics"], d["targets3"], d[wkey], d["ii"], d["jj"], d["kk"],
                                       d["bounds"], fixedp=fixedp, structure_only=so, loss=loss, dtype=np.float32, **F.ABI_SCALARS)
                hp, hd = rel(ref32["poses_out"], ref["poses_out"]), rel(ref32["patches_out"], ref["patches_out"])
                ep, ed = rel(Pn.cpu().numpy(), ref["poses_out"]), rel(full.cpu().numpy(), ref["patches_out"])
                floor = 8e-6 if f32 else 3e-7
                ok = ep < max(floor, 2 * hp) and ed < max(floor, 2 * hd) and xstatus == 0
                lines.append(f"{'ok  ' if ok else 'FAIL'} {desc} | world {world} {exchange} {kern} f{'32' if f32 else '64'} status {status} "
                             f"poses={ep:.2e} patches={ed:.2e} ref32 {hp:.2e} {hd:.2e}")
            dist.barrier()
        out[rank] = lines
    finally:
        dist.destroy_process_group()


def run(seeds, world, big=False):
    port = 29900 + (os.getpid() % 1000)
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, list(seeds), out, big), nprocs=world, join=True)
    lines = [l for r in range(world) for l in out.get(r, [])]
    return lines


@pytest.mark.parametrize("world,first", [(2, 7100), (4, 7200), (8, 7300)])
def test_random_graphs_sharded_vs_oracle(world, first):
    lines = run(range(first, first + 16), world)
    bad = [l for l in lines if not l.startswith("ok")]
    assert len(lines) >= 16 and not bad, bad


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    lines = run(range(first, first + count), world, big=len(sys.argv) > 4 and sys.argv[4] == "big")
    bad = [l for l in lines if not l.startswith("ok")]
    for l in lines:
        print(l)
    print(f"{count} seeds from {first}, world {world}: {len(bad)} failed")
    sys.exit(1 if bad else 0)
