import os, sys
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from batrack_amd.parallel import ShardedBA
    from test_gpu_sharded import _inputs
    dev = torch.device("cuda:0")
    g, d = _inputs(shape="band")
    T = lambda a: torch.as_tensor(a, device=dev)
    poses, patches, mono, intr, t3, w = (T(d[k]) for k in ("poses", "patches", "mono", "intr", "t3", "w"))
    ii, jj, kk = T(d["ii"]), T(d["jj"]), T(d["kk"])
    eng = ShardedBA(ii, jj, kk, poses.shape[0], patches.shape[0], 1, dev, exchange="ipc")
    if rank == 0:
        import ctypes
        from batrack_amd import _lib
        st = eng.stepper
        Pn, Xn = torch.empty_like(poses), torch.empty_like(patches)
        a = st._fill(poses, patches, mono, intr, t3, 3, w, Pn, Xn, list(g.bounds), 1e-4, 10.0, 0.05, "huber", False)
        L, h, ws = st._lib, eng.plan.handle, st.ws.data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        print("push", L.bt_ba_reduce_push(h, ctypes.byref(a), ws, eng._peers, 2, 0, 1, stream))
        torch.cuda.synchronize()
        D = 6 * eng.plan.n
        print("S diag before pull", st.system[:D * D].reshape(D, D).diagonal()[:6].cpu().numpy())
        # pull only (no solve): bt_ba_pull_solve_update does both; look at the state after it
        print("pull", L.bt_ba_pull_solve_update(h, ctypes.byref(a), ws, eng._xbuf, 2, 1, stream))
        torch.cuda.synchronize()
        print("status", st.status(), "xstatus", eng.exchange_status(), "jk", eng.plan.jacobian_kernel)
        print("S diag after", st.system[:D * D].reshape(D, D).diagonal()[:6].cpu().numpy())
        print("dx", st.dx[:6].cpu().numpy(), "pose diff", float((Pn - poses).abs().max()))
    eng.close()
    dist.destroy_process_group()

if __name__ == "__main__":
    os.environ["BT_XCHG_SPIN_LIMIT"] = "20000"
    mp.spawn(worker, args=(2, 29555), nprocs=2, join=True)
