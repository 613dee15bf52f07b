"""Track-sharded multi-GPU BA step (new functionality — the
reference is single-GPU, main/batrack.py:73-104).

One process per GPU.  Tracks are split into `world` contiguous ranges of the
sorted track list, balanced by edge count; a rank owns the edges, disparities
and priors of its tracks; poses and intrinsics are replicated.  Per step:

    bt_ba_reduce        partial reduced system [S | y] of the rank's tracks
    bt_ba_pack          its non-zero blocks, contiguous (140 KB instead of 1.15 MB at 64 keyframes)
    all_reduce(SUM)     the ONE exchange, float64 (RCCL over xGMI)
    bt_ba_unpack        back into [S | y]
    bt_ba_solve_update  identical solve on every rank, own depths, all poses

No second exchange inside the iteration; `gather_patches` merges the ranks'
disparities when the caller wants the full buffer back.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition_tracks(kk, world):
    """Split the sorted distinct tracks of `kk` into `world` contiguous ranges with
    near-equal edge counts.  Returns the list of (lo, hi) patch-id bounds, hi exclusive."""
    kk = np.asarray(kk.cpu() if isinstance(kk, torch.Tensor) else kk)
    if kk.size == 0:
        return [(0, 0)] * world
    ids, counts = np.unique(kk, return_counts=True)
    csum = np.cumsum(counts)
    total = csum[-1]
    bounds, lo_idx = [], 0
    for r in range(world):
        target = total * (r + 1) / world
        hi_idx = int(np.searchsorted(csum, target - 1e-9, side="left")) + 1 if r < world - 1 else len(ids)
        hi_idx = max(hi_idx, lo_idx)
        hi_idx = min(hi_idx, len(ids))
        lo = int(ids[lo_idx]) if lo_idx < len(ids) else int(ids[-1]) + 1
        hi = int(ids[hi_idx]) if hi_idx < len(ids) else int(ids[-1]) + 1
        bounds.append((lo, hi))
        lo_idx = hi_idx
    return bounds


def shard_edges(kk, world, rank):
    """Indices (ascending, int64 tensor on kk's device) of the edges rank `rank` owns."""
    lo, hi = partition_tracks(kk, world)[rank]
    kt = kk if isinstance(kk, torch.Tensor) else torch.as_tensor(kk)
    return torch.nonzero((kt >= lo) & (kt < hi), as_tuple=False).reshape(-1)


def allreduce_system(system, group=None):
    """Sum the partial reduced systems in place.  `system` is Stepper.system (float64
    [S | y]); backend nccl (= RCCL) on GPUs, gloo in CPU tests.  With a gloo group and a
    device tensor (two test ranks sharing one GPU) the sum is staged through the host."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if system.is_cuda and dist.get_backend(group) == "gloo":
            host = system.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            system.copy_(host)
        else:
            dist.all_reduce(system, op=dist.ReduceOp.SUM, group=group)
    return system


class ShardedBA:
    """BA_rgbd_droid over a track shard.  Construct on every rank with the FULL edge list;
    per-edge inputs (targets, weights) stay the full tensors on every rank — a rank's plan
    only touches the edges of its own track range."""

    def __init__(self, ii, jj, kk, n_buf, p_tot, fixedp, device, world=None, rank=None, group=None):
        from .plan import Plan, Stepper
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.device = torch.device(device)
        self.owned = partition_tracks(kk, self.world)[self.rank]
        self.plan = Plan(ii, jj, kk, n_buf, p_tot, fixedp, own=self.owned if self.world > 1 else (0, 0))
        self.stepper = Stepper(self.plan, self.device)

    def local(self, per_edge):
        """Kept for callers written against the first version: per-edge tensors are used whole."""
        return per_edge

    def step(self, poses, patches, mono, intrinsics, targets, tstride, weights, poses_out, patches_out,
             bounds, lmbda, ep, alpha, loss, structure_only):
        """targets / weights are already local (see `local`).  Same argument order as Stepper.step."""
        args = (poses, patches, mono, intrinsics, targets, tstride, weights, poses_out, patches_out,
                bounds, lmbda, ep, alpha, loss, structure_only)
        so = bool(structure_only) or self.plan.n == 0
        st = self.stepper
        if so or self.world == 1:
            st.step(*args)
            return
        # one argument block, five enqueues on the current stream (the all-reduce is enqueued by torch on the same stream)
        import ctypes
        from . import _lib
        a = st._fill(*args)
        L, h, ws = st._lib, self.plan.handle, st.ws.data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None
        _lib.check(L.bt_ba_reduce(h, ctypes.byref(a), ws, stream), "bt_ba_reduce")
        _lib.check(L.bt_ba_pack(h, ctypes.byref(a), ws, stream), "bt_ba_pack")
        allreduce_system(st.packed, self.group)
        _lib.check(L.bt_ba_unpack(h, ctypes.byref(a), ws, stream), "bt_ba_unpack")
        _lib.check(L.bt_ba_solve_update(h, ctypes.byref(a), ws, stream), "bt_ba_solve_update")

    def gather_patches(self, patches_out):
        """Merge disparities: every patch slot is owned by exactly one rank (its track
        range); slots outside every range are identical on all ranks."""
        if self.world == 1:
            return patches_out
        ranges = [None] * self.world
        dist.all_gather_object(ranges, self.owned, group=self.group)
        lo, hi = self.owned
        mine = torch.zeros_like(patches_out)
        mine[lo:hi] = patches_out[lo:hi]
        allreduce_system(mine, self.group)
        covered = torch.zeros(patches_out.shape[0], dtype=torch.bool, device=patches_out.device)
        for a, b in ranges:
            covered[a:b] = True
        return torch.where(covered[:, None], mine, patches_out)
