"""Track-sharded multi-GPU BA step (new functionality — the
reference is single-GPU, main/batrack.py:73-104).

One process per GPU.  Tracks are split into `world` contiguous ranges of the
sorted track list, balanced by edge count; a rank owns the edges, disparities
and priors of its tracks; poses and intrinsics are replicated.  Per step, two
calls into the C ABI around ONE exchange of the packed reduced system (140 KB at
64 keyframes instead of the dense 1.15 MB), selectable per engine:

  exchange="rccl"   bt_ba_reduce_pack -> dist.all_reduce(SUM, float64; backend nccl = RCCL over xGMI) ->
                    bt_ba_unpack_solve_update
  exchange="ipc"    bt_ba_reduce_push -> bt_ba_pull_solve_update: every rank writes its packed partial system
                    straight into a slot of every peer's hipIpc-mapped exchange buffer and raises a flag; every
                    rank sums the slots of its own buffer in rank order while unpacking.  No collective library,
                    no host round trip: two kernels on the compute stream (include/batrack_ba.h).  The only
                    collective is the one-time swap of the 64-byte buffer handles at construction.

Every rank then runs the identical solve, updates its own depths and all poses.

No second exchange inside the iteration; `gather_patches` merges the ranks'
disparities when the caller wants the full buffer back.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition_tracks(kk, world):
    """Split the sorted distinct tracks of `kk` into `world` contiguous ranges with
    near-equal edge counts.  Returns the list of (lo, hi) patch-id bounds, hi exclusive."""
    if isinstance(kk, torch.Tensor) and kk.is_cuda and kk.numel() > 0:
        # on the device (a host np.unique of 8.4M indices is 0.3 s, on every rank): the same bounds, `world` numbers come back
        ids, counts = torch.unique(kk, return_counts=True)
        csum = counts.cumsum(0).to(torch.float64)
        total, n_ids = float(csum[-1]), ids.numel()
        targets = torch.tensor([total * (r + 1) / world - 1e-9 for r in range(world - 1)], dtype=torch.float64, device=kk.device)
        cut = [min(int(c) + 1, n_ids) for c in torch.searchsorted(csum, targets, right=False).cpu().tolist()] + [n_ids]
        for r in range(1, world):
            cut[r] = max(cut[r], cut[r - 1])
        edges_idx = [0] + cut                                         # first track of rank r = edges_idx[r], one past its last = edges_idx[r + 1]
        got = ids[torch.tensor([min(i, n_ids - 1) for i in edges_idx], device=kk.device)].cpu().tolist()
        last = int(ids[-1]) + 1
        val = [int(v) if i < n_ids else last for i, v in zip(edges_idx, got)]
        return [(val[r], val[r + 1]) for r in range(world)]
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


def plan_range(own, p_tot):
    """(own_lo, own_hi) as bt_plan_create takes it: an empty range must not read as "all tracks" (own_hi = 0)."""
    lo, hi = own
    return (max(int(p_tot), 1),) * 2 if hi <= lo else (int(lo), int(hi))


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

    def __init__(self, ii, jj, kk, n_buf, p_tot, fixedp, device, world=None, rank=None, group=None, exchange="rccl", status_every=16):
        from .plan import Plan, Stepper
        if exchange not in ("rccl", "ipc"):
            raise ValueError("exchange must be 'rccl' or 'ipc'")
        self.group, self.exchange = group, exchange
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.device = torch.device(device)
        # every rank derives every rank's track range from the same edge list: nothing about the partition is exchanged
        self.ranges = partition_tracks(kk, self.world)
        self.owned = self.ranges[self.rank]
        # (a rank without tracks — more ranks than tracks — plans an empty range and still takes part in the all-reduce and the solve)
        self.plan = Plan(ii, jj, kk, n_buf, p_tot, fixedp, own=plan_range(self.owned, p_tot) if self.world > 1 else (0, 0))
        self.stepper = Stepper(self.plan, self.device)
        self._covered = None
        self._xbuf, self._peers, self._epoch = None, None, 0
        # exchange='ipc': a pull that gave up waiting for a peer makes its step a no-op for the poses (the solver sees a failed
        # factorisation) and leaves BT_XCHG_TIMEOUT in the workspace; the status word is read back every `status_every` steps
        # (one stream synchronisation: 0 = never inside step) and ALWAYS in gather_patches, i.e. once per update() of the caller
        # (batrack.py:856-895 merges the patches after its 2 x ITER calls): a time-out is reported at the end of the update() it
        # happened in, at the latest; a caller that steps without gathering calls check_exchange() itself at that point
        self.status_every = int(status_every)
        if exchange == "ipc" and self.world > 1 and self.plan.n > 0:
            self._open_exchange()

    def _open_exchange(self):
        """The rank's exchange buffer and the peers' buffers mapped into this process (once per engine).  Every rank runs
        the same collectives whatever happens locally (a rank that cannot allocate or map still takes part in the
        all-gather and in the verdict), so a failure raises on ALL ranks together instead of leaving the others waiting."""
        import ctypes
        from . import _lib
        L = self.stepper._lib
        if self.world > 16:
            raise RuntimeError("exchange='ipc' supports up to 16 ranks")
        on_dev = dist.get_backend(self.group) != "gloo"
        err = None
        buf, handle, opened = ctypes.c_void_p(), ctypes.create_string_buffer(64), []
        ptrs = (ctypes.c_void_p * self.world)()
        with torch.cuda.device(self.device):
            try:
                nbytes = L.bt_xchg_bytes(self.plan.handle, self.world)
                _lib.check(L.bt_xchg_alloc(nbytes, ctypes.byref(buf), handle), "bt_xchg_alloc")
            except Exception as e:                                        # noqa: BLE001
                err = e
            mine = torch.frombuffer(bytearray(handle.raw), dtype=torch.uint8).clone()
            # the one data collective of this path: 64 bytes per rank, once (on the host for a gloo group, on the device for nccl)
            mine = mine.to(self.device) if on_dev else mine
            allh = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(allh, mine, group=self.group)
            if err is None:
                try:
                    for r in range(self.world):
                        if r == self.rank:
                            ptrs[r] = buf.value
                        else:
                            p = ctypes.c_void_p()
                            _lib.check(L.bt_xchg_open(bytes(allh[r].cpu().numpy().tobytes()), ctypes.byref(p)), f"bt_xchg_open (rank {r})")
                            ptrs[r] = p.value
                            opened.append(p)
                except Exception as e:                                    # noqa: BLE001
                    err = e
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self.device if on_dev else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)  # (also the barrier: nobody pushes before everybody has mapped)
        if int(ok.item()) == 0:
            for p in opened:
                L.bt_xchg_close(p)
            if buf.value:
                L.bt_xchg_free(buf)
            raise RuntimeError(f"exchange='ipc' could not be set up on every rank (this rank: {err!r})")
        self._xbuf, self._peers, self._opened = buf, ptrs, opened

    def close(self):
        if self._xbuf is not None:
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)            # no peer may still be writing into a buffer that is about to go
            L = self.stepper._lib
            for p in self._opened:
                L.bt_xchg_close(p)
            L.bt_xchg_free(self._xbuf)
            self._xbuf = self._peers = None

    def step(self, poses, patches, mono, intrinsics, targets, tstride, weights, poses_out, patches_out,
             bounds, lmbda, ep, alpha, loss, structure_only, lmbda_per_track=None):
        """Same argument order as Stepper.step; per-edge tensors are the full ones on every rank.  `lmbda_per_track`: the
        reference's lmbda tensor (ba.py:299-300), one float32 per distinct track of the FULL edge list on every rank."""
        args = (poses, patches, mono, intrinsics, targets, tstride, weights, poses_out, patches_out,
                bounds, lmbda, ep, alpha, loss, structure_only)
        so = bool(structure_only) or self.plan.n == 0
        st = self.stepper
        if so or self.world == 1:
            st.step(*args, lmbda_per_track=lmbda_per_track)
            return
        # one argument block, two enqueues on the current stream around the exchange
        import ctypes
        from . import _lib
        a = st._fill(*args)
        a.lmbda_per_track = lmbda_per_track.data_ptr() if lmbda_per_track is not None else None
        L, h, ws = st._lib, self.plan.handle, st.ws.data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None
        if self._xbuf is not None:
            self._epoch += 1
            _lib.check(L.bt_ba_reduce_push(h, ctypes.byref(a), ws, self._peers, self.world, self.rank, self._epoch, stream), "bt_ba_reduce_push")
            _lib.check(L.bt_ba_pull_solve_update(h, ctypes.byref(a), ws, self._xbuf, self.world, self._epoch, stream), "bt_ba_pull_solve_update")
            if self.status_every and self._epoch % self.status_every == 0:
                self.check_exchange()
            return
        _lib.check(L.bt_ba_reduce_pack(h, ctypes.byref(a), ws, stream), "bt_ba_reduce_pack")
        allreduce_system(st.packed, self.group)          # (enqueued by torch on the same stream under nccl = RCCL)
        _lib.check(L.bt_ba_unpack_solve_update(h, ctypes.byref(a), ws, stream), "bt_ba_unpack_solve_update")

    def exchange_status(self):
        """0, or BT_XCHG_TIMEOUT (1) once a pull gave up waiting for a peer (exchange='ipc')."""
        import ctypes
        s = ctypes.c_int32()
        self.stepper._lib.bt_ba_xchg_status(self.plan.handle, self.stepper.ws.data_ptr(),
                                            torch.cuda.current_stream(self.device).cuda_stream, ctypes.byref(s))
        return int(s.value)

    def check_exchange(self):
        """Raises if a pull of this engine ever timed out (the steps since then did not move the poses)."""
        if self._xbuf is not None and self.exchange_status() != 0:
            raise RuntimeError(f"rank {self.rank}: exchange='ipc' timed out waiting for a peer's partial system (BT_XCHG_TIMEOUT); "
                               "the affected steps left the poses unchanged — a rank is hung or was delayed beyond BT_XCHG_SPIN_LIMIT polls")

    def gather_patches(self, patches_out):
        """Merge disparities: every patch slot is owned by exactly one rank (its track range); slots outside every
        range are identical on all ranks.  One fixed-shape all-reduce over the span of the ranges (a sum of disjoint
        pieces); no Python objects are exchanged."""
        if self.world == 1:
            return patches_out
        self.check_exchange()
        span_lo = min(a for a, b in self.ranges if b > a)
        span_hi = max(b for a, b in self.ranges if b > a)
        lo, hi = self.owned
        mine = torch.zeros_like(patches_out[span_lo:span_hi])
        if hi > lo:
            mine[lo - span_lo:hi - span_lo] = patches_out[lo:hi]
        allreduce_system(mine, self.group)
        out = patches_out.clone()
        out[span_lo:span_hi] = mine                     # the ranges tile [span_lo, span_hi): every slot in it has exactly one owner
        return out
