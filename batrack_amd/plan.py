"""Plan and step objects over the C ABI (include/batrack_ba.h).

`Plan` owns one bt_plan (the structure of an edge list: what the reference
recomputes per call at ba.py:219,276).  `Stepper` binds a plan to device buffers
and runs BA_rgbd_droid-equivalent steps with no allocation on the call path.
"""
import ctypes

import numpy as np

import os

from . import _lib

_USE_CTYPES = os.environ.get("BT_PY_BINDING", "") == "ctypes"      # measurement only: the ctypes route for every step

_NP_TYPES = {"slot_lab": np.uint16, "slot_lp": np.uint8, "act_bits": np.uint32, "slot_code": np.uint16, "tile_la": np.uint8, "tile_sinfo": np.uint32, "tile_cut8": np.uint16, "tile_cut16": np.uint16, "pm_lb": np.uint8, "pm_la": np.uint8}
PLAN_ARRAYS = ("kx", "trk_of_patch", "trk_loc", "pair_i", "pair_j", "tile_trk0", "tile_ntrk", "tile_ncam",
               "tile_cam0", "tile_slot0", "tile_nslot", "tile_erow0", "tile_cams", "slot_edge", "slot_pair",
               "slot_lab", "col_ptr", "row_idx", "upd_ptr", "upd", "blk_col", "upd_next", "perm", "blk_src",
               "lvl_ptr", "lvl_cols", "col_lvl", "dp_ptr", "dp", "tile_pair0", "tile_npair", "tile_pairs", "slot_lp", "tile_flags",
               "fz_pend_ptr", "fz_pend", "fz_lazy_ptr", "fz_lazy", "fz_yurg", "fz_meta", "fz_pmeta", "bs_sync", "fz_rowinfo", "fz_pfirst", "fz_psecond", "act_bits", "act_rank", "tile_ij", "tile_kx", "lvl_meta", "slot_code", "tile_la", "tile_rec", "it_edge", "tile_sinfo", "tile_cut8", "tile_cut16", "pm_edge", "pm_rec", "pm_lb", "pm_la", "pp_ptr", "pp_idx", "sg_ptr")


def wave_per_tile_kernels(enable=None):
    """The wave-per-tile kernels (k_stream, k_edge2 / k_edge2u) for graphs of >= 2048 tiles, on by default: mixed precision (float64
    reprojection and residual, float32 Jacobians), update within 1e-5 of the reference's float64 run, about three times the
    float64 tile kernels' throughput.  `wave_per_tile_kernels(False)` lays the plans created afterwards out for the float64
    tile kernels whatever their size (include/batrack_ba.h: bt_config_wave_per_tile_kernels).  Returns the previous setting; no
    argument only queries."""
    return bool(_lib.lib().bt_config_wave_per_tile_kernels(-1 if enable is None else int(bool(enable))))


class Plan:
    """bt_plan handle.  ii/jj/kk: int64 torch tensors (CPU or GPU) or numpy arrays."""

    def __init__(self, ii, jj, kk, n_buf, p_tot, fixedp, upload=True, n_all_min=0, own=(0, 0), sync=True):
        L = _lib.lib()
        self._lib = L
        self._h = ctypes.c_void_p()
        self._keep = None
        on_device = 0
        if isinstance(ii, np.ndarray):
            arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (ii, jj, kk)]
            ptrs = [a.ctypes.data for a in arrs]
            E = arrs[0].shape[0]
        else:
            import torch
            arrs = [a.contiguous() for a in (ii, jj, kk)]
            for a in arrs:
                if a.dtype != torch.int64:
                    raise TypeError("edge indices must be int64 (batrack.py:100-102)")
            on_device = 1 if arrs[0].is_cuda else 0
            if on_device and sync:
                torch.cuda.current_stream(arrs[0].device).synchronize()   # the indices must be complete (batrack_ba.h)
            ptrs = [a.data_ptr() for a in arrs]
            E = arrs[0].numel()
        self._keep = arrs
        rc = L.bt_plan_create(ptrs[0], ptrs[1], ptrs[2], E, int(n_buf), int(p_tot), int(fixedp),
                              int(n_all_min), int(own[0]), int(own[1]), on_device, 1 if upload else 0, ctypes.byref(self._h))
        _lib.check(rc, "bt_plan_create")
        self._keep = None
        info = _lib.PlanInfo()
        _lib.check(L.bt_plan_get_info(self._h, ctypes.byref(info)), "bt_plan_get_info")
        self.info = {n: int(getattr(info, n)) for n, _ in _lib.PlanInfo._fields_}
        self.uploaded = bool(upload)

    @classmethod
    def shifted(cls, src, ii, jj, kk, n_buf, p_tot, fixedp, sync=True):
        """The plan of an edge list that is `src`'s with every frame index moved by one constant and every patch index by
        another (the caller's sliding window in steady state): a device-side copy of `src`'s tables with those numbers
        shifted, no host analysis (bt_plan_create_shifted).  Returns None when the list is no such copy."""
        import torch
        if not (src.uploaded and all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64 for t in (ii, jj, kk))):
            return None
        arrs = [a.contiguous() for a in (ii, jj, kk)]
        if arrs[0].numel() != src.E:
            return None
        if sync:
            torch.cuda.current_stream(arrs[0].device).synchronize()
        h = ctypes.c_void_p()
        rc = src._lib.bt_plan_create_shifted(src.handle, arrs[0].data_ptr(), arrs[1].data_ptr(), arrs[2].data_ptr(), arrs[0].numel(),
                                             int(n_buf), int(p_tot), int(fixedp), ctypes.byref(h))
        if rc > 0:
            return None
        _lib.check(rc, "bt_plan_create_shifted")
        self = cls.__new__(cls)
        self._lib, self._h, self._keep = src._lib, h, None
        info = _lib.PlanInfo()
        _lib.check(self._lib.bt_plan_get_info(self._h, ctypes.byref(info)), "bt_plan_get_info")
        self.info = {n: int(getattr(info, n)) for n, _ in _lib.PlanInfo._fields_}
        self.uploaded = True
        return self

    @classmethod
    def shifted_any(cls, srcs, ii, jj, kk, n_buf, p_tot, fixedp, sync=True):
        """`shifted` against several candidate plans at once (bt_plan_create_shifted_any: one comparison pass and one host
        wait for all of them).  Returns (plan, the source plan that matched) or (None, None) — the plan itself, not a position:
        candidates that were never uploaded are skipped here, so a position would not be one in the caller's list."""
        import torch
        srcs = [s for s in srcs if s.uploaded][:4]
        if not srcs or not (ii.is_cuda and ii.dtype == jj.dtype == kk.dtype == torch.int64 and jj.is_cuda and kk.is_cuda):
            return None, None
        arrs = [a if a.is_contiguous() else a.contiguous() for a in (ii, jj, kk)]
        if sync:
            torch.cuda.current_stream(arrs[0].device).synchronize()
        lib = srcs[0]._lib
        handles = (ctypes.c_void_p * len(srcs))(*[s._h for s in srcs])
        h, which = ctypes.c_void_p(), ctypes.c_int32(-1)
        rc = lib.bt_plan_create_shifted_any(handles, len(srcs), arrs[0].data_ptr(), arrs[1].data_ptr(), arrs[2].data_ptr(), arrs[0].numel(),
                                            int(n_buf), int(p_tot), int(fixedp), ctypes.byref(which), ctypes.byref(h))
        if rc > 0:
            return None, None
        _lib.check(rc, "bt_plan_create_shifted_any")
        self = cls.__new__(cls)
        self._lib, self._h, self._keep = lib, h, None
        # (the clone's figures are its source's but for the two the shift moves: this is on the critical path of every frame)
        matched = srcs[which.value]
        src = matched.info
        self.info = dict(src, fixedp=int(fixedp), n_all=src["n_all"] + int(fixedp) - src["fixedp"])
        self.uploaded = True
        return self, matched

    @classmethod
    def shifted_spec(cls, src, ii, jj, kk, n_buf, p_tot, fixedp):
        """The plan of a list ASSUMED to be `src`'s shifted by fixedp - src.fixedp frames (bt_plan_create_shifted_spec): no
        synchronisation, no host wait — the comparison that proves the assumption runs on the GPU beside the clone's copies.
        The plan can be stepped at once; `confirm()` must be called before its results are used (it returns False where the
        assumption was wrong: discard the plan and what it computed).  None where the shift is not of that form."""
        import torch
        if not (src.uploaded and ii.is_cuda and jj.is_cuda and kk.is_cuda and ii.dtype == jj.dtype == kk.dtype == torch.int64
                and ii.is_contiguous() and jj.is_contiguous() and kk.is_contiguous()):
            return None
        h = ctypes.c_void_p()
        # (the raw handle of the current stream: building a torch.cuda.Stream object for it costs more than the C call's share)
        stream = torch._C._cuda_getCurrentRawStream(ii.device.index if ii.device.index is not None else torch.cuda.current_device())
        rc = src._lib.bt_plan_create_shifted_spec(src._h, ii.data_ptr(), jj.data_ptr(), kk.data_ptr(), ii.numel(), int(n_buf), int(p_tot),
                                                  int(fixedp), stream, ctypes.byref(h))
        if rc > 0:
            return None
        _lib.check(rc, "bt_plan_create_shifted_spec")
        self = cls.__new__(cls)
        self._lib, self._h, self._keep = src._lib, h, (ii, jj, kk)       # (the index tensors are read by kernels still queued)
        s = src.info
        self.info = dict(s, fixedp=int(fixedp), n_all=s["n_all"] + int(fixedp) - s["fixedp"])
        self.uploaded = True
        self.speculative = True
        return self

    @classmethod
    def preshift(cls, src, df):
        """The clone of `src` for a list that does not exist yet: src's moved up by `df` frames (bt_plan_preshift) — enqueued now, on
        the library's plan stream, so that the call that brings the list only has to `bind` it.  None where no such shift fits the
        buffers.  The plan cannot be stepped before `bind`."""
        if not src.uploaded or src.__dict__.get("speculative"):
            return None
        h = ctypes.c_void_p()
        rc = src._lib.bt_plan_preshift(src._h, int(df), ctypes.byref(h))
        if rc > 0:
            return None
        _lib.check(rc, "bt_plan_preshift")
        self = cls.__new__(cls)
        self._lib, self._h, self._keep = src._lib, h, None
        s = src.info
        self.info = dict(s, fixedp=s["fixedp"] + int(df), n_all=s["n_all"] + int(df))
        self.uploaded = True
        self.speculative = True
        self.unbound = True
        return self

    def bind(self, ii, jj, kk, n_buf, p_tot, fixedp):
        """Give a pre-shifted clone its list (bt_plan_spec_bind: one comparison kernel, no host wait).  True: step it, then `confirm()`
        as after `shifted_spec`.  False: the list cannot be the one the clone was made for (size, buffers or fixedp differ)."""
        import torch
        if not self.__dict__.get("unbound"):
            return False
        if not (ii.is_cuda and jj.is_cuda and kk.is_cuda and ii.dtype == jj.dtype == kk.dtype == torch.int64
                and ii.is_contiguous() and jj.is_contiguous() and kk.is_contiguous()):
            return False
        stream = torch._C._cuda_getCurrentRawStream(ii.device.index if ii.device.index is not None else torch.cuda.current_device())
        rc = self._lib.bt_plan_spec_bind(self._h, ii.data_ptr(), jj.data_ptr(), kk.data_ptr(), ii.numel(), int(n_buf), int(p_tot), int(fixedp), stream)
        if rc > 0:
            return False
        _lib.check(rc, "bt_plan_spec_bind")
        self._keep = (ii, jj, kk)                                    # (the index tensors are read by a kernel still queued)
        self.unbound = False
        return True

    def confirm(self):
        """True: the plan is the list's plan (always, unless it came from `shifted_spec`).  False: the speculation failed."""
        if not self.__dict__.get("speculative"):
            return True
        if self.__dict__.get("invalid"):
            return False
        rc = self._lib.bt_plan_spec_confirm(self._h)
        self._keep = None
        if rc != 0:
            # wrong guess (rc > 0) or an invalid list (rc < 0): either way this object is no plan of the caller's list — it stays
            # marked speculative and `invalid`, so that nobody who still holds it steps on it as a confirmed plan
            self.invalid = True
            if rc > 0:
                return False
            _lib.check(rc, "bt_plan_spec_confirm")
        self.speculative = False
        return True

    def __getattr__(self, name):
        info = self.__dict__.get("info")
        if info is not None and name in info:
            return info[name]
        raise AttributeError(name)

    @property
    def handle(self):
        return self._h

    @property
    def jacobian_kernel(self):
        """'k_tile' | 'k_stream' | 'k_edge2' | 'k_etile': what the steps of this plan launch (bt_plan_jacobian_kernel; 'k_edge2': k_edge2 for the
        pose+structure reduce, k_edge2u for structure-only steps and the depth back-substitution)."""
        return {0: "k_tile", 1: "k_stream", 2: "k_edge2", 3: "k_etile"}.get(self._lib.bt_plan_jacobian_kernel(self._h), "host-only")

    @property
    def edge_precision(self):
        """8 | 6 | 4: float64 per edge, mixed (float64 reprojection and residual, float32 Jacobians: k_stream / k_edge2), float32
        (bt_plan_edge_precision)."""
        return self._lib.bt_plan_edge_precision(self._h)

    @property
    def built_on_device(self):
        """True if the planner's passes over the edges ran on the device (bt_plan_built_on_device)."""
        return bool(self._lib.bt_plan_built_on_device(self._h))

    def array(self, name):
        """Host copy of a plan array (tests / tooling)."""
        p = ctypes.c_void_p()
        n = self._lib.bt_plan_array(self._h, name.encode(), ctypes.byref(p))
        if n < 0:
            raise KeyError(name)
        dt = np.dtype(_NP_TYPES.get(name, np.int32))
        if n == 0:
            return np.zeros(0, dt)
        buf = (ctypes.c_char * (n * dt.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def arrays(self):
        return {k: self.array(k) for k in PLAN_ARRAYS}

    def close(self):
        if self._h:
            self._lib.bt_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_WS_STREAM = {}            # workspace address -> raw stream it was last handed to a Stepper under


def _raw_stream(device):
    """hipStream_t of PyTorch's current stream on `device` (without building a Stream object per call)."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
    except AttributeError:
        return torch.cuda.current_stream(device).cuda_stream


class Stepper:
    """A plan bound to a device: owns the workspace, fills bt_ba_args, launches."""

    def __init__(self, plan, device, ws=None):
        import torch
        if not plan.uploaded:
            raise RuntimeError("plan was built host-only")
        self.plan = plan
        self.device = torch.device(device)
        # zero-filled: the accumulators must start clear (bt_ba_workspace_init); every step leaves them clear.  `ws`: the
        # workspace of another stepper whose plan has the same workspace layout (a shifted clone and its source), used on the
        # same stream: every step leaves it as it found it, so the two can take turns
        # (... on the SAME stream: the stream a workspace was last handed out under is remembered, and a stepper created under
        #  another one gets a workspace of its own instead of racing the queued steps of the first — ADVICE round 5)
        cur = _raw_stream(self.device)
        if (ws is not None and ws.numel() >= max(plan.workspace_bytes, 256) and ws.device == self.device
                and _WS_STREAM.get(ws.data_ptr(), cur) == cur):
            self.ws = ws
        else:
            self.ws = torch.zeros(max(plan.workspace_bytes, 256), dtype=torch.uint8, device=self.device)
        if len(_WS_STREAM) > 64:
            _WS_STREAM.clear()
        _WS_STREAM[self.ws.data_ptr()] = cur
        self._args = _lib.BaArgs()
        self._lib = _lib.lib()
        self._ops = None if _USE_CTYPES else _lib.torch_ops()
        self._ws_ptr = self.ws.data_ptr()
        self._views = {}

    def _view(self, name):
        """Typed views into the workspace, made on first use (the per-frame path never needs them)."""
        v = self._views.get(name)
        if v is None:
            import torch
            plan, base = self.plan, self._ws_ptr
            cnt = ctypes.c_int64()
            if name == "system":          # [S | y], dense
                off = self._lib.bt_ba_system(plan.handle, base, ctypes.byref(cnt)) - base
                v = self.ws[off:off + 8 * cnt.value].view(torch.float64)
            elif name == "packed":        # its non-zero blocks, for the all-reduce
                off = self._lib.bt_ba_packed(plan.handle, base, ctypes.byref(cnt)) - base
                v = self.ws[off:off + 8 * cnt.value].view(torch.float64)
            else:                         # dX [n, 6]
                off = self._lib.bt_ba_dx(plan.handle, base) - base
                v = self.ws[off:off + 4 * 6 * plan.n].view(torch.float32).view(plan.n, 6)
            self._views[name] = v
        return v

    system = property(lambda self: self._view("system"))
    packed = property(lambda self: self._view("packed"))
    dx = property(lambda self: self._view("dx"))

    def _fill(self, poses, patches, mono, intrinsics, targets, tstride, weights, poses_out, patches_out,
              bounds, lmbda, ep, alpha, loss, structure_only):
        a = self._args
        a.poses, a.patches, a.mono_disp = poses.data_ptr(), patches.data_ptr(), mono.data_ptr()
        a.intrinsics, a.targets, a.weights = intrinsics.data_ptr(), targets.data_ptr(), weights.data_ptr()
        a.target_stride = int(tstride)
        a.mono_stride = int(mono.stride(0)) if mono.dim() == 1 and mono.numel() > 1 else 1     # a strided 1-D view is used in place
        a.poses_out, a.patches_out = poses_out.data_ptr(), patches_out.data_ptr()
        a.bounds[0], a.bounds[1], a.bounds[2], a.bounds[3] = (float(b) for b in bounds)
        a.lmbda, a.ep, a.alpha = float(lmbda), float(ep), float(alpha)
        a.loss = _lib.LOSS[loss] if isinstance(loss, str) else int(loss)
        a.structure_only = 1 if structure_only else 0
        a.lmbda_per_track = None
        return a

    _PHASES = {"all": 0, "reduce": 1, "pack": 2, "unpack": 3, "solve_update": 4}

    def step(self, poses, patches, mono, intrinsics, targets, tstride, weights, poses_out, patches_out,
             bounds, lmbda, ep, alpha, loss, structure_only, stream=None, phase="all", lmbda_per_track=None):
        """One BA_rgbd_droid call (or one of its multi-GPU phases) on PyTorch's current stream, through
        torch.ops.batrack_hip.ba_step — the operator registered over the C ABI (csrc/torch_ops.cpp).  `stream` (a raw
        hipStream_t) selects the ctypes route instead (tools that launch on a stream of their own)."""
        if stream is not None or self._ops is None:
            a = self._fill(poses, patches, mono, intrinsics, targets, tstride, weights, poses_out, patches_out,
                           bounds, lmbda, ep, alpha, loss, structure_only)
            a.lmbda_per_track = lmbda_per_track.data_ptr() if lmbda_per_track is not None else None
            st = _raw_stream(self.device) if stream is None else stream
            fn = {"all": self._lib.bt_ba_step, "reduce": self._lib.bt_ba_reduce, "pack": self._lib.bt_ba_pack,
                  "unpack": self._lib.bt_ba_unpack, "solve_update": self._lib.bt_ba_solve_update}[phase]
            _lib.check(fn(self.plan.handle, ctypes.byref(a), self._ws_ptr, st), f"bt_ba_{phase}")
            return
        mstride = int(mono.stride(0)) if mono.dim() == 1 and mono.numel() > 1 else 1
        rc = self._ops.ba_step(self.plan.handle.value, self.ws, poses, patches, mono, mstride, intrinsics, targets, int(tstride),
                               weights, poses_out, patches_out, [float(b) for b in bounds], float(lmbda), float(ep), float(alpha),
                               _lib.LOSS[loss] if isinstance(loss, str) else int(loss), bool(structure_only), self._PHASES[phase],
                               lmbda_per_track)
        _lib.check(rc, f"batrack_hip::ba_step (phase {phase})")

    def step_timed(self, *args, stream=None):
        """One step with per-kernel HIP-event timing -> dict of milliseconds."""
        import torch
        a = self._fill(*args)
        st = _raw_stream(self.device) if stream is None else stream
        ms = (ctypes.c_float * 6)()
        _lib.check(self._lib.bt_ba_step_timed(self.plan.handle, ctypes.byref(a), self.ws.data_ptr(), st, ms),
                   "bt_ba_step_timed")
        # "depth": the walk over the edges that back-substitutes the depths where it is a kernel of its own (k_edge2u / k_stream)
        return dict(zip(("prep", "tile", "pair_finalize", "solve", "update", "depth"), [float(v) for v in ms]))

    def status(self):
        import torch
        s = ctypes.c_int32(-1)
        st = _raw_stream(self.device)
        _lib.check(self._lib.bt_ba_status(self.plan.handle, self.ws.data_ptr(), st, ctypes.byref(s)), "bt_ba_status")
        return s.value
