"""`BA_rgbd_droid` — the reference's bundle-adjustment entry point
(/root/reference/main/backend/ba.py:217-339), same signature and argument
meaning, executed by the gfx950 kernels behind include/batrack_ba.h.

Called exactly as BATRACK.update() does (/root/reference/main/batrack.py:869-875):

    Gs, patches = BA_rgbd_droid(Gs, patches, patches_monodisp, intrinsics,
                                targets_3d[..., :2], targets_3d[..., 2:], weights, lmbda,
                                ii, jj, kk, bounds, ep=ep, fixedp=t0,
                                structure_only=..., loss=..., alpha=0.05)

Behaviour kept from the reference: returns NEW tensors (inputs untouched); with
structure_only (or no free pose) the input SE3 object itself is returned
(ba.py:336-339); `targets_disp` is accepted and never read; the whole disparity
buffer is clamped to [1e-3, 10] (ba.py:333); a failed Cholesky leaves the poses
re-normalised but unmoved (ba.py:9-13), nothing raises.

There is no CPU path: tensors must live on a ROCm device and the HIP library
must be built, otherwise this raises.
"""
import os
import weakref
from concurrent.futures import ThreadPoolExecutor

import torch

from .. import _lib
from ..plan import Plan, Stepper
from .lietorch import SE3

_CACHE = {}          # key -> (stepper, (weakref ii, jj, kk))
_CACHE_MAX = 8
_PENDING = {}        # key -> (future, result box, (ii, jj, kk)): plans being built by prefetch_plan
_LAST_SHIFT = [None]  # frame shift of the last plan that was made as a shifted copy
_PRE = {}            # id(source stepper) -> (source stepper, its clone for the NEXT list, made ahead: Plan.preshift) — a handful at most
_WORKER = None       # one long-lived host thread (the planner keeps edge-sized scratch per thread)


def _key(ii, jj, kk, n_buf, p_tot, fixedp, device):
    return (id(ii), id(jj), id(kk), ii._version, jj._version, kk._version,
            ii.data_ptr(), jj.data_ptr(), kk.data_ptr(), ii.numel(), int(fixedp), int(n_buf), int(p_tot), str(device))


def _store(key, stepper, ii, jj, kk):
    if len(_CACHE) >= _CACHE_MAX:
        old = _CACHE.pop(next(iter(_CACHE)))
        _PRE.pop(id(old[0]), None)                # (a clone made ahead from a plan that has left the cache is of no use either)
    _CACHE[key] = (stepper, tuple(weakref.ref(t) for t in (ii, jj, kk)))


_CALLS_BEFORE = [0]  # how many calls the previous edge list got (an update() makes 2 x ITER): decides when the next clone is made
_REPEAT = [False]    # the call in progress found the plan of the call before it in place (not the first call of an update())
_LAST = [None]       # (ii, jj, kk, versions, fixedp, n_buf, p_tot, device, stepper) of the last call: an update() makes 2*ITER calls on one list


def _plan_for(ii, jj, kk, n_buf, p_tot, fixedp, device):
    """One plan per (edge list, fixedp): the caller keeps self.ii/jj/kk alive and
    unmodified across the 2*ITER calls of an update() (batrack.py:869-875) and
    replaces the tensors when edges are appended/removed (batrack.py:189-204)."""
    last = _LAST[0]
    if (last is not None and last[0] is ii and last[1] is jj and last[2] is kk and last[3] == (ii._version, jj._version, kk._version, ii.data_ptr(), jj.data_ptr(), kk.data_ptr())
            and last[4] == fixedp and last[5] == n_buf and last[6] == p_tot and last[7] == device):
        _REPEAT[0] = True
        return last[8]
    _REPEAT[0] = False
    if last is not None:
        _CALLS_BEFORE[0] = last[8].__dict__.get("_repeats", 0) + 1
    stepper = _plan_lookup(ii, jj, kk, n_buf, p_tot, fixedp, device)
    _LAST[0] = (ii, jj, kk, (ii._version, jj._version, kk._version, ii.data_ptr(), jj.data_ptr(), kk.data_ptr()), fixedp, n_buf, p_tot, device, stepper)
    return stepper


def _plan_lookup(ii, jj, kk, n_buf, p_tot, fixedp, device):
    key = _key(ii, jj, kk, n_buf, p_tot, fixedp, device)
    hit = _CACHE.get(key)
    if hit is not None:
        stepper, refs = hit
        if all(r() is t for r, t in zip(refs, (ii, jj, kk))):
            return stepper
        del _CACHE[key]
    pend = _PENDING.pop(key, None)
    if pend is not None:                          # prefetch_plan is building exactly this one: wait for it
        future, box, held = pend
        future.result()
        if "stepper" in box and all(h is t for h, t in zip(held, (ii, jj, kk))):
            _store(key, box["stepper"], ii, jj, kk)
            return box["stepper"]
        if "error" in box and not isinstance(box["error"], Exception):
            raise box["error"]                    # KeyboardInterrupt and the like; ordinary errors are re-raised by the build below
    pl, ws = _build_plan(ii, jj, kk, n_buf, p_tot, fixedp, True)
    stepper = pl if isinstance(pl, Stepper) else Stepper(pl, device, ws)
    _store(key, stepper, ii, jj, kk)
    return stepper


def _speculate(ii, jj, kk, n_buf, p_tot, fixedp, probe=False):
    """The plan of a list ASSUMED to be an earlier one moved up by the shift that was right last time — no synchronisation, no host
    wait, `confirm()` after the first step: a clone made ahead (Plan.preshift, with its stepper) bound to the list by one comparison
    kernel, else a clone made here (Plan.shifted_spec).  (plan or stepper, workspace to share) or None.  `probe`: only say whether a
    clone made ahead is there (its stepper), touching nothing."""
    if _LAST_SHIFT[0] is None or os.environ.get("BT_PLAN_SHIFT", "1") == "0" or os.environ.get("BT_PLAN_SPECULATE", "1") == "0":
        return None
    E, nb, pt, fp = ii.numel(), int(n_buf), int(p_tot), int(fixedp)
    try:
        cached = list(_CACHE.values())
    except RuntimeError:
        return None
    for st, _ in reversed(cached):
        inf = st.plan.info
        if (inf["E"] == E and inf["n_buf"] == nb and inf["p_tot"] == pt and fp - inf["fixedp"] == _LAST_SHIFT[0]
                and not st.plan.__dict__.get("speculative")):
            if probe:                         # (prefetch_plan asking: is there a clone made ahead for this list?  Nothing is touched)
                pre = _PRE.get(id(st))
                return (pre[1], st.ws) if pre is not None and pre[0] is st and pre[1].plan.info["fixedp"] == fp else None
            pre = _PRE.pop(id(st), None)
            if pre is not None and pre[0] is st and pre[1].plan.info["fixedp"] == fp and pre[1].plan.bind(ii, jj, kk, n_buf, p_tot, fixedp):
                return pre[1], st.ws          # made (stepper and all) while the previous update()'s steps ran: only the comparison is left
            pl = Plan.shifted_spec(st.plan, ii, jj, kk, n_buf, p_tot, fixedp)
            if pl is not None:
                return pl, st.ws
            return None
    return None


def _build_plan(ii, jj, kk, n_buf, p_tot, fixedp, sync, speculate=True):
    """A new plan: first as a shifted copy of one of the most recent plans (the caller's window in steady state repeats its
    edge list with all frame / patch indices moved up, batrack.py:189-212 — ~0.1 ms on the device), else from scratch
    (host analysis, ~1 ms for the 138k-edge window).  Returns (plan, workspace to share or None).

    Steady state of the caller's window: the list is the one of an earlier update() moved up by as many frames as `fixedp`
    moved (batrack.py:189-212, :858).  That is ASSUMED of the most recent plan whose fixedp differs by the shift that was right
    last time (Plan.shifted_spec: the clone is enqueued with no synchronisation and no host wait, the comparison that proves the
    assumption runs on the GPU beside it); BA_rgbd_droid confirms after it has enqueued the call's step and repeats the call on a
    properly built plan where the assumption was wrong."""
    if speculate:
        got = _speculate(ii, jj, kk, n_buf, p_tot, fixedp)
        if got is not None:
            return got
    if sync:
        # once, here: the index tensors must be complete before any of the builds below reads them on the plan stream
        # (Plan.shifted may return before it gets to synchronise, so nothing below relies on it having done so)
        torch.cuda.current_stream(ii.device).synchronize()
        sync = False
    if os.environ.get("BT_PLAN_SHIFT", "1") != "0":
        E = ii.numel()
        try:
            cached = list(_CACHE.values())             # (prefetch_plan calls this from its worker thread while the cache may change)
        except RuntimeError:
            cached = []
        nb, pt, fp = int(n_buf), int(p_tot), int(fixedp)
        cands = []
        for st, _ in reversed(cached):                 # most recent first (the figures straight from the plans' dicts: every frame passes here)
            inf = st.plan.info
            if inf["E"] == E and inf["n_buf"] == nb and inf["p_tot"] == pt and inf["fixedp"] < fp:
                cands.append(st.plan)
        # (with a keyframe stride of 2 the match is two updates back: the frame shift that worked last time is tried first)
        cands.sort(key=lambda pl: fp - pl.info["fixedp"] != _LAST_SHIFT[0])
        if cands:                                      # (one comparison pass and one host wait for all of them; a list that matches none is built the ordinary way)
            pl, matched = Plan.shifted_any(cands[:3], ii, jj, kk, n_buf, p_tot, fixedp, sync=False)
            if pl is not None:
                _LAST_SHIFT[0] = fp - matched.info["fixedp"]
                return pl, None
    return Plan(ii, jj, kk, n_buf, p_tot, fixedp, sync=False), None


def _preshift(stepper):
    """Make the clone for the list the NEXT update() will most likely bring — this plan's, moved up by the shift that was right last
    time — now, while this update()'s steps run: called after the step of every call that found its plan in place has been enqueued, it
    picks ONE of them (below) to spend its ~45 us behind; the first call of the next update(), which has the GPU idle behind it, is
    left with one comparison kernel.  Once per plan."""
    stepper._repeats = stepper.__dict__.get("_repeats", 0) + 1          # this is call number _repeats + 1 on the plan
    if stepper.__dict__.get("_pre_tried"):
        return
    # WHEN: behind a call that leaves the host ~45 us it can spend unnoticed, i.e. with that much GPU work queued in front of it.  The
    # first call of an update() is late by its plan; the host catches up by ~10-25 us a call (a structure-only step is 19 us of
    # kernels, a pose+structure one 61): behind call 2 the clone cost the update() 16-37 us (the GPU ran dry before call 3 arrived),
    # behind call 5 or 7 nothing (tools/gpu_update_floor.py, AB_AT).  So: call 5 where the update()s have that many calls (how many the
    # previous list got), else call 3, else 2.  BT_PLAN_PRESHIFT_AT: the measurement's override.
    at = os.environ.get("BT_PLAN_PRESHIFT_AT")
    at = int(at) if at else (5 if _CALLS_BEFORE[0] >= 6 else 3 if _CALLS_BEFORE[0] >= 4 else 2)
    if stepper._repeats + 1 < at:
        return
    df = _LAST_SHIFT[0]
    if df is None or stepper.plan.__dict__.get("speculative"):
        return
    stepper._pre_tried = True
    if os.environ.get("BT_PLAN_SHIFT", "1") == "0" or os.environ.get("BT_PLAN_SPECULATE", "1") == "0" or os.environ.get("BT_PLAN_PRESHIFT", "1") == "0":
        return
    pl = Plan.preshift(stepper.plan, df)
    if pl is not None:
        while len(_PRE) >= 3:
            _PRE.pop(next(iter(_PRE)))
        # (the clone shares its source's workspace, as a clone made in the call does: same layout, same stream, taking turns)
        _PRE[id(stepper)] = (stepper, Stepper(pl, stepper.device, stepper.ws))
        # ... and the room the next update()'s plan will need in the cache is made now (destroying a plan is not free either)
        if len(_CACHE) >= _CACHE_MAX:
            old = next(iter(_CACHE))
            if _CACHE[old][0] is not stepper:
                _PRE.pop(id(_CACHE.pop(old)[0]), None)


def _discard(stepper):
    """A plan whose speculation failed: out of every cache (and the shift that was assumed with it)."""
    for k in [k for k, (st, _) in list(_CACHE.items()) if st is stepper]:
        _CACHE.pop(k, None)
    if _LAST[0] is not None and _LAST[0][8] is stepper:
        _LAST[0] = None
    _LAST_SHIFT[0] = None
    _PRE.clear()                                 # (clones made ahead under the shift that just proved wrong)


def _confirm(stepper):
    """Settle a speculative plan.  True: it is the list's plan.  False: the guess was wrong — the plan is out of every cache.
    An error of the confirmation itself (an index out of range in the new list) also takes the plan out of the caches before
    it is raised again: a caller that catches it and calls once more must not step on the unconfirmed clone."""
    try:
        ok = stepper.plan.confirm()
    except Exception:
        _discard(stepper)
        raise
    if not ok:
        _discard(stepper)
    return ok


def _print_residual(poses, patches, intrinsics, targets_2d, ii, jj, kk, bounds):
    """What the reference prints under PRINT=True (ba.py:244-245): the mean over the edges of |v * r|, r = target - reprojection,
    v = Z > 0.2 and |r| < 250 and inside the bounds (ba.py:228-242) — a debugging aid, computed beside the step (the fused
    reprojection kernel, include/batrack_projective.h) and synchronising like the reference's .item()."""
    from . import projective_ops as pops
    coords, v = pops.transform(poses, patches, intrinsics, ii, jj, kk, valid=True)
    p = coords.shape[3]
    c = coords[..., p // 2, p // 2, :]
    r = targets_2d - c
    v = (v[..., p // 2, p // 2] if v.dim() == 4 else v).reshape(r.shape[:-1]).float()
    v = v * (r.norm(dim=-1) < 250).float()
    v = v * ((c[..., 0] > bounds[0]) & (c[..., 1] > bounds[1]) & (c[..., 0] < bounds[2]) & (c[..., 1] < bounds[3])).float()
    print((r * v[..., None]).norm(dim=-1).mean().item())


def prefetch_plan(ii, jj, kk, n_buf, p_tot, fixedp, device=None, background=True):
    """Build the plan of an edge list ahead of the BA calls that will use it.

    The caller knows `(ii, jj, kk)` and `fixedp` of an `update()` as soon as it has appended / removed factors —
    a whole tracker pass before it calls `BA_rgbd_droid` (batrack.py:983-993: `append_factors`, then
    `predict_target`, then `update`).  Called there, this takes the per-frame plan (host analysis + table upload,
    ~1.5 ms for the 138k-edge window) off the critical path: a host thread builds it while the GPU and the main
    thread are busy; the first `BA_rgbd_droid` call with the same tensors picks it up (or waits for it).
    The index tensors must not be modified in place afterwards (the reference replaces them, it never edits them
    between `append_factors` and `update`)."""
    dev = torch.device(device) if device is not None else ii.device
    if dev.type != "cuda":
        raise RuntimeError("prefetch_plan: the edge list must be on the GPU (no CPU fallback in batrack_amd)")
    if ii.numel() == 0:
        return
    key = _key(ii, jj, kk, n_buf, p_tot, fixedp, dev)
    if key in _CACHE or key in _PENDING:
        return
    _lib.lib()
    # A clone made ahead for this list (Plan.preshift, during the previous update()) waits for it: nothing to build.  It is NOT bound
    # here — measured (tools/gpu_update_floor.py, PREFETCH=1 AB=1): with the comparison launched at prefetch time the update() took
    # 0.412 ms against 0.38-0.39 with the first BA call launching it — in front of the step the comparison also wakes the idle GPU
    # while the host is still preparing the step's launches.
    if _speculate(ii, jj, kk, n_buf, p_tot, fixedp, probe=True) is not None:
        return
    ready = torch.cuda.Event()
    caller_stream = torch.cuda.current_stream(dev)
    ready.record(caller_stream)                                # the indices are complete once this has passed
    box = {}

    def build():
        try:
            ready.synchronize()
            # the current device and stream are per host thread: the workspace is allocated and zero-filled on the stream
            # the steps will run on, so its accumulators are clear before the first of them whatever stream that is
            with torch.cuda.device(dev), torch.cuda.stream(caller_stream):
                box["stepper"] = Stepper(_build_plan(ii, jj, kk, n_buf, p_tot, fixedp, False, speculate=False)[0], dev)
        except BaseException as e:                              # reported (or retried in the open) by _plan_for
            box["error"] = e

    while len(_PENDING) >= 4:                                  # stale prefetches (edge lists that were never used)
        _PENDING.pop(next(iter(_PENDING)))[0].result()
    if background:
        global _WORKER
        if _WORKER is None:
            _WORKER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="batrack-plan")
        _PENDING[key] = (_WORKER.submit(build), box, (ii, jj, kk))
    else:
        build()
        if "error" in box:
            raise box["error"]
        _store(key, box["stepper"], ii, jj, kk)


def clear_plan_cache():
    for fut, _, _ in list(_PENDING.values()):
        fut.result()
    _PENDING.clear()
    _PRE.clear()
    _CACHE.clear()
    _LAST[0] = None


def _f32c(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"BA_rgbd_droid: `{what}` must be on the GPU (no CPU fallback in batrack_amd)")
    if t.dtype != torch.float32:
        raise TypeError(f"BA_rgbd_droid: `{what}` must be float32 (batrack.py:74-91), got {t.dtype}")
    return t


def BA_rgbd_droid(poses, patches, patches_monodisp, intrinsics, targets_2d, targets_disp, weights, lmbda,
                  ii, jj, kk, bounds, ep=100.0, PRINT=False, fixedp=1, structure_only=False,
                  loss='trivial', alpha=0.5):
    _lib.lib()                                   # raises if the HIP library is missing
    P = _f32c(poses.data, "poses")
    if P.dim() != 3 or P.shape[0] != 1 or P.shape[-1] != 7:
        raise ValueError("poses must wrap a [1, N, 7] tensor (batch b = 1, ba.py:218)")
    if patches.dim() == 5 and patches.shape[0] == 1 and patches.shape[2] == 3 and patches.shape[3] == patches.shape[4] and patches.shape[3] > 1:
        # Patch size p > 1 (BA-Track itself uses 1, batrack.py:45).  The reference projects the patch's CENTRE pixel (ba.py:228-230:
        # coords[..., p//2, p//2, :]), takes the depth prior at its first pixel (disps[:, kx, 0, 0], ba.py:307) and adds dZ to the
        # whole disparity plane of the patch (ba.py:332-334).  A patch's disparity plane holds ONE value in the reference's callers
        # (the tracker fills it per patch); for such patches that is the p = 1 step on the centres with the result broadcast
        # back.  Planes that are not constant are refused: the clamp of ba.py:333 would act on each pixel separately.
        pp = patches.shape[3]
        disps = patches[:, :, 2]
        if not bool((disps == disps[..., :1, :1]).all()):
            raise NotImplementedError("BA_rgbd_droid: patches of size > 1 whose disparity plane is not constant over the patch")
        centre = patches[:, :, :, pp // 2, pp // 2].contiguous()[..., None, None]
        new_poses, out_c = BA_rgbd_droid(poses, centre, patches_monodisp, intrinsics, targets_2d, targets_disp, weights, lmbda,
                                         ii, jj, kk, bounds, ep=ep, PRINT=PRINT, fixedp=fixedp, structure_only=structure_only,
                                         loss=loss, alpha=alpha)
        out = patches.clone()
        out[:, :, 2] = out_c[:, :, 2].expand(-1, -1, pp, pp)
        return new_poses, out
    if patches.dim() < 3 or patches.shape[0] != 1 or patches.shape[2] != 3 or patches.numel() != 3 * patches.shape[1]:
        raise ValueError("patches must be [1, P_tot, 3, p, p] (batrack.py:45: p = 1)")
    if loss not in _lib.LOSS:
        raise NotImplementedError(loss)              # ba.py:98-99
    n_buf, p_tot, E = P.shape[1], patches.shape[1], ii.numel()
    if E == 0:
        raise ValueError("empty edge list")
    dev = P.device
    for t, what in ((patches, "patches"), (patches_monodisp, "patches_monodisp"), (intrinsics, "intrinsics"), (targets_2d, "targets_2d"), (weights, "weights")):
        _f32c(t, what)
    if patches_monodisp.numel() != p_tot or intrinsics.numel() != 4 * n_buf:
        raise ValueError("patches_monodisp / intrinsics do not match the patch / pose buffers")
    if targets_2d.shape[-1] != 2 or targets_2d.numel() != 2 * E:
        raise ValueError("targets_2d must be [1, E, 2]")
    stepper = _plan_for(ii, jj, kk, n_buf, p_tot, fixedp, dev)
    lmbda_in = lmbda                             # (what a repeated call after a failed speculation must be given again)
    lm_trk = None
    if isinstance(lmbda, torch.Tensor):
        if lmbda.numel() == 1:
            lmbda = float(lmbda)
        else:
            # a per-track tensor is checked against the plan's track count: a speculative clone carries its SOURCE's count, so
            # the speculation is settled first (a wrong guess is rebuilt here, before anything is enqueued)
            if stepper.plan.__dict__.get("speculative") and not _confirm(stepper):
                stepper = _plan_for(ii, jj, kk, n_buf, p_tot, fixedp, dev)
        if isinstance(lmbda, float):
            pass
        elif lmbda.numel() == stepper.plan.m:           # ba.py:299-300: lmbda.reshape(*C.shape), one value per distinct track
            lm_trk = _f32c(lmbda, "lmbda").reshape(-1).contiguous()
            lmbda = 0.0
        else:
            raise ValueError(f"a lmbda tensor must hold 1 or m = {stepper.plan.m} values (ba.py:299-300), got {lmbda.numel()}")
    so = bool(structure_only) or stepper.plan.n == 0
    if stepper._ops is not None:
        # the whole call in ONE operator (csrc/torch_ops.cpp ba_droid: the views the ABI needs, the output tensors, the step):
        # the tensor operations below cost 28 us of host time a call, more than a structure-only step takes on the GPU
        poses_out, out_patches = stepper._ops.ba_droid.default(stepper.plan.handle.value, stepper.ws, P, patches, patches_monodisp, intrinsics,
                                                       targets_2d, weights, [float(b) for b in bounds], float(lmbda), float(ep),
                                                       float(alpha), _lib.LOSS[loss], so, lm_trk)
        if PRINT:
            _print_residual(poses, patches, intrinsics, targets_2d, ii, jj, kk, bounds)
        if stepper.plan.__dict__.get("speculative") and not _confirm(stepper):
            # the list was no shifted copy after all: what was just enqueued is void (the inputs are untouched) — once more, properly
            return BA_rgbd_droid(poses, patches, patches_monodisp, intrinsics, targets_2d, targets_disp, weights, lmbda_in, ii, jj, kk, bounds,
                                 ep=ep, PRINT=False, fixedp=fixedp, structure_only=structure_only, loss=loss, alpha=alpha)
        if _REPEAT[0]:
            _preshift(stepper)                   # (behind this call's launches: the clone for the next update()'s list)
        return (poses, out_patches) if so else (SE3(poses_out), out_patches)
    Pc = P.contiguous()
    pat = patches.reshape(p_tot, 3).contiguous()
    mono = patches_monodisp
    if mono.numel() == p_tot and not mono.is_contiguous() and p_tot > 1:
        # the caller's prior is a strided view (patches_local[:, :, mid, 2:], batrack.py:866): used in place through mono_stride
        # (only a genuine stride >= 1: an expanded tensor — stride 0 — or a negative stride is materialised instead)
        d = [i for i, n in enumerate(mono.shape) if n == p_tot]
        if len(d) == 1 and mono.stride(d[0]) >= 1:
            mono = torch.as_strided(mono, (p_tot,), (mono.stride(d[0]),), mono.storage_offset())
        else:
            mono = mono.reshape(-1).contiguous()
    else:
        mono = mono.reshape(-1)
    intr = intrinsics.reshape(-1, 4).contiguous()
    tg = targets_2d
    tg = tg.reshape(E, 2) if tg.is_contiguous() else tg[0]
    if tg.stride(1) != 1:                          # the caller's view has strides (3, 1): used in place
        tg = tg.contiguous()
    w = weights.reshape(E, 2).contiguous()
    patches_out = torch.empty_like(pat)
    poses_out = Pc if so else torch.empty_like(Pc)
    stepper.step(Pc, pat, mono, intr, tg, tg.stride(0), w, poses_out, patches_out,
                 bounds, lmbda, ep, alpha, loss, so, lmbda_per_track=lm_trk)
    if PRINT:
        _print_residual(poses, patches, intrinsics, targets_2d, ii, jj, kk, bounds)
    if stepper.plan.__dict__.get("speculative") and not _confirm(stepper):
        return BA_rgbd_droid(poses, patches, patches_monodisp, intrinsics, targets_2d, targets_disp, weights, lmbda_in, ii, jj, kk, bounds,
                             ep=ep, PRINT=False, fixedp=fixedp, structure_only=structure_only, loss=loss, alpha=alpha)
    if _REPEAT[0]:
        _preshift(stepper)
    out_patches = patches_out.view(1, p_tot, 3, 1, 1)
    if so:
        return poses, out_patches
    return SE3(poses_out.view(1, n_buf, 7)), out_patches
