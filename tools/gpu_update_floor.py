#!/usr/bin/env python3
"""What an update() of the replay costs beyond its GPU work: in the steady state every update()'s 2 x ITER BA calls are timed as the
caller times them (sync, calls, sync), then the SAME calls are repeated on the same list (every plan cached: nothing but argument
handling and launches on the host) and timed again.  The difference is what the new plan of the first call costs on the critical
path.  GPU box:  python tools/gpu_update_floor.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from batrack_amd.backend import ba as hip_ba
from batrack_amd.sequence import SlamConfig, SyntheticObservations, WindowedBA

rec = dict(first=[], again=[], calls=[], bound=[])
from batrack_amd import plan as plan_mod
_bind = plan_mod.Plan.bind
state = dict(bound=False, k=0)


def counting_bind(self, *a, **k):
    r = _bind(self, *a, **k)
    state["bound"] = state["bound"] or bool(r)
    return r


plan_mod.Plan.bind = counting_bind
import torch
slow = []                                   # (what, update index, ms) of any wrapped call that took more than 2 ms


def timed(obj, name, what):
    f = getattr(obj, name)

    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); dt = (time.perf_counter() - t) * 1e3
        if dt > 2.0:
            slow.append((what, len(rec["first"]), round(dt, 2)))
        return r
    setattr(obj, name, w)


timed(plan_mod.Plan, "confirm", "Plan.confirm")
timed(plan_mod.Plan, "shifted_spec", "Plan.shifted_spec")
timed(plan_mod.Plan, "shifted_any", "Plan.shifted_any")
timed(plan_mod.Plan, "__init__", "Plan.__init__")
timed(plan_mod.Stepper, "__init__", "Stepper.__init__")
timed(torch.cuda, "synchronize", "synchronize")
timed(hip_ba, "_preshift", "_preshift")
timed(hip_ba, "_plan_lookup", "_plan_lookup")
timed(hip_ba, "_store", "_store")
AB = os.environ.get("AB", "0") == "1"          # alternate: clones made ahead (Plan.preshift) on / off, update() by update()
pf = hip_ba.prefetch_plan if os.environ.get("PREFETCH", "0") == "1" else None
obs = SyntheticObservations(n_frames=int(os.environ.get("FRAMES", 200)), M=256, seed=0)
trk = WindowedBA(obs, hip_ba.BA_rgbd_droid, SlamConfig(PATCHES_PER_FRAME=256, BUFFER_SIZE=1024), device="cuda:0", prefetch=pf)
orig_ba = trk.ba
log = []


def recording_ba(*a, **k):
    log.append((a, k))
    return orig_ba(*a, **k)


orig_update = trk.update


def update():
    log.clear()
    trk.ba = recording_ba
    t0 = trk.stats["ba_seconds"]
    if AB:
        state["k"] += 1
        if os.environ.get("AB_AT"):                # instead: the call of the update() behind which the clone is made, two settings alternating
            a0, a1 = os.environ["AB_AT"].split(",")
            os.environ["BT_PLAN_PRESHIFT_AT"] = a1 if (state["k"] // 2) % 2 else a0
            rec.setdefault("at", []).append(os.environ["BT_PLAN_PRESHIFT_AT"])
        else:
            os.environ["BT_PLAN_PRESHIFT"] = "1" if (state["k"] // 2) % 2 else "0"       # (two updates each: the clone serves the update AFTER it was made)
    orig_update()
    rec["first"].append(trk.stats["ba_seconds"] - t0)
    rec["bound"].append(state["bound"])
    state["bound"] = False                  # (with prefetch the clone is bound BEFORE update(), where the caller hands over the list)
    trk.ba = orig_ba
    calls = list(log)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for a, k in calls:                       # (results discarded: the state of the replay is the first pass's)
        orig_ba(*a, **k)
    torch.cuda.synchronize()
    rec["again"].append(time.perf_counter() - t)
    per = []
    for a, k in calls[:2]:                   # one pose+structure and one structure-only call on their own
        torch.cuda.synchronize(); t = time.perf_counter(); orig_ba(*a, **k); torch.cuda.synchronize(); per.append(time.perf_counter() - t)
    rec["calls"].append(per)


trk.update = update
trk.run()
if AB and "at" in rec:
    f, at = np.array(rec["first"][-160:]) * 1e3, np.array(rec["at"][-160:])
    # (the setting of update k decides where the clone for update k + 1 is made: it costs update k its host time, and update k + 1 is
    #  served the same either way — so update k's own time is what differs)
    for v in sorted(set(at)):
        print(f"clone made behind call {v} of the update(): update() median {np.median(f[at == v]):.3f} ms, p10 {np.percentile(f[at == v], 10):.3f}, p90 {np.percentile(f[at == v], 90):.3f} (n = {int((at == v).sum())})")
elif AB:
    f, b = np.array(rec["first"][-160:]) * 1e3, np.array(rec["bound"][-160:])
    print(f"update() as the caller times it, median over the last 160 updates by how its first call got its plan: a clone made ahead, bound by one kernel "
          f"{np.median(f[b]):.3f} ms (n = {int(b.sum())}); a clone made in the call (shifted_spec) {np.median(f[~b]):.3f} ms (n = {int((~b).sum())})")
    allf = np.array(rec["first"]) * 1e3
    top = np.argsort(allf)[-8:][::-1]
    print("  slowest updates (index of update, ms, first call bound a clone made ahead): " + ", ".join(f"#{i} {allf[i]:.2f} {rec['bound'][i]}" for i in top))
    print("  calls over 2 ms (what, update, ms): " + ", ".join(f"{w} #{i} {d}" for w, i, d in slow[:24]))
    q = lambda v: " ".join(f"{x:.3f}" for x in np.percentile(v, [10, 50, 90, 99])) + f" mean {v.mean():.3f}"
    print(f"  percentiles 10 / 50 / 90 / 99 and mean: made ahead {q(f[b])}; made in the call {q(f[~b])}")
allf = np.array(rec["first"]) * 1e3
top = np.argsort(allf)[-6:][::-1]
print("slowest updates (#, ms): " + ", ".join(f"#{i} {allf[i]:.2f}" for i in top) + " | calls over 2 ms: " + ", ".join(f"{w} #{i} {d}" for w, i, d in slow[:12]))
f, a = np.array(rec["first"][-100:]) * 1e3, np.array(rec["again"][-100:]) * 1e3
c = np.array(rec["calls"][-100:]) * 1e6
print(f"prefetch {'on' if pf else 'off'}: update() as the caller times it {np.median(f):.3f} ms; the same {len(log)} calls again, plans cached {np.median(a):.3f} ms; "
      f"difference {1e3 * (np.median(f) - np.median(a)):.0f} us")
print(f"one call alone, synchronised: pose+structure {np.median(c[:, 0]):.1f} us, structure-only {np.median(c[:, 1]):.1f} us")
