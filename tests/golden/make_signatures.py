#!/usr/bin/env python3
"""Record the INTERFACE of the reference's backend package as data (tests/golden/signatures.json): the call signatures of
the functions /root/reference/main/batrack.py uses through `main.backend.*`, and the names it reaches on the SE3 class and
the lietorch / altcorr packages.  Run in the build container only (needs /root/reference), same stand-ins as
make_golden.py.  Names and signature strings only — no reference source is written out.

    python tests/golden/make_signatures.py
"""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "refstubs"), os.path.join(REF, "main"), REF]

import backend.ba as ref_ba                      # noqa: E402  (reference, unmodified)
import backend.projective_ops as ref_pops        # noqa: E402
import backend.lietorch as ref_lie               # noqa: E402

sig = lambda f: str(inspect.signature(f))
out = {
    "source": "wrchen530/batrack main/backend (signatures only)",
    "ba": {"BA_rgbd_droid": sig(ref_ba.BA_rgbd_droid)},
    "projective_ops": {n: sig(getattr(ref_pops, n)) for n in
                       ("transform", "point_cloud", "proj", "iproj", "flow_mag", "back_proj", "proj_to_frames", "coords_grid")},
    "lietorch": {"module_names": sorted(n for n in ("SE3", "SO3", "RxSO3", "Sim3", "LieGroupParameter", "cat", "stack") if hasattr(ref_lie, n)),
                 "stack": sig(ref_lie.stack), "cat": sig(ref_lie.cat),
                 # what batrack.py / ba.py / projective_ops.py call on an SE3 (batrack.py:179-184,337,864,883,905-906,1042,1086-1087)
                 "SE3_used": ["data", "vec", "inv", "matrix", "exp", "log", "retr", "adjT", "act", "mul", "__mul__", "__getitem__",
                              "Identity", "translation", "detach", "shape", "device"]},
    # altcorr's compiled half cannot be imported here (CUDA extension); batrack.py:321-323,438 call patchify(net, coords, radius[, mode])
    "altcorr": {"patchify_call": ["net", "coords", "radius", "mode='bilinear'"]},
    "batrack_imports": ["from main.backend import altcorr, lietorch", "from main.backend.lietorch import SE3",
                        "from main.backend import projective_ops as pops", "from main.backend.ba import BA_rgbd_droid"],
}
for n in out["lietorch"]["SE3_used"]:
    assert hasattr(ref_lie.SE3, n) or n in ("data", "shape", "device"), n
json.dump(out, open(os.path.join(HERE, "signatures.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1)[:600])
