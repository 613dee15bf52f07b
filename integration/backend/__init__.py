"""The bare `backend` root the reference's ba.py imports through (`import backend.projective_ops`, ba.py:3): the same
modules as `main.backend`."""
import sys

import main.backend as _b
from main.backend import altcorr, ba, lietorch, projective_ops  # noqa: F401

for _n in ("altcorr", "ba", "lietorch", "projective_ops"):
    sys.modules[__name__ + "." + _n] = getattr(_b, _n)
