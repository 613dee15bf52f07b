"""`main.backend` of wrchen530/batrack, served by batrack_amd (integration/README.md)."""
from . import altcorr, lietorch, projective_ops  # noqa: F401
from . import ba  # noqa: F401
