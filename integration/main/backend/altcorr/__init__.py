"""`main.backend.altcorr` (/root/reference/main/backend/altcorr/__init__.py): patchify from batrack_amd; `corr` is dead
code in the reference (no caller) and raises."""
from batrack_amd.backend.altcorr import patchify  # noqa: F401


def corr(*a, **k):
    raise NotImplementedError("altcorr.corr has no caller in BA-Track (SURVEY.md §2b) and is not provided by batrack_amd")
