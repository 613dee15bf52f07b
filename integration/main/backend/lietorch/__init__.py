"""`main.backend.lietorch` (/root/reference/main/backend/lietorch/__init__.py): SE3, stack, cat from batrack_amd; the
groups BA-Track's inference path never touches import fine and raise on use."""
from batrack_amd.backend.lietorch import SE3, cat, stack  # noqa: F401

__all__ = ["groups"]


def _absent(name):
    class _Absent:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"lietorch.{name} is outside the BA hot path and is not provided by batrack_amd "
                                      "(SURVEY.md §2 row 4); SE3 is")
    _Absent.__name__ = _Absent.__qualname__ = name
    return _Absent


SO3, RxSO3, Sim3, LieGroupParameter = (_absent(n) for n in ("SO3", "RxSO3", "Sim3", "LieGroupParameter"))
