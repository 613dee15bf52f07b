"""`main.backend.ba` (/root/reference/main/backend/ba.py:217): forwards to the gfx950 backend."""
from batrack_amd.backend.ba import BA_rgbd_droid, clear_plan_cache, prefetch_plan  # noqa: F401
