"""`main.backend.projective_ops` (/root/reference/main/backend/projective_ops.py:19-175): forwards."""
from batrack_amd.backend.projective_ops import (back_proj, coords_grid, flow_mag, iproj, point_cloud, proj,  # noqa: F401
                                                proj_to_frames, transform)
