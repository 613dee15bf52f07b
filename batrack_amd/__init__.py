"""batrack_amd — MI355X-native bundle-adjustment backend for BA-Track's sparse-SLAM
hot path (drop-in for main/backend/ba.py:BA_rgbd_droid of wrchen530/batrack).

  batrack_amd.backend.ba.BA_rgbd_droid      the reference's entry point, HIP underneath
  batrack_amd.backend.lietorch.SE3          pose wrapper the caller passes / receives
  batrack_amd.plan.Plan / Stepper           explicit plan + preallocated step objects (torch.ops.batrack_hip underneath)
  batrack_amd.parallel                      track-sharded multi-GPU step (RCCL all-reduce)
  batrack_amd.graphgen                      synthetic factor graphs (inputs only)
"""
from ._lib import build, LIB_PATH  # noqa: F401

__all__ = ["build", "LIB_PATH"]
