"""Host-side mirror of the reference's `main/backend` package for the BA path:
same module and symbol names (`ba.BA_rgbd_droid`, `lietorch.SE3`), HIP kernels
underneath through the C ABI in include/batrack_ba.h."""
