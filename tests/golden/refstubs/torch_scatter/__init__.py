"""Stand-in for the un-vendored torch-scatter 2.1.2 dependency of the reference
(requirements.txt:10), used ONLY by tests/golden/make_golden.py in the build
container so that the reference's unmodified ba.py imports.  scatter_sum is a
plain sum along `dim` into `dim_size` bins; this is our statement of it."""
import torch


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    dim = dim % src.dim()
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    acc = torch.zeros(shape, dtype=src.dtype, device=src.device)
    return acc.index_add_(dim, index.to(torch.long), src)
