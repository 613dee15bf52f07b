"""`altcorr.patchify` with the reference's signature
(/root/reference/main/backend/altcorr/correlation.py:49-68), HIP underneath
(batrack_amd/csrc/patchify_kernels.hip through include/batrack_patchify.h).

    patchify(net [B,C,H,W], coords [B,M,2], radius, mode='bilinear') -> [B,M,C,d,d], d = 2*radius+1
    mode != 'bilinear'                                              -> [B,M,C,D,D], D = 2*radius+2

Inference only (no autograd), float32, GPU tensors; no CPU fallback.  `corr` is dead code in the
reference's caller and is not provided."""
import torch

from .. import _lib


def patchify(net, coords, radius, mode='bilinear'):
    if not (net.is_cuda and coords.is_cuda):
        raise RuntimeError("altcorr.patchify: tensors must be on the GPU (no CPU fallback in batrack_amd)")
    if net.dim() != 4 or coords.dim() != 3 or coords.shape[-1] != 2 or coords.shape[0] != net.shape[0]:
        raise ValueError("altcorr.patchify: net [B,C,H,W], coords [B,M,2]")
    netf = net.detach().float().contiguous()
    cf = coords.detach().float().contiguous()
    B, C, H, W = netf.shape
    M = cf.shape[1]
    bil = 1 if mode == 'bilinear' else 0
    d = 2 * radius + 1 if bil else 2 * radius + 2
    out = torch.empty((B, M, C, d, d), dtype=torch.float32, device=netf.device)
    L = _lib.lib()
    st = torch.cuda.current_stream(netf.device).cuda_stream
    _lib.check(L.bt_patchify(netf.data_ptr(), B, C, H, W, cf.data_ptr(), M, int(radius), bil, out.data_ptr(), st), "bt_patchify")
    return out
