"""Drop-in for the reference's utils/extracter.py: same names, same arguments, same outputs, computed by
the HIP kernels of csrc/detect.hip through libkpb.so.

    from keypoint_bench_amd.utils.extracter import detection      # instead of utils.extracter
"""
import ctypes

import torch

from .._lib import Context, DetectParams, ptr


def _as_maps(t: torch.Tensor):
    if t.dim() != 4 or t.shape[1] != 1:
        raise ValueError("score map must be B x 1 x H x W, got %s" % (tuple(t.shape),))
    if not t.is_cuda:
        raise RuntimeError("keypoint_bench_amd.detection needs a CUDA/HIP tensor (MI355X); there is no CPU path")
    return t.detach().to(torch.float32).contiguous()


def fast_nms(image_probs: torch.Tensor, nms_dist: int = 4, max_iter: int = -1, min_value: float = 0.0) -> torch.Tensor:
    """utils/extracter.py:6-100.  BxCxHxW non-negative map -> same shape, suppressed pixels = 0.
    max_iter / min_value other than the defaults are not supported (the reference never passes them)."""
    if max_iter != -1 or min_value != 0.0:
        raise NotImplementedError("fast_nms: only max_iter=-1, min_value=0.0 (the values detection() uses)")
    if nms_dist == 0:
        return image_probs
    if not image_probs.is_cuda:
        raise RuntimeError("keypoint_bench_amd.fast_nms needs a CUDA/HIP tensor")
    x = image_probs.detach().to(torch.float32).contiguous()
    B, C, H, W = x.shape
    ctx = Context.get(x.device)
    out = torch.empty_like(x)
    ctx.check(ctx.lib.kpb_fast_nms(ctx.handle, ptr(x), B * C, H, W, int(nms_dist), ptr(out)))
    return out


def detection_batch(score_map: torch.Tensor, params: dict = None, sync: bool = True):
    """All batch elements at once: returns (kps [B, top_k, 3], flat_idx [B, top_k], n [B]) device tensors."""
    if params is None:  # utils/extracter.py:200-205
        params = dict(nms_dist=4, threshold=0.0, border_dist=8, top_k=300, min_score=0.0)
    x = _as_maps(score_map)
    B, _, H, W = x.shape
    ctx = Context.get(x.device)
    top_k = min(int(params["top_k"]), H * W)
    prm = DetectParams(int(params["nms_dist"]), float(params["threshold"]), int(params["border_dist"]), top_k,
                       float(params["min_score"]))
    kps = torch.empty((B, top_k, 3), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, top_k), dtype=torch.int32, device=x.device)
    n = torch.empty((B,), dtype=torch.int32, device=x.device)
    ctx.check(ctx.lib.kpb_detect(ctx.handle, ptr(x), B, H, W, ctypes.byref(prm), ptr(kps), ptr(idx), ptr(n),
                                 1 if sync else 0))
    return kps, idx, n


def detection(score_map: torch.Tensor, params: dict = None):
    """utils/extracter.py:193-221.  score_map Bx1xHxW -> Nx3 (x, y, prob) of batch element 0."""
    kps, _, _ = detection_batch(score_map[:1], params, sync=True)
    ctx = Context.get(kps.device)
    n = (ctypes.c_int32 * 1)()
    ctx.check(ctx.lib.kpb_detect_counts(ctx.handle, n, 1))      # host integer left by the kernels: no second read-back (n[0].item())
    return kps[0, : int(n[0])].clone()
