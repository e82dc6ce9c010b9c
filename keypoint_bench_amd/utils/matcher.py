"""Drop-in for the brute-force branch of the reference's utils/matcher.py (lines 206-234), computed by
csrc/match.hip through libkpb.so."""
import ctypes

import torch

from .._lib import Context, MatchParams, ptr


def sample_descriptors(pts: torch.Tensor, desc_map, n=None) -> torch.Tensor:
    """utils/matcher.py:221-226: bilinear grid_sample(align_corners=True) of desc_map [1,C,Hd,Wd] at
    pts [N, >=2] (x, y normalised) -> [N, C].  desc_map may be a LazyDescriptors handle (models/)."""
    if hasattr(desc_map, "sample"):
        return desc_map.sample(pts)
    if not desc_map.is_cuda:
        raise RuntimeError("keypoint_bench_amd needs CUDA/HIP tensors; there is no CPU path")
    d = desc_map.detach()
    if d.dtype != torch.float32:
        d = d.float()
    p = pts.detach().to(torch.float32).contiguous()
    _, C, Hd, Wd = d.shape
    N = p.shape[0]
    out = torch.empty((N, C), dtype=torch.float32, device=d.device)
    if N:
        ctx = Context.get(d.device)
        sb, sc, sh, sw = d.stride()
        ctx.check(ctx.lib.kpb_sample(ctx.handle, ptr(d), 1, C, Hd, Wd, sb, sc, sh, sw, ptr(p), p.shape[1], N,
                                     ptr(None), ptr(out)))
    return out


def match_descriptors(desc0: torch.Tensor, desc1: torch.Tensor, metric="euclidean", max_distance=float("inf"),
                      cross_check=True, return_distance=False):
    """skimage.feature.match_descriptors as the reference calls it (matcher.py:227-230), on device.
    Returns int64 [K, 2] (and float64 [K] distances when asked)."""
    if metric != "euclidean":
        raise NotImplementedError("only metric='euclidean' (config/config_MHA.yaml:83)")
    a = desc0.detach().to(torch.float32).contiguous()
    b = desc1.detach().to(torch.float32).contiguous()
    n, m = a.shape[0], b.shape[0]
    dev = a.device
    if n == 0 or m == 0:
        e = torch.zeros((0, 2), dtype=torch.int64, device=dev)
        return (e, torch.zeros((0,), dtype=torch.float64, device=dev)) if return_distance else e
    ctx = Context.get(dev)
    pairs = torch.empty((n, 2), dtype=torch.int32, device=dev)
    dist = torch.empty((n,), dtype=torch.float64, device=dev)
    k = torch.empty((1,), dtype=torch.int32, device=dev)
    prm = MatchParams(float(max_distance), 1 if cross_check else 0)
    ctx.check(ctx.lib.kpb_match(ctx.handle, ptr(a), ptr(b), 1, a.shape[1], n, m, ptr(None), ptr(None),
                                ctypes.byref(prm), ptr(pairs), ptr(dist), ptr(k)))
    kk = int(k.item())
    pairs = pairs[:kk].to(torch.int64)
    return (pairs, dist[:kk].clone()) if return_distance else pairs


def brute_force_matcher(pts0: torch.Tensor, pts1: torch.Tensor, desc_map_0, desc_map_1, params=None):
    """utils/matcher.py:206-234.  pts0 (n, >=2), pts1 (m, >=2) in [0,1]; returns the matched rows of pts0 and
    pts1 (all columns kept), ordered by ascending index into pts0."""
    desc0 = sample_descriptors(pts0, desc_map_0)
    desc1 = sample_descriptors(pts1, desc_map_1)
    matches = match_descriptors(desc0, desc1, metric=params["metric"], max_distance=params["max_distance"],
                                cross_check=params["cross_check"])
    k = matches.shape[0]
    if k == 0:
        return pts0[:0], pts1[:0]
    ctx = Context.get(pts0.device)
    m32 = matches.to(torch.int32).contiguous()
    kk = torch.tensor([k], dtype=torch.int32, device=pts0.device)
    outs = []
    for col, pts in ((0, pts0), (1, pts1)):
        p = pts.detach().to(torch.float32).contiguous()
        out = torch.empty((k, p.shape[1]), dtype=torch.float32, device=p.device)
        ctx.check(ctx.lib.kpb_gather_rows(ctx.handle, ptr(p), 1, p.shape[0], p.shape[1], ptr(m32), k, 2, col, ptr(kk),
                                          ptr(out)))
        outs.append(out)
    return outs[0], outs[1]
