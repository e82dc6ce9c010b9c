"""Drop-in for the brute-force branch of the reference's utils/matcher.py (lines 206-234), computed by
csrc/match.hip through libkpb.so."""
import ctypes

import torch

from .._lib import Context, LkParams, MatchParams, ptr


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


def _match32(desc0: torch.Tensor, desc1: torch.Tensor, max_distance, cross_check, want_dist=False):
    """kpb_match for one pair: (pairs int32 [n, 2] -- the first K rows are the matches --, dist float64 [n] or None, K device
    tensor [1], K as a host integer).  K comes from pinned host memory the kernel wrote (kpb_match_counts): one stream
    synchronisation, no read-back of its own."""
    a = desc0.detach().to(torch.float32).contiguous()
    b = desc1.detach().to(torch.float32).contiguous()
    n, m = a.shape[0], b.shape[0]
    dev = a.device
    ctx = Context.get(dev)
    pairs = torch.empty((n, 2), dtype=torch.int32, device=dev)
    dist = torch.empty((n,), dtype=torch.float64, device=dev) if want_dist else None
    k = torch.empty((1,), dtype=torch.int32, device=dev)
    prm = MatchParams(float(max_distance), 1 if cross_check else 0)
    ctx.check(ctx.lib.kpb_match(ctx.handle, ptr(a), ptr(b), 1, a.shape[1], n, m, ptr(None), ptr(None),
                                ctypes.byref(prm), ptr(pairs), ptr(dist), ptr(k)))
    kk = (ctypes.c_int32 * 1)()
    ctx.check(ctx.lib.kpb_match_counts(ctx.handle, kk, 1))
    return pairs, dist, k, int(kk[0])


def match_descriptors(desc0: torch.Tensor, desc1: torch.Tensor, metric="euclidean", max_distance=float("inf"),
                      cross_check=True, return_distance=False):
    """skimage.feature.match_descriptors as the reference calls it (matcher.py:227-230), on device.
    Returns int64 [K, 2] (and float64 [K] distances when asked)."""
    if metric != "euclidean":
        raise NotImplementedError("only metric='euclidean' (config/config_MHA.yaml:83)")
    dev = desc0.device
    if desc0.shape[0] == 0 or desc1.shape[0] == 0:
        e = torch.zeros((0, 2), dtype=torch.int64, device=dev)
        return (e, torch.zeros((0,), dtype=torch.float64, device=dev)) if return_distance else e
    pairs, dist, _, kk = _match32(desc0, desc1, max_distance, cross_check, want_dist=True)
    pairs = pairs[:kk].to(torch.int64)
    return (pairs, dist[:kk].clone()) if return_distance else pairs


def brute_force_matcher(pts0: torch.Tensor, pts1: torch.Tensor, desc_map_0, desc_map_1, params=None):
    """utils/matcher.py:206-234.  pts0 (n, >=2), pts1 (m, >=2) in [0,1]; returns the matched rows of pts0 and
    pts1 (all columns kept), ordered by ascending index into pts0."""
    if params["metric"] != "euclidean":
        raise NotImplementedError("only metric='euclidean' (config/config_MHA.yaml:83)")
    desc0 = sample_descriptors(pts0, desc_map_0)
    desc1 = sample_descriptors(pts1, desc_map_1)
    if desc0.shape[0] == 0 or desc1.shape[0] == 0:
        return pts0[:0], pts1[:0]
    # the int32 pairs and the device count feed the row gather as they are (r03 converted to int64 and back and uploaded K again)
    m32, _, k_dev, k = _match32(desc0, desc1, params["max_distance"], params["cross_check"])
    if k == 0:
        return pts0[:0], pts1[:0]
    ctx = Context.get(pts0.device)
    outs = []
    for col, pts in ((0, pts0), (1, pts1)):
        p = pts.detach().to(torch.float32).contiguous()
        out = torch.empty((k, p.shape[1]), dtype=torch.float32, device=p.device)
        ctx.check(ctx.lib.kpb_gather_rows(ctx.handle, ptr(p), 1, p.shape[0], p.shape[1], ptr(m32), k, 2, col, ptr(k_dev),
                                          ptr(out)))
        outs.append(out)
    return outs[0], outs[1]


class OpticalFlow(object):
    """utils/matcher.py:7-142: pyramidal Lucas-Kanade on win_size x win_size x C windows, computed by csrc/lk.hip (one
    wave per keypoint, nothing unfolded).  Same constructor keys, same call, same return: (points [1,N,2] in PIXELS of
    the full image, error [1,N])."""

    def __init__(self, params=None):
        if params is None:
            params = {"distance": 3, "win_size": 3, "levels": 1, "interation": 40, "gray": False}      # matcher.py:9-16
        self.distance, self.win_size, self.levels = params["distance"], params["win_size"], params["levels"]
        self.interation, self.gray = params["interation"], params["gray"]

    def __call__(self, img1, img2, pts1, pts2, random_angle=None):
        if not img1.is_cuda:
            raise RuntimeError("keypoint_bench_amd needs CUDA/HIP tensors; there is no CPU path")
        N, C, H, W = img1.shape
        if N != 1:
            raise ValueError("one image pair per call (the reference indexes batch element 0, matcher.py:60)")
        if C != (1 if self.gray else 3):        # the reference's Sobel weight is [C,C,3,3] with C fixed by `gray` (26-36)
            raise ValueError("images must have %d channel(s) for gray=%s" % (1 if self.gray else 3, self.gray))
        dev = img1.device
        a = img1.detach().to(torch.float32).contiguous()
        b = img2.detach().to(torch.float32).contiguous()
        p1 = pts1.detach().to(torch.float32).contiguous()
        p2 = pts2.detach().to(torch.float32).contiguous()
        n = p1.shape[0]
        if random_angle is None:
            random_angle = torch.randn(n, device=dev) * 6.28                                             # 55
        unit = torch.stack([torch.cos(random_angle), torch.sin(random_angle)], dim=1).to(torch.float32).contiguous()   # 56
        out = torch.empty((n, 2), dtype=torch.float32, device=dev)
        err = torch.empty((n,), dtype=torch.float32, device=dev)
        if n:
            ctx = Context.get(dev)
            prm = LkParams(float(self.distance), int(self.win_size), int(self.levels), int(self.interation))
            ctx.check(ctx.lib.kpb_lk_track(ctx.handle, ptr(a), ptr(b), C, H, W, ptr(p1), ptr(p2), p1.shape[1], ptr(unit), n,
                                           ctypes.byref(prm), ptr(out), ptr(err)))
        return out.unsqueeze(0), err.unsqueeze(0)


def optical_flow_tensor(pts0, pts1, img0, img1, params=None):
    """utils/matcher.py:186-203."""
    pts1_, _ = OpticalFlow(params)(img0, img1, pts0, pts1)
    return pts1_
