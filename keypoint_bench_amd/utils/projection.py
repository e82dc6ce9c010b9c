"""Drop-in for the homography branch of the reference's utils/projection.py (`warp`, `warp_homography`,
lines 137-167 and 186-193), computed by csrc/covis.hip through libkpb.so."""
import torch

from .._lib import Context, ptr


def _scalar(v):
    return int(v.item()) if torch.is_tensor(v) else int(v)


def warp_homography_device(kpts0: torch.Tensor, params: dict):
    """The device call without a host read-back: returns (kps0_valid[n,2], kps01_valid[n,2], ids[n], n_valid[1]);
    rows past n_valid of the first two are undefined and ids[n_valid:] are the rejected rows."""
    if not kpts0.is_cuda:
        raise RuntimeError("keypoint_bench_amd needs CUDA/HIP tensors; there is no CPU path")
    dev = kpts0.device
    p = kpts0.detach().to(torch.float32).contiguous()
    n = p.shape[0]
    hm = torch.as_tensor(params["homography_matrix"], dtype=torch.float32, device=dev).contiguous().reshape(9)
    wh = torch.tensor([_scalar(params["width"]), _scalar(params["height"])], dtype=torch.int32, device=dev)
    a = torch.empty((n, 2), dtype=torch.float32, device=dev)
    b = torch.empty((n, 2), dtype=torch.float32, device=dev)
    ids = torch.empty((n,), dtype=torch.int32, device=dev)
    nv = torch.zeros((1,), dtype=torch.int32, device=dev)
    if n:
        ctx = Context.get(dev)
        ctx.check(ctx.lib.kpb_warp_homography(ctx.handle, ptr(p), 1, n, p.shape[1], ptr(None), ptr(hm), ptr(wh), ptr(a),
                                              ptr(b), ptr(ids), ptr(nv)))
    return a, b, ids, nv


def warp_homography(kpts0: torch.Tensor, params: dict):
    """utils/projection.py:137-167.  kpts0 [N,2] normalised; params has 'homography_matrix' [3,3], 'width',
    'height'.  Returns (kpts0_valid, kpts01_valid, ids, ids_out) as the reference does."""
    a, b, ids, nv = warp_homography_device(kpts0, params)
    k = int(nv.item())
    ids = ids.to(torch.int64)
    return a[:k], b[:k], ids[:k], ids[k:]


def warp_se3(kpts0: torch.Tensor, params: dict):
    """utils/projection.py:195-268.  params: pose01 [4,4], bbox0 / bbox1 (row, col), depth0 / depth1 [H,W], intrinsics0 /
    intrinsics1 [3,3].  Returns (kpts0_valid, kpts01_valid, ids_valid, ids_out) as the reference does."""
    if not kpts0.is_cuda:
        raise RuntimeError("keypoint_bench_amd needs CUDA/HIP tensors; there is no CPU path")
    dev = kpts0.device
    f = lambda v: torch.as_tensor(v).detach().to(torch.float32)
    # inverse(intrinsics0) is nine numbers: taken on the host with the routine unproject (projection.py:43) uses
    cam = torch.cat([torch.inverse(f(params["intrinsics0"]).cpu()).reshape(9), f(params["intrinsics1"]).cpu().reshape(9),
                     f(params["pose01"]).cpu().reshape(16), f(params["bbox0"]).cpu().reshape(2), f(params["bbox1"]).cpu().reshape(2)]).to(dev)
    d0, d1 = f(params["depth0"]).to(dev).contiguous(), f(params["depth1"]).to(dev).contiguous()
    p = kpts0.detach().to(torch.float32).contiguous()
    n = p.shape[0]
    a = torch.empty((n, 2), dtype=torch.float32, device=dev); b = torch.empty((n, 2), dtype=torch.float32, device=dev)
    ids = torch.empty((n,), dtype=torch.int32, device=dev); out = torch.empty((n,), dtype=torch.int32, device=dev)
    cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
    if n:
        ctx = Context.get(dev)
        ctx.check(ctx.lib.kpb_warp_se3(ctx.handle, ptr(p), n, p.shape[1], ptr(d0), d0.shape[0], d0.shape[1], ptr(d1), d1.shape[0], d1.shape[1],
                                       ptr(cam), ptr(a), ptr(b), ptr(ids), ptr(out), ptr(cnt)))
    k, m = (int(v) for v in cnt.tolist())
    return a[:k], b[:k], ids[:k].to(torch.int64), out[:m].to(torch.int64)


def warp(kpts0: torch.Tensor, params: dict):
    """utils/projection.py:186-193."""
    mode = params["mode"]
    if mode == "homo":
        return warp_homography(kpts0[:, 0:2], params)
    if mode == "se3":
        return warp_se3(kpts0[:, 0:2], params)
    raise ValueError("unknown mode!")
