"""Robust two-view geometry on the device: what the reference gets from OpenCV --
`cv2.findHomography(pts0, pts1, cv2.RANSAC)` (tasks/MHA.py:45-47), `cv2.findEssentialMat` + `cv2.recoverPose`
(tasks/AUC.py:50-64), `cv2.findFundamentalMat(pts0, pts1, cv2.FM_RANSAC)` (utils/mvg.py:16, the reference module this one
stands in for: `fundamental_estimate` keeps its name and return shape) -- computed by csrc/geometry.hip through libkpb.so.

PARITY UNPINNED: cv2 is a third-party dependency, absent from the reference tree and from this image, and its RANSAC is
driven by its own RNG; the kernels restate OpenCV's published algorithm with its default parameters and are validated
against analytic ground truth and the numpy restatement in oracle/geometry_ref.py (tests/test_gpu_geometry.py)."""
import ctypes

import numpy as np
import torch

from .._lib import Context, RansacParams, ptr


def find_homography(m0, m1, scale, k_dev=None, seeds=None, seed=0, threshold=3.0, confidence=0.995, max_iters=2000, refine=True):
    """B independent estimations in one launch.  m0, m1 [B, K, c>=2] (or [K, c]): matched rows, normalised (x, y, ...);
    scale [B, 4] or [4] = (sx0, sy0, sx1, sy1) normalised -> pixels; k_dev int32 [B] valid rows (None: all K);
    seeds: per-pair sampler seeds (int sequence / tensor), default `seed` for every pair.
    Returns (H [B, 3, 3] float64, mask [B, K] uint8, info [B, 4] int32 = found, inliers, hypotheses, 0) on the device."""
    a = m0.detach().to(torch.float32).contiguous()
    b = m1.detach().to(torch.float32).contiguous()
    if a.dim() == 2:
        a, b = a[None], b[None]
    dev = a.device
    if not a.is_cuda:
        raise RuntimeError("keypoint_bench_amd needs CUDA/HIP tensors; there is no CPU path")
    B, K = a.shape[0], a.shape[1]
    sc = torch.as_tensor(scale, dtype=torch.float32).to(dev).reshape(-1, 4)
    if sc.shape[0] == 1 and B > 1:
        sc = sc.expand(B, 4)
    sc = sc.contiguous()
    sd = None
    if seeds is not None:       # uint32 bit patterns carried in an int32 tensor
        sd = torch.from_numpy((np.asarray(seeds, np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32).reshape(B).copy()).to(dev)
    H = torch.zeros((B, 3, 3), dtype=torch.float64, device=dev)
    mask = torch.zeros((B, max(K, 1)), dtype=torch.uint8, device=dev)
    info = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    ctx = Context.get(dev)
    prm = RansacParams(float(threshold), float(confidence), int(max_iters), 1 if refine else 0)
    ctx.check(ctx.lib.kpb_find_homography(ctx.handle, ptr(a), a.shape[2], ptr(b), b.shape[2], B, K, ptr(k_dev), ptr(sc), ptr(sd),
                                          ctypes.c_uint32(int(seed) & 0xFFFFFFFF), ctypes.byref(prm), ptr(H), ptr(mask), ptr(info)))
    return H, mask[:, :K], info


def _seeds(seeds, B, dev):
    if seeds is None:
        return None
    return torch.from_numpy((np.asarray(seeds, np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32).reshape(B).copy()).to(dev)


def estimate_pose(m0, m1, scale, K0, K1, thresh=1.0, conf=0.99999, k_dev=None, seeds=None, seed=0, max_iters=1000, recover_all=False, dist=1e9):
    """tasks/AUC.py:40-64 (`estimate_pose`) for B pairs in two launches: cv2.findEssentialMat(RANSAC) + cv2.recoverPose restated
    (PARITY UNPINNED, see the module docstring).  m0, m1 [B, K, c>=2] matched rows, normalised; scale [B, 4] -> pixels;
    K0, K1 [B, 3, 3] (or [3, 3]) intrinsics.  Returns (Rt [B, 12] float64 = R row-major, t; mask [B, K] uint8 = cv2's `mask`
    after recoverPose; good [B] int32 = recoverPose's count (0: no pose); info [B, 4] = found, RANSAC inliers, hypotheses, 0).
    recover_all: recoverPose is called WITHOUT the RANSAC mask, on every match (tasks/visual_odometer.py:76-77); dist: its
    distanceThresh (1e9 at AUC.py:60, cv2's default 50 at visual_odometer.py:76)."""
    a = m0.detach().to(torch.float32).contiguous()
    b = m1.detach().to(torch.float32).contiguous()
    if a.dim() == 2:
        a, b = a[None], b[None]
    if not a.is_cuda:
        raise RuntimeError("keypoint_bench_amd needs CUDA/HIP tensors; there is no CPU path")
    dev = a.device
    B, K = a.shape[0], a.shape[1]
    sc = torch.as_tensor(scale, dtype=torch.float32).reshape(-1, 4)
    sc = (sc.expand(B, 4) if sc.shape[0] == 1 else sc).contiguous().to(dev)
    k0 = torch.as_tensor(K0).detach().cpu().numpy()
    k1 = torch.as_tensor(K1).detach().cpu().numpy()
    f32 = k0.dtype == np.float32 and k1.dtype == np.float32         # numpy keeps float32 keypoints / float32 K in float32 (AUC.py:44-48)
    if not f32:
        k0, k1 = k0.astype(np.float64), k1.astype(np.float64)
    k0 = np.broadcast_to(k0.reshape(-1, 3, 3), (B, 3, 3))
    k1 = np.broadcast_to(k1.reshape(-1, 3, 3), (B, 3, 3))
    cam = np.stack([k0[:, 0, 2], k0[:, 1, 2], k0[:, 0, 0], k0[:, 1, 1], k1[:, 0, 2], k1[:, 1, 2], k1[:, 0, 0], k1[:, 1, 1]], 1)
    f_mean = np.mean(np.stack([k0[:, 0, 0], k1[:, 1, 1], k0[:, 0, 0], k1[:, 1, 1]], 1), axis=1)              # AUC.py:44 (as written there), in K's dtype
    thr = (thresh / f_mean).astype(np.float64)
    cam_d = torch.from_numpy(np.ascontiguousarray(cam, np.float64)).to(dev)
    thr_d = torch.from_numpy(np.ascontiguousarray(thr)).to(dev)
    E = torch.zeros((B, 9), dtype=torch.float64, device=dev)
    mask = torch.zeros((B, max(K, 1)), dtype=torch.uint8, device=dev)
    mask2 = torch.zeros((B, max(K, 1)), dtype=torch.uint8, device=dev)
    info = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    pts = torch.empty((B, max(K, 1), 4), dtype=torch.float64, device=dev)
    rt = torch.zeros((B, 12), dtype=torch.float64, device=dev)
    good = torch.zeros((B,), dtype=torch.int32, device=dev)
    ctx = Context.get(dev)
    ctx.check(ctx.lib.kpb_find_essential(ctx.handle, ptr(a), a.shape[2], ptr(b), b.shape[2], B, K, ptr(k_dev), ptr(sc), ptr(cam_d), 1 if f32 else 0, ptr(thr_d),
                                         ptr(_seeds(seeds, B, dev)), ctypes.c_uint32(int(seed) & 0xFFFFFFFF), float(conf), int(max_iters), ptr(E),
                                         ptr(mask), ptr(info), ptr(pts)))
    if recover_all:
        mask.fill_(1)
    ctx.check(ctx.lib.kpb_recover_pose(ctx.handle, ptr(E), ptr(pts), ptr(mask), B, K, ptr(k_dev), ptr(info), float(dist), ptr(rt), ptr(mask2), ptr(good)))
    return rt, mask2[:, :K], good, info


def find_fundamental(m0, m1, scale, k_dev=None, seeds=None, seed=0, threshold=3.0, confidence=0.99, max_iters=1000):
    """cv2.findFundamentalMat(.., cv2.FM_RANSAC) for B pairs in one launch (PARITY UNPINNED, see the module docstring).
    Arguments as find_homography.  Returns (F [B, 3, 3] float64 with F[2,2] = 1, mask [B, K] uint8, info [B, 4] int32 =
    found, inliers, hypotheses, 0) on the device; found = 0 below 8 matches."""
    a = m0.detach().to(torch.float32).contiguous()
    b = m1.detach().to(torch.float32).contiguous()
    if a.dim() == 2:
        a, b = a[None], b[None]
    if not a.is_cuda:
        raise RuntimeError("keypoint_bench_amd needs CUDA/HIP tensors; there is no CPU path")
    dev = a.device
    B, K = a.shape[0], a.shape[1]
    sc = torch.as_tensor(scale, dtype=torch.float32).to(dev).reshape(-1, 4)
    sc = (sc.expand(B, 4) if sc.shape[0] == 1 else sc).contiguous()
    F = torch.zeros((B, 3, 3), dtype=torch.float64, device=dev)
    mask = torch.zeros((B, max(K, 1)), dtype=torch.uint8, device=dev)
    info = torch.zeros((B, 4), dtype=torch.int32, device=dev)
    ctx = Context.get(dev)
    prm = RansacParams(float(threshold), float(confidence), int(max_iters), 0)
    ctx.check(ctx.lib.kpb_find_fundamental(ctx.handle, ptr(a), a.shape[2], ptr(b), b.shape[2], B, K, ptr(k_dev), ptr(sc), ptr(_seeds(seeds, B, dev)),
                                           ctypes.c_uint32(int(seed) & 0xFFFFFFFF), ctypes.byref(prm), ptr(F), ptr(mask), ptr(info)))
    return F, mask[:, :K], info


def fundamental_estimate(pts0, pts1, seed=0):
    """Drop-in for utils/mvg.py:4-19: pts0, pts1 [n, 2] pixel coordinates (device tensors).  Returns (F 3x3 numpy or None,
    inlier pts0, inlier pts1) as numpy arrays, like the reference; fewer than 8 points: (None, all points)."""
    p0 = pts0.detach().to(torch.float32)
    p1 = pts1.detach().to(torch.float32)
    if p0.shape[0] < 8:
        print("\n too few points to estimate fundamental matrix \n")
        return None, p0.cpu().numpy(), p1.cpu().numpy()
    F, mask, info = find_fundamental(p0[None], p1[None], [1.0, 1.0, 1.0, 1.0], seed=seed)
    found = int(info[0, 0])
    if not found:                   # cv2 returns (None, None) there and the reference fails on mask.ravel()
        raise AttributeError("'NoneType' object has no attribute 'ravel'")
    keep = mask[0].bool()
    return F[0].cpu().numpy(), p0[keep].cpu().numpy(), p1[keep].cpu().numpy()
