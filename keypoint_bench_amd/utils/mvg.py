"""Robust two-view geometry on the device: what the reference gets from OpenCV --
`cv2.findHomography(pts0, pts1, cv2.RANSAC)` (tasks/MHA.py:45-47) -- computed by csrc/geometry.hip through libkpb.so.

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
