"""Synthetic inputs of BASELINE.md section 3 (no datasets travel to the GPU box).

Everything is built from elementwise IEEE operations on numpy ``default_rng`` streams so that the
same seed gives the same bytes here and on the GPU box (fixtures store a checksum to prove it).
"""
import hashlib

import numpy as np


def _box_blur(a, k):
    """k x k box filter, edge-replicated, accumulated in float64 in a fixed order."""
    r = k // 2
    p = np.pad(a.astype(np.float64), ((r, r), (r, r)), mode="edge")
    H, W = a.shape
    acc = np.zeros((H, W), np.float64)
    for dy in range(k):
        for dx in range(k):
            acc = acc + p[dy:dy + H, dx:dx + W]
    return acc / float(k * k)


def image_pair(i, H=480, W=640):
    """Pair i: fp32 RGB [3,H,W] in [0,1] x2.  Canvas (H+20)x(W+20) uniform noise, 5x5 box blur per
    channel, min-max normalised; view0 = canvas[10:10+H, 10:10+W]; view1 = canvas[12:, 13:] + N(0, 0.02^2)."""
    rng = np.random.default_rng(1234 + i)
    canvas = rng.random((3, H + 20, W + 20), dtype=np.float32)
    canvas = np.stack([_box_blur(c, 5) for c in canvas])
    lo, hi = canvas.min(), canvas.max()
    canvas = ((canvas - lo) / (hi - lo)).astype(np.float32)
    v0 = canvas[:, 10:10 + H, 10:10 + W].copy()
    noise = rng.normal(0.0, 0.02, size=(3, H, W))
    v1 = np.clip(canvas[:, 12:12 + H, 13:13 + W].astype(np.float64) + noise, 0.0, 1.0).astype(np.float32)
    return v0, v1


def score_uniform(seed, H, W):
    """Detector microbench map (i): uniform [0,1) float32."""
    return np.random.default_rng(seed).random((H, W), dtype=np.float32)


def score_smooth(seed, H, W):
    """Detector microbench map (ii): sigmoid(36 * twice-9x9-box-blurred N(0,1))."""
    z = np.random.default_rng(seed).normal(size=(H, W))
    z = _box_blur(_box_blur(z, 9), 9) * 36.0
    return (1.0 / (1.0 + np.exp(-z))).astype(np.float32)


def checksum(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
