"""GPU parity of the depth-based covisibility warp (kpb_warp_se3 through the C ABI) against the reference's fixtures and the
oracle.  libkpb always evaluates the fused (BLAS) form of the three small matmuls: bit-exact against the oracle's fused
form everywhere and against the reference wherever it ran its BLAS (>= 45 points with depth in image 0); below that the
reference's unfused loop differs by an ulp of the pixel coordinate (5e-7 normalised) and the id lists still agree on these
fixtures."""
import numpy as np
import pytest
import torch

import oracle
from test_oracle_se3 import G, se3_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("c", range(int(G["n_cases"])))
def test_warp_se3(c):
    from keypoint_bench_amd.utils.projection import warp
    s = se3_case(c)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    prm = dict(mode="se3", pose01=t(s["pose"]), bbox0=t(s["bbox0"]), bbox1=t(s["bbox1"]), depth0=t(s["depth0"]), depth1=t(s["depth1"]),
               intrinsics0=t(s["k0"]), intrinsics1=t(s["k1"]))
    a, b, ids, out = warp(t(s["kps"]), prm)
    assert ids.dtype == torch.int64 and out.dtype == torch.int64 and a.shape == b.shape == (len(ids), 2)
    ea, eb, eids, eout = oracle.warp_se3(s["kps"], s["depth0"], s["depth1"], s["kinv0"], s["k1"], s["pose"], s["bbox0"], s["bbox1"], fused=1)
    for got, want in zip((a, b, ids, out), (ea, eb, eids, eout)):
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    wa, wb, wids, wout = s["want"]
    np.testing.assert_array_equal(ids.cpu().numpy(), wids)
    np.testing.assert_array_equal(out.cpu().numpy(), wout)
    np.testing.assert_array_equal(a.cpu().numpy(), wa)
    if len(wids) and (np.asarray(s["kps"]).shape[0] >= 200):        # the reference's matmuls went through its BLAS
        np.testing.assert_array_equal(b.cpu().numpy(), wb)
    else:
        np.testing.assert_allclose(b.cpu().numpy(), wb, rtol=0, atol=5e-7)     # an ulp of a ~300 px coordinate, normalised


def test_warp_se3_feeds_val_key_points():
    """The repeatability core runs on se3 warps as it does on homographies."""
    from keypoint_bench_amd.utils.projection import warp
    from keypoint_bench_amd.tasks.repeatability import gt_mutual
    s = se3_case(0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    prm = dict(mode="se3", pose01=t(s["pose"]), bbox0=t(s["bbox0"]), bbox1=t(s["bbox1"]), depth0=t(s["depth0"]), depth1=t(s["depth1"]),
               intrinsics0=t(s["k0"]), intrinsics1=t(s["k1"]))
    a, b, ids, out = warp(t(s["kps"]), prm)
    pairs, dist, errors, gt = gt_mutual(a, b, a, b, 640.0, 640.0)
    assert errors.shape[0] == len(ids) and len(set(ids.tolist()) & set(out.tolist())) == 0
