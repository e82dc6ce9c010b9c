"""Empty and degenerate inputs through every drop-in: the reference returns empty tensors of the right shape (or, for
val_key_points, its zero dict); nothing may crash, hang or touch memory it was not given."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
BF = dict(metric="euclidean", max_distance=5, cross_check=True)


def test_no_keypoints_anywhere():
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import brute_force_matcher, OpticalFlow
    from keypoint_bench_amd.utils.projection import warp
    from keypoint_bench_amd.tasks.repeatability import val_key_points
    zero = torch.zeros((1, 1, 96, 128), device=DEV)
    k = detection(zero, EP)
    assert k.shape == (0, 3)
    desc = torch.randn((1, 64, 96, 128), device=DEV)
    m0, m1 = brute_force_matcher(k, k, desc, desc, BF)
    assert m0.shape[0] == 0 and m1.shape[0] == 0
    some = torch.rand((5, 3), device=DEV)
    m0, m1 = brute_force_matcher(some, k, desc, desc, BF)
    assert m0.shape[0] == 0 and m1.shape[0] == 0
    w = dict(mode="homo", homography_matrix=torch.eye(3, device=DEV), width=128, height=96)
    a, b, ids, out = warp(k, w)
    assert a.shape == (0, 2) and b.shape == (0, 2) and ids.numel() == 0 and out.numel() == 0
    r = val_key_points(k, some, w, w)
    assert r == {"num_feat": 0, "repeatability": 0, "mean_error": 0, "errors": None}
    far = dict(mode="homo", homography_matrix=torch.tensor([[1.0, 0, 1e4], [0, 1, 0], [0, 0, 1]], device=DEV), width=128, height=96)
    r = val_key_points(some, some, far, far)            # everything warps out of the image
    assert r["num_feat"] == 0 and r["errors"] is None
    img = torch.rand((1, 3, 96, 128), device=DEV)
    p, e = OpticalFlow()(img, img, k[:, :2], k[:, :2])
    assert p.shape == (1, 0, 2) and e.shape == (1, 0)


def test_single_keypoint_and_tiny_sets():
    from keypoint_bench_amd.utils.matcher import brute_force_matcher
    from keypoint_bench_amd.tasks.repeatability import val_key_points
    desc = torch.randn((1, 64, 96, 128), device=DEV)
    one = torch.tensor([[0.5, 0.5, 0.9]], device=DEV)
    m0, m1 = brute_force_matcher(one, one, desc, desc, dict(BF, max_distance=1e9))
    assert m0.shape == (1, 3) and torch.equal(m0, one) and torch.equal(m1, one)
    w = dict(mode="homo", homography_matrix=torch.eye(3, device=DEV), width=128, height=96)
    r = val_key_points(one, one, w, w)                  # the masked diagonal (repeatability.py:72-73) leaves nothing to match
    assert r["num_feat"] == 1 and float(r["repeatability"]) == 0.0


def test_constant_and_saturated_maps():
    from keypoint_bench_amd.utils.extracter import detection
    import oracle
    for m in (np.full((64, 96), 0.5, np.float32), np.ones((64, 96), np.float32), np.full((64, 96), 1e-30, np.float32)):
        got = detection(torch.from_numpy(m)[None, None].to(DEV), dict(EP, nms_dist=4, border_dist=0)).cpu().numpy()
        exp, _ = oracle.detection(m, dict(EP, nms_dist=4, border_dist=0))
        np.testing.assert_array_equal(got, exp)


def test_host_counts_of_detection_and_matching():
    """kpb_detect_counts / kpb_match_counts (r04): the counts the kernels leave in pinned host memory equal the device counts, for one
    image and for a batch whose size grows (the mirror is re-allocated); asking with the wrong batch is an error, not stale numbers."""
    import ctypes
    from keypoint_bench_amd import synthetic
    from keypoint_bench_amd._lib import Context, KpbError, MatchParams, ptr
    from keypoint_bench_amd.utils.extracter import detection_batch
    ctx = Context.get(torch.device(DEV))
    for B in (1, 3, 70):
        maps = torch.from_numpy(np.stack([synthetic.score_uniform(900 + i, 64, 96) for i in range(B)]))[:, None].to(DEV)
        p = dict(nms_dist=3, threshold=0.5 if B == 3 else 0.0, border_dist=4, top_k=200, min_score=0.0)
        _, _, n = detection_batch(maps, p)
        host = (ctypes.c_int32 * B)()
        ctx.check(ctx.lib.kpb_detect_counts(ctx.handle, host, B))
        assert list(host) == n.cpu().tolist()
        with pytest.raises(KpbError):
            ctx.check(ctx.lib.kpb_detect_counts(ctx.handle, host, B + 1))
    for B in (1, 9):
        a = torch.randn((B, 40, 64), device=DEV)
        b = a + 0.01 * torch.randn((B, 40, 64), device=DEV)
        nn = torch.tensor([40 - (i % 3) for i in range(B)], dtype=torch.int32, device=DEV)
        pairs = torch.empty((B, 40, 2), dtype=torch.int32, device=DEV)
        k = torch.empty((B,), dtype=torch.int32, device=DEV)
        prm = MatchParams(5.0, 1)
        ctx.check(ctx.lib.kpb_match(ctx.handle, ptr(a), ptr(b), B, 64, 40, 40, ptr(nn), ptr(nn), ctypes.byref(prm), ptr(pairs), ptr(None), ptr(k)))
        host = (ctypes.c_int32 * B)()
        ctx.check(ctx.lib.kpb_match_counts(ctx.handle, host, B))
        assert list(host) == k.cpu().tolist() == nn.cpu().tolist()
        with pytest.raises(KpbError):
            ctx.check(ctx.lib.kpb_match_counts(ctx.handle, host, B + 2))
