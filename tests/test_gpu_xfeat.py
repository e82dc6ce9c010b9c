"""GPU parity of N3 (XFeat, csrc/convnet.hip through the C ABI) against the reference's outputs on seeded random
weights (xfeat.pt is absent from the reference tree) and against the torch-fp32 oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from keypoint_bench_amd import synthetic, weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# the heat-map is a softmax over logits of a 4-layer head on the instance-normalised image: rtol on probabilities;
# features are unit vectors: north_star's 1e-4 absolute.
RTOL_HEAT, ATOL_HEAT, ATOL_DESC = 2e-3, 1e-7, 1e-4


def _net(seed=9):
    from keypoint_bench_amd.models.XFeat import xfeat_random
    return xfeat_random(seed).eval()


def test_xfeat_small_against_reference_golden():
    g = load_golden("nets.npz")
    v0, _ = synthetic.image_pair(0, 64, 96)
    heat, feats = _net()(torch.from_numpy(v0)[None].to(DEV))
    assert heat.shape == (1, 1, 64, 96) and feats.shape == (1, 64, 8, 12)
    np.testing.assert_allclose(heat[0, 0].cpu().numpy(), g["xf.small.heat"], rtol=RTOL_HEAT, atol=ATOL_HEAT)
    np.testing.assert_allclose(feats[0].cpu().numpy(), g["xf.small.desc"], rtol=0, atol=ATOL_DESC)


def test_xfeat_full_size_batch_and_oracle():
    from oracle import xfeat_ref
    g = load_golden("nets.npz")
    v0, v1 = synthetic.image_pair(0)
    net = _net()
    heat, feats = net(torch.from_numpy(np.stack([v0, v1])).to(DEV))
    np.testing.assert_allclose(heat[0, 0].cpu().numpy(), g["xf.full.heat"], rtol=RTOL_HEAT, atol=ATOL_HEAT)
    np.testing.assert_allclose(feats[0, :, ::4, ::4].cpu().numpy(), g["xf.full.desc"], rtol=0, atol=ATOL_DESC)
    h1, f1 = net(torch.from_numpy(v1)[None].to(DEV))
    np.testing.assert_allclose(h1[0].cpu().numpy(), heat[1].cpu().numpy(), rtol=1e-5, atol=0)   # per-image stats: batch-independent
    t = {k: torch.from_numpy(v) for k, v in weights.fold_xfeat(weights.random_xfeat_state_dict(9)).items()}
    with torch.no_grad():
        ho, fo = xfeat_ref.xfeat_forward(torch.from_numpy(v1)[None], t)
    np.testing.assert_allclose(heat[1].cpu().numpy(), ho[0].numpy(), rtol=RTOL_HEAT, atol=ATOL_HEAT)
    np.testing.assert_allclose(feats[1].cpu().numpy(), fo[0].numpy(), rtol=0, atol=ATOL_DESC)


def test_xfeat_feeds_detection_and_matcher():
    import oracle
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import brute_force_matcher, sample_descriptors
    v0, v1 = synthetic.image_pair(6, 128, 160)
    net = _net(3)
    h0, d0 = net(torch.from_numpy(v0)[None].to(DEV))
    h1, d1 = net(torch.from_numpy(v1)[None].to(DEV))
    ep = dict(nms_dist=4, threshold=0.0, border_dist=8, top_k=300, min_score=0.0)
    k0, k1 = detection(h0, ep), detection(h1, ep)
    ok0, _ = oracle.detection(h0[0, 0].cpu().numpy(), ep)
    np.testing.assert_array_equal(k0.cpu().numpy().view(np.uint32), ok0.view(np.uint32))
    np.testing.assert_array_equal(sample_descriptors(k0, d0).cpu().numpy(), oracle.sample(d0[0].cpu().numpy(), ok0))
    m0, m1 = brute_force_matcher(k0, k1, d0, d1, {"metric": "euclidean", "max_distance": 5, "cross_check": True})
    assert m0.shape == m1.shape and m0.shape[1] == 3
