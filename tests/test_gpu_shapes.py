"""MegaDepth-like shapes.  datasets/megadepth.py:247-248 hands test images over at their native
size and model_interface.py:192-204 crops them to multiples of 32: 800 x 1216 and 1216 x 1600 stand for what BASELINE configs[2]
and configs[4] really run on -- four to six times the pixels of the 480 x 640 every other test uses, other tile counts in every
kernel's grid, other strip geometry in the ALIKE head.  Each net against its torch-fp32 oracle restatement (tolerances of the
per-net test files), then the whole extract -> NMS / top-K -> sampling -> match pipeline against the oracle chain on the
GPU's own maps, bit for bit."""
import numpy as np
import pytest
import torch

import oracle
from keypoint_bench_amd import synthetic, weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(800, 1216), (1216, 1600)]
EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)       # config/config_MHA.yaml:68-73
BF = dict(metric="euclidean", max_distance=5, cross_check=True)


def _image(seed, H, W):
    return synthetic.image_pair(seed, H, W)[0]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dense", [True, False])
def test_alike_at_megadepth_shapes(shape, dense):
    from oracle import alike_ref
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.matcher import sample_descriptors
    H, W = shape
    img = _image(41, H, W)
    t = {k: torch.from_numpy(v) for k, v in weights.load_alike_t().items()}
    with torch.no_grad():
        so, do = alike_ref.alnet_forward(torch.from_numpy(img)[None], t)
    score, desc = alike_t(dense_descriptors=dense).eval()(torch.from_numpy(img)[None].to(DEV))
    np.testing.assert_allclose(score[0, 0].cpu().numpy(), so[0, 0].numpy(), rtol=0, atol=1e-5)
    if dense:
        np.testing.assert_allclose(desc[0].cpu().numpy(), do[0].numpy(), rtol=0, atol=1e-4)
    else:
        k, _ = oracle.detection(so[0, 0].numpy(), EP)
        got = sample_descriptors(torch.from_numpy(k).to(DEV), desc).cpu().numpy()
        np.testing.assert_allclose(got, oracle.sample(do[0].numpy(), k), rtol=0, atol=1e-4)


@pytest.mark.parametrize("shape", SHAPES)
def test_superpoint_at_megadepth_shapes(shape):
    from oracle import superpoint_ref
    from keypoint_bench_amd.models.SuperPoint import superpoint_random
    H, W = shape
    img = _image(42, H, W)
    tw = weights.random_superpoint(7)
    with torch.no_grad():
        ho, do = superpoint_ref.superpoint_forward(torch.from_numpy(img)[None], {k: torch.from_numpy(v) for k, v in tw.items()})
    heat, desc = superpoint_random(7).eval()(torch.from_numpy(img)[None].to(DEV))
    assert heat.shape == (1, 1, H, W) and desc.shape == (1, 256, H // 8, W // 8)
    np.testing.assert_allclose(heat.cpu().numpy(), ho.numpy(), rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(desc.cpu().numpy(), do.numpy(), rtol=0, atol=1e-4)


@pytest.mark.parametrize("shape", SHAPES)
def test_xfeat_at_megadepth_shapes(shape):
    from oracle import xfeat_ref
    from keypoint_bench_amd.models.XFeat import xfeat_random
    H, W = shape
    img = _image(43, H, W)
    t = {k: torch.from_numpy(v) for k, v in weights.fold_xfeat(weights.random_xfeat_state_dict(9)).items()}
    with torch.no_grad():
        ho, fo = xfeat_ref.xfeat_forward(torch.from_numpy(img)[None], t)
    heat, feats = xfeat_random(9).eval()(torch.from_numpy(img)[None].to(DEV))
    np.testing.assert_allclose(heat.cpu().numpy(), ho.numpy(), rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(feats.cpu().numpy(), fo.numpy(), rtol=0, atol=1e-4)


@pytest.mark.parametrize("shape", SHAPES)
def test_disk_at_megadepth_shapes(shape):
    from oracle import disk_ref
    from keypoint_bench_amd.models.disk import disk_random
    H, W = shape
    img = _image(44, H, W)
    t = {k: torch.from_numpy(v) for k, v in weights.tensors_disk(weights.random_disk_state_dict(5)).items()}
    with torch.no_grad():
        so, do = disk_ref.disk_forward(torch.from_numpy(img)[None], t)
    score, desc = disk_random(5).eval()(torch.from_numpy(img)[None].to(DEV))
    np.testing.assert_allclose(score.cpu().numpy(), so.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(desc.cpu().numpy(), do.numpy(), rtol=0, atol=1e-4)


@pytest.mark.parametrize("name", ["alike", "superpoint", "xfeat", "disk"])
def test_pipeline_at_a_megadepth_shape(name):
    """PairPipeline (one launch wave: net, NMS / top-K, sampling, match, gather) at 800 x 1216, top_k 1000: every integer stage
    equals the oracle evaluated on the GPU's own score / descriptor maps, bit for bit (as tests/test_gpu_configs.py at 480 x 640)."""
    from keypoint_bench_amd.pipeline import PairPipeline
    H, W = 800, 1216
    if name == "alike":
        from keypoint_bench_amd.models.ALike import alike_t
        net = alike_t().eval()
    elif name == "superpoint":
        from keypoint_bench_amd.models.SuperPoint import superpoint_random
        net = superpoint_random(7).eval()
    elif name == "xfeat":
        from keypoint_bench_amd.models.XFeat import xfeat_random
        net = xfeat_random(9).eval()
    else:
        from keypoint_bench_amd.models.disk import disk_random
        net = disk_random(5).eval()
    v0, v1 = synthetic.image_pair(77, H, W)
    images = torch.from_numpy(np.stack([v0, v1])).to(DEV).contiguous()
    pipe = PairPipeline(net, EP, BF, 1, H, W, device=DEV).run(images)
    n = pipe.n.cpu().numpy()
    assert (n == 1000).all(), n
    kps, sdesc = [], []
    for i in (0, 1):
        k, idx = oracle.detection(pipe.score[i, 0].cpu().numpy(), EP)
        assert np.array_equal(pipe.kps[i, : n[i]].cpu().numpy().view(np.uint32), k.view(np.uint32)), (name, i)
        assert np.array_equal(pipe.idx[i, : n[i]].cpu().numpy(), idx)
        d = oracle.sample(pipe.desc[i].permute(2, 0, 1).cpu().numpy(), k)
        assert np.array_equal(pipe.sdesc[i, : n[i]].cpu().numpy(), d), (name, i)
        kps.append(k); sdesc.append(d)
    pairs, dist = oracle.match(sdesc[0], sdesc[1], BF["max_distance"], BF["cross_check"])
    kk = int(pipe.k[0])
    assert kk == len(pairs), (name, kk, len(pairs))
    assert np.array_equal(pipe.pairs[0, :kk].cpu().numpy(), pairs) and np.array_equal(pipe.dist[0, :kk].cpu().numpy(), dist)
    assert np.array_equal(pipe.m0[0, :kk].cpu().numpy(), kps[0][pairs[:, 0]]) and np.array_equal(pipe.m1[0, :kk].cpu().numpy(), kps[1][pairs[:, 1]])
