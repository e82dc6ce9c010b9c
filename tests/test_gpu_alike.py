"""GPU parity of N1 (csrc/alike.hip through the C ABI): ALIKE-t forward against the reference's golden
outputs and the torch-fp32 oracle restatement."""
import numpy as np
import pytest
import torch

import oracle
from oracle import alike_ref
from conftest import load_golden, assert_kps_equal
from keypoint_bench_amd import synthetic, weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# north_star: descriptors within 1e-4 fp32.  Score maps: 1e-5 (observed ~4e-6 against the reference,
# from BN folding and a different fp32 summation order than oneDNN).
ATOL_DESC, ATOL_SCORE = 1e-4, 7e-6      # score: measured 4.9e-6 on 256 full-size pairs with the round-to-nearest split (r04); r03's truncating split (8.0e-6) would fail here, as it should


def _oracle_forward(img, intermediates=False):
    t = {k: torch.from_numpy(v) for k, v in weights.load_alike_t().items()}
    with torch.no_grad():
        return alike_ref.alnet_forward(torch.from_numpy(img)[None], t, intermediates)


def test_forward_small_against_reference_golden():
    from keypoint_bench_amd.models.ALike import alike_t
    g = load_golden("alike_t.npz")
    v0, v1 = synthetic.image_pair(0, 64, 96)
    net = alike_t().eval()
    for img, s_key, d_key in ((v0, "small.score0", "small.desc0"), (v1, "small.score1", "small.desc1")):
        score, desc = net(torch.from_numpy(img)[None].to(DEV))
        assert score.shape == (1, 1, 64, 96) and desc.shape == (1, 64, 64, 96)
        np.testing.assert_allclose(score[0, 0].cpu().numpy(), g[s_key], rtol=0, atol=ATOL_SCORE)
        np.testing.assert_allclose(desc[0].cpu().numpy(), g[d_key], rtol=0, atol=ATOL_DESC)


def test_forward_full_size_against_reference_golden_and_oracle():
    from keypoint_bench_amd.models.ALike import alike_t
    g = load_golden("alike_t.npz")
    v0, v1 = synthetic.image_pair(0)
    net = alike_t().eval()
    score, desc = net(torch.from_numpy(np.stack([v0, v1])).to(DEV))     # batch of 2 through one launch wave
    for b, (sk, dk) in enumerate((("full.score0", "full.desc0_sub16"), ("full.score1", "full.desc1_sub16"))):
        np.testing.assert_allclose(score[b, 0].cpu().numpy(), g[sk], rtol=0, atol=ATOL_SCORE)
        np.testing.assert_allclose(desc[b, :, ::16, ::16].cpu().numpy(), g[dk], rtol=0, atol=ATOL_DESC)
    so, do = _oracle_forward(v0)
    np.testing.assert_allclose(score[0, 0].cpu().numpy(), so[0, 0].numpy(), rtol=0, atol=ATOL_SCORE)
    np.testing.assert_allclose(desc[0].cpu().numpy(), do[0].numpy(), rtol=0, atol=ATOL_DESC)


@pytest.mark.parametrize("dense", [True, False])
def test_a_batch_that_fills_the_chip_and_a_single_image_take_different_launch_shapes_to_the_same_maps(dense):
    """From 16 images on, block 4 runs on 16 x 16 tiles with two n-tiles per workgroup, the head walks up to 30 row groups per
    persistent workgroup; a single image (the drop-in path) gets 8-row tiles with one n-tile and 2 groups per workgroup so that its
    few workgroups spread over the chip.  Same arithmetic per output: the maps must agree to the last bit, and with the oracle."""
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.matcher import sample_descriptors
    imgs = np.stack([synthetic.image_pair(60 + i, 96, 160)[i & 1] for i in range(18)])
    net = alike_t(dense_descriptors=dense).eval()
    sb, db = net(torch.from_numpy(imgs).to(DEV))
    pts = torch.rand((64, 2), device=DEV)
    for i in (0, 7, 17):
        s1, d1 = net(torch.from_numpy(imgs[i:i + 1]).to(DEV))
        assert torch.equal(s1[0], sb[i])
        if dense:
            assert torch.equal(d1[0], db[i])
        so, do = _oracle_forward(imgs[i])
        np.testing.assert_allclose(sb[i, 0].cpu().numpy(), so[0, 0].numpy(), rtol=0, atol=ATOL_SCORE)
        if dense:
            np.testing.assert_allclose(db[i].cpu().numpy(), do[0].numpy(), rtol=0, atol=ATOL_DESC)


@pytest.mark.parametrize("shape", [(512, 512), (480, 608), (96, 32), (32, 160), (224, 736)])
@pytest.mark.parametrize("dense", [True, False])
def test_forward_other_shapes_against_oracle(shape, dense):
    """HPatches runs at 512 x 512 (config_MHA.yaml:16); widths that are not multiples of 128 end a row with a short head
    segment; 32-pixel extents are the smallest the net accepts (one pixel at 1/32 resolution)."""
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.matcher import sample_descriptors
    H, W = shape
    img = synthetic.image_pair(7, max(H, 64), max(W, 64))[0][:, :H, :W].copy()
    score, desc = alike_t(dense_descriptors=dense).eval()(torch.from_numpy(img)[None].to(DEV))
    so, do = _oracle_forward(img)
    np.testing.assert_allclose(score[0, 0].cpu().numpy(), so[0, 0].numpy(), rtol=0, atol=ATOL_SCORE)
    if dense:
        np.testing.assert_allclose(desc[0].cpu().numpy(), do[0].numpy(), rtol=0, atol=ATOL_DESC)
    else:
        rng = np.random.default_rng(H + W)
        pts = torch.from_numpy(rng.random((200, 2)).astype(np.float32)).to(DEV)
        want = sample_descriptors(pts, do.to(DEV)).cpu().numpy()
        np.testing.assert_allclose(sample_descriptors(pts, desc).cpu().numpy(), want, rtol=0, atol=ATOL_DESC)


def test_state_dict_loading_equals_packed_blob():
    """load_state_dict on an ALNet-shaped state dict (BN un-folded) gives the same network as the blob."""
    from keypoint_bench_amd.models.ALike import ALNet, alike_t
    t = weights.load_alike_t()
    # rebuild an un-folded state dict whose BN is the identity: folding it must reproduce the blob
    sd = {}
    for blk, q in (("block1", "b1"), ("block2", "b2"), ("block3", "b3"), ("block4", "b4")):
        for c in ("1", "2"):
            w, b = t[q + "c" + c + ".w"], t[q + "c" + c + ".b"]
            sd[blk + ".conv" + c + ".weight"] = torch.from_numpy(w)
            n = w.shape[0]
            sd[blk + ".bn" + c + ".weight"] = torch.full((n,), float(np.sqrt(1.0 + 1e-5)))
            sd[blk + ".bn" + c + ".bias"] = torch.from_numpy(b)
            sd[blk + ".bn" + c + ".running_mean"] = torch.zeros(n)
            sd[blk + ".bn" + c + ".running_var"] = torch.ones(n)
        if blk != "block1":
            sd[blk + ".downsample.weight"] = torch.from_numpy(t[q + "ds.w"])[:, :, None, None]
            sd[blk + ".downsample.bias"] = torch.from_numpy(t[q + "ds.b"])
    for i in (1, 2, 3, 4):
        sd["conv%d.weight" % i] = torch.from_numpy(t["agg%d.w" % i])[:, :, None, None]
    sd["convhead2.weight"] = torch.from_numpy(t["head.w"])[:, :, None, None]
    a = ALNet({"c1": 8, "c2": 16, "c3": 32, "c4": 64, "dim": 64})
    a.load_state_dict(sd)
    img = torch.from_numpy(synthetic.image_pair(3, 64, 96)[0])[None].to(DEV)
    s1, d1 = a.eval()(img)
    s2, d2 = alike_t().eval()(img)
    np.testing.assert_allclose(s1.cpu().numpy(), s2.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(d1.cpu().numpy(), d2.cpu().numpy(), rtol=0, atol=2e-5)


def test_lazy_descriptors_equal_dense_sampling():
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import sample_descriptors, brute_force_matcher
    v0, v1 = synthetic.image_pair(5)
    dense, lazy = alike_t().eval(), alike_t(dense_descriptors=False).eval()
    p = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
    img = torch.from_numpy(v0)[None].to(DEV)
    s_d, d_d = dense(img)
    s_l, d_l = lazy(img)
    # the score-only path evaluates the head's score row in its linear form (different summation order)
    np.testing.assert_allclose(s_l.cpu().numpy(), s_d.cpu().numpy(), rtol=0, atol=2e-6)
    assert tuple(d_l.shape) == tuple(d_d.shape)
    kps = detection(s_d, p)
    want = sample_descriptors(kps, d_d).cpu().numpy()
    got = sample_descriptors(kps, d_l).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)   # linear re-association only
    # corner / outside points exercise the zero-padding taps
    pts = torch.tensor([[0.0, 0.0], [1.0, 1.0], [1.0, 0.0], [0.5, 0.5], [0.9999, 0.0001]], device=DEV)
    np.testing.assert_allclose(sample_descriptors(pts, d_l).cpu().numpy(), sample_descriptors(pts, d_d).cpu().numpy(), rtol=0, atol=2e-5)


def test_lazy_descriptors_survive_the_reference_call_pattern():
    """model(img0), model(img1), then the matcher (model_interface.py:205-212 -> tasks/MHA.py:38-39): both handles must
    still be alive and give what the dense maps give; a third forward retires the oldest one, loudly."""
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import brute_force_matcher
    v0, v1 = synthetic.image_pair(6)
    p = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
    bf = dict(metric="euclidean", max_distance=5, cross_check=True)
    i0, i1 = torch.from_numpy(v0)[None].to(DEV), torch.from_numpy(v1)[None].to(DEV)
    dense, lazy = alike_t().eval(), alike_t(dense_descriptors=False).eval()
    sd0, dd0 = dense(i0); sd1, dd1 = dense(i1)
    sl0, dl0 = lazy(i0); sl1, dl1 = lazy(i1)
    k0, k1 = detection(sd0, p), detection(sd1, p)
    want0, want1 = brute_force_matcher(k0, k1, dd0, dd1, bf)
    got0, got1 = brute_force_matcher(k0, k1, dl0, dl1, bf)
    assert torch.equal(got0, want0) and torch.equal(got1, want1) and got0.shape[0] > 500
    lazy(i0)
    with pytest.raises(RuntimeError, match="later forwards"):
        dl0.sample(k0)
    dl1.sample(k1)      # the newer of the two is still alive


def test_end_to_end_pair_against_reference_golden():
    """image pair -> ALIKE-t -> detection -> brute-force match, against what the reference produced from the same pixels: the SAME
    1000 + 1000 keypoint pixels and the SAME match pixel pairs (r04: exact; r03 accepted 99 % / 97 %).  Rows are compared as sets: a
    score-map ulp may permute near-equal rows of the score-descending output (SURVEY 'Top-K ties'); 24 more reference pairs, with
    viewpoint homographies, are in tests/test_gpu_metric_from_pixels.py."""
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import brute_force_matcher
    g = load_golden("alike_t.npz")
    v0, v1 = synthetic.image_pair(0)
    p = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
    bf = {"metric": "euclidean", "max_distance": 5, "cross_check": True}
    xy = lambda k: list(map(tuple, np.round(np.asarray(k)[:, :2] * np.array([640, 480]) - 0.5).astype(int).tolist()))
    want_pairs = set(zip(xy(g["full.m0"]), xy(g["full.m1"])))
    for dense in (True, False):
        net = alike_t(dense_descriptors=dense).eval()
        s0, d0 = net(torch.from_numpy(v0)[None].to(DEV))
        s1, d1 = net(torch.from_numpy(v1)[None].to(DEV))
        k0, k1 = detection(s0, p), detection(s1, p)
        for got, want in ((k0, g["full.kps0"]), (k1, g["full.kps1"])):
            a, b = set(xy(got.cpu().numpy())), set(xy(want))
            assert a == b, "keypoint sets differ from the reference's: %d of %d" % (len(a ^ b) // 2, len(b))
        m0, m1 = brute_force_matcher(k0, k1, d0, d1, bf)
        got_pairs = set(zip(xy(m0.cpu().numpy()), xy(m1.cpu().numpy())))
        assert got_pairs == want_pairs, "match sets differ from the reference's: %d of %d" % (len(got_pairs ^ want_pairs), len(want_pairs))
