"""The batched pair pipeline must equal the single-pair drop-in path row for row, and the runner's
per-pair rows must equal both (GPU, through the C ABI)."""
import numpy as np
import pytest
import torch

from keypoint_bench_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
BF = dict(metric="euclidean", max_distance=5, cross_check=True)


def _pairs(n, H=480, W=640):
    v = [synthetic.image_pair(20 + i, H, W) for i in range(n)]
    return np.stack([a for a, _ in v]), np.stack([b for _, b in v])


@pytest.mark.parametrize("dense", [True, False])
def test_pipeline_equals_single_pair_path(dense):
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.pipeline import PairPipeline
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import sample_descriptors, match_descriptors
    B = 3
    i0, i1 = _pairs(B)
    net = alike_t(dense_descriptors=dense).eval()
    pipe = PairPipeline(net, EP, BF, B, 480, 640, device=DEV)
    pipe.run(torch.from_numpy(np.concatenate([i0, i1])).to(DEV))
    single = alike_t(dense_descriptors=dense).eval()
    for b in range(B):
        got = pipe.pair(b)
        s0, d0 = single(torch.from_numpy(i0[b])[None].to(DEV))
        k0 = detection(s0, EP)
        f0 = sample_descriptors(k0, d0)
        s1, d1 = single(torch.from_numpy(i1[b])[None].to(DEV))
        k1 = detection(s1, EP)
        f1 = sample_descriptors(k1, d1)
        np.testing.assert_array_equal(got["kps0"], k0.cpu().numpy())
        np.testing.assert_array_equal(got["kps1"], k1.cpu().numpy())
        pairs, dist = match_descriptors(f0, f1, max_distance=5, cross_check=True, return_distance=True)
        np.testing.assert_array_equal(got["pairs"], pairs.cpu().numpy())
        np.testing.assert_array_equal(got["dist"], dist.cpu().numpy())
        np.testing.assert_array_equal(got["m0"], k0.cpu().numpy()[got["pairs"][:, 0]])
        np.testing.assert_array_equal(got["m1"], k1.cpu().numpy()[got["pairs"][:, 1]])
        assert got["pairs"].shape[0] > 500


def test_runner_rows_and_install_shim():
    from keypoint_bench_amd import runner
    from keypoint_bench_amd.pipeline import PairPipeline
    B = 2
    i0, i1 = _pairs(B, 500, 660)     # not multiples of 32: the runner must crop like model_interface.py:192-204
    params = {"model_type": "Alike", "task_type": "match_stats", "Alike_params": dict(c1=8, c2=16, c3=32, c4=64, dim=64),
              "extractor_params": EP, "matcher_params": {"type": "brute_force", "brute_force_params": BF}}
    ds = [{"image0": torch.from_numpy(i0[i])[None], "image1": torch.from_numpy(i1[i])[None], "dataset": ["image_pair"]}
          for i in range(B)]
    r = runner.PairRunner(params, device=DEV)
    agg, rows = r.run(ds)
    pipe = PairPipeline(r.model, EP, BF, B, 480, 640, device=DEV)
    crop = lambda a: np.ascontiguousarray(a[:, :, :480, :640])
    pipe.run(torch.from_numpy(np.concatenate([crop(i0), crop(i1)])).to(DEV))
    want = np.stack([pipe.n[:B].cpu().numpy(), pipe.n[B:].cpu().numpy(), pipe.k.cpu().numpy()], 1)
    np.testing.assert_array_equal(rows[:, :3].astype(np.int64), want)
    assert agg["mean_matches"] == pytest.approx(want[:, 2].mean())
    assert runner.install() == []     # no reference checkout on the path: nothing to swap, and no error


def test_placing_the_descriptor_map_changes_where_it_lives_and_nothing_else(monkeypatch):
    """PairPipeline._place_map (r06, profiles/r06_head_modes.txt): the dense map's allocation is chosen by timing the forward into a few candidates.
    At a size where it would not bother (PLACE_MIN_BYTES lowered for the test) the placed pipeline must return exactly the rows of the unplaced one,
    leave a record of its choice, hand the losing candidates back, and place only once."""
    from keypoint_bench_amd import pipeline
    from keypoint_bench_amd.models.ALike import alike_t
    B = 2
    i0, i1 = _pairs(B, 96, 128)
    images = torch.from_numpy(np.concatenate([i0, i1])).to(DEV)
    ep = dict(EP, top_k=200)
    ref = pipeline.PairPipeline(alike_t(dense_descriptors=True).eval(), ep, BF, B, 96, 128, device=DEV, place_map=False)
    ref.run(images)
    assert ref.placement is None
    monkeypatch.setattr(pipeline, "PLACE_MIN_BYTES", 0)
    monkeypatch.setattr(pipeline, "PLACE_CANDIDATES", 3)
    placed = pipeline.PairPipeline(alike_t(dense_descriptors=True).eval(), ep, BF, B, 96, 128, device=DEV, place_map=True)
    before = torch.cuda.memory_allocated()
    placed.run(images)
    rec = placed.placement
    assert rec["candidates"] == 3 and len(rec["forward_ms"]) == 3 and rec["forward_ms"][rec["chosen"]] == min(rec["forward_ms"])
    assert torch.cuda.memory_allocated() <= before + (1 << 20)          # the two losers are gone
    placed.run(images)
    assert placed.placement is rec and placed.place_map is False        # once per pipeline
    for a, b in ((placed.n, ref.n), (placed.k, ref.k), (placed.desc, ref.desc), (placed.score, ref.score)):
        assert torch.equal(a, b)
    for i in range(2 * B):          # rows beyond an image's count are whatever torch.empty left there
        n = int(ref.n[i])
        assert n > 50 and torch.equal(placed.kps[i, :n], ref.kps[i, :n])
    for b in range(B):
        k = int(ref.k[b])
        assert k > 10 and torch.equal(placed.pairs[b, :k], ref.pairs[b, :k]) and torch.equal(placed.dist[b, :k], ref.dist[b, :k])
    sparse = pipeline.PairPipeline(alike_t(dense_descriptors=False).eval(), ep, BF, B, 96, 128, device=DEV, place_map=True)
    sparse.run(images)
    assert sparse.placement is None                                      # no dense map, nothing to place
