"""Run-to-run determinism.  Every kernel of this library is a fixed sequence of fp32 / integer operations with no atomics on floating
point, so the same input must give the same BITS on every run.  In r04 it did not: from the commit that made the ALIKE head's a2
interpolation four weighted taps (775c39f) about 0.5 % of the dense maps differed from one run to the next in 16-pixel groups
(pixels 16..31 of a 32-pixel tile) by 1e-5 .. 1e-1 -- inside every parity tolerance, so no oracle comparison saw it; one shape test
failing once did.  r05 named the cause: gfx950 returns a wrong LOW result in lanes 48..63 for v_pk_{mul,add,fma}_f32 with op_sel[0] = 0,
op_sel[1] = 1 while an f16 MFMA executes on the SIMD (scripts/ubench/pk_opsel.hip; the build rewrites the encoding,
keypoint_bench_amd/isa_fixup.py, and scripts/isa_lint.py checks the library).  Faults of this kind depend on what else the SIMD is
doing: they come and go with unrelated edits and with occupancy, and only repetition shows them.  Hence this file: each network, several
runs of the same batch, bit for bit; sizes chosen so that thousands of tiles pass through every matrix kernel per run."""
import numpy as np
import pytest
import torch

from keypoint_bench_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(seed, H, W, B):
    imgs = [synthetic.image_pair(seed + i, H, W)[0] for i in range(min(B, 4))]
    x = torch.from_numpy(np.stack(imgs)).to(DEV)
    return x.repeat((B + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:B].contiguous()


def _same_bits(model, x, runs):
    with torch.no_grad():
        model(x)                                    # allocations, lazy initialisation
        ref = [t.clone() for t in model(x) if isinstance(t, torch.Tensor)]
        for it in range(runs):
            out = [t for t in model(x) if isinstance(t, torch.Tensor)]
            for k, (a, b) in enumerate(zip(out, ref)):
                if not torch.equal(a, b):
                    d = (a != b)
                    where = torch.nonzero(d.reshape(d.shape[0], -1).any(dim=1)).flatten().tolist()
                    raise AssertionError("run %d, output %d: %d elements differ from the first run (images %s), max |d| %.3g"
                                         % (it, k, int(d.sum()), where[:8], (a - b).abs().max().item()))


@pytest.mark.parametrize("shape,batch,runs", [((480, 640), 96, 30), ((480, 640), 1, 300), ((480, 640), 3, 100), ((800, 1216), 4, 60), ((1216, 1600), 1, 300)])
@pytest.mark.parametrize("dense", [True, False])
def test_alike_gives_the_same_bits_every_run(shape, batch, runs, dense):
    from keypoint_bench_amd.models.ALike import alike_t
    _same_bits(alike_t(dense_descriptors=dense).eval(), _batch(41, *shape, batch), runs if dense else max(runs // 3, 4))


@pytest.mark.parametrize("batch,runs", [(8, 20), (1, 60)])
def test_superpoint_gives_the_same_bits_every_run(batch, runs):
    from keypoint_bench_amd.models.SuperPoint import superpoint_random
    _same_bits(superpoint_random(7).eval(), _batch(42, 480, 640, batch), runs)


@pytest.mark.parametrize("batch,runs", [(64, 30), (1, 200)])
def test_xfeat_gives_the_same_bits_every_run(batch, runs):
    from keypoint_bench_amd.models.XFeat import xfeat_random
    _same_bits(xfeat_random(5).eval(), _batch(43, 480, 640, batch), runs)


@pytest.mark.parametrize("batch,runs", [(2, 10), (1, 20)])
def test_disk_gives_the_same_bits_every_run(batch, runs):
    from keypoint_bench_amd.models.disk import disk_random
    _same_bits(disk_random(3).eval(), _batch(44, 480, 640, batch), runs)


def test_lightglue_gives_the_same_matches_every_run():
    import sys
    from conftest import GOLDEN
    from keypoint_bench_amd import weights
    from keypoint_bench_amd.models.lightglue import LightGlue
    sys.path.insert(0, GOLDEN)
    import make_golden_lightglue as mk
    dim, scale, seed = 256, 8, 23
    dm0, dm1, p0, p1 = mk.inputs(seed, dim, scale, n0=1000, n1=1000)
    m = LightGlue(features=None, desc_scale=scale)
    m.load_state_dict(weights.random_lightglue_state_dict(seed, dim, "plain"))
    T = lambda a: torch.from_numpy(a).to(DEV)
    args = (T(p0), T(p1), T(dm0), T(dm1), {"w": 320, "h": 240})
    m.match_indices(*args)
    pairs0, scores0, stop0 = m.match_indices(*args)
    for it in range(20):
        pairs, scores, stop = m.match_indices(*args)
        assert stop == stop0 and torch.equal(pairs, pairs0) and torch.equal(scores, scores0), "run %d differs from the first" % it


def test_detection_and_matching_give_the_same_rows_every_run():
    """The integer half of the path builds its lists with atomics (NMS maxima, undecided pixels, candidates): the ORDER in which
    lanes arrive differs from run to run, the rows handed back must not."""
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import brute_force_matcher
    EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
    BF = dict(metric="euclidean", max_distance=5, cross_check=True)
    v0, v1 = synthetic.image_pair(3)
    net = alike_t(dense_descriptors=True).eval()
    with torch.no_grad():
        s0, d0 = net(torch.from_numpy(v0)[None].to(DEV))
        s1, d1 = net(torch.from_numpy(v1)[None].to(DEV))
    k0, k1 = detection(s0, EP), detection(s1, EP)
    m0, m1 = brute_force_matcher(k0, k1, d0, d1, BF)
    for it in range(40):
        a0, a1 = detection(s0, EP), detection(s1, EP)
        assert torch.equal(a0, k0) and torch.equal(a1, k1), "detection, run %d" % it
        b0, b1 = brute_force_matcher(a0, a1, d0, d1, BF)
        assert torch.equal(b0, m0) and torch.equal(b1, m1), "matcher, run %d" % it


def test_batched_detection_gives_the_same_rows_every_run():
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.extracter import detection_batch
    EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
    net = alike_t(dense_descriptors=False).eval()
    with torch.no_grad():
        s, _ = net(_batch(11, 480, 640, 64))
    def rows():
        kps, idx, n = detection_batch(s, EP)
        valid = torch.arange(kps.shape[1], device=DEV)[None, :] < n[:, None]       # rows past n are not written
        return n.clone(), torch.where(valid[..., None], kps, torch.zeros_like(kps)), torch.where(valid, idx, torch.zeros_like(idx))
    ref = rows()
    for it in range(10):
        for a, b in zip(rows(), ref):
            assert torch.equal(a, b), "run %d" % it


@pytest.mark.parametrize("task", ["repeatability", "MHA", "FundamentalMatrix"])
def test_task_rows_through_the_runner_are_the_same_every_run(task):
    """The whole chain a task row comes out of -- nets, detection, covisibility warp, matcher, the seeded RANSAC estimators on the
    device -- five runs of the same eight pairs."""
    from keypoint_bench_amd import runner
    EP = dict(nms_dist=4, threshold=0.0, border_dist=8, top_k=300, min_score=0.0)
    BF = dict(metric="euclidean", max_distance=5, cross_check=True)
    H01 = np.array([[1, 0, -3], [0, 1, -2], [0, 0, 1]], np.float32)
    prm = {"model_type": "Alike", "task_type": task, "Alike_params": dict(c1=8, c2=16, c3=32, c4=64, dim=64), "extractor_params": EP,
           "matcher_params": {"type": "brute_force", "brute_force_params": BF}, "repeatability_params": {"th": 3},
           "FundamentalMatrix_params": {"th": 3.0}, "MHA_params": {"th": [3, 5, 7]}}
    ds = []
    for i in range(8):
        v0, v1 = synthetic.image_pair(200 + i, 192, 256)
        ds.append({"image0": v0, "image1": v1, "dataset": "HPatches",
                   "warp01_params": dict(mode="homo", homography_matrix=H01, width=np.int64(256), height=np.int64(192), resize=np.int64(256)),
                   "warp10_params": dict(mode="homo", homography_matrix=np.linalg.inv(H01).astype(np.float32), width=256, height=192)})
    _, ref = runner.PairRunner(prm, device=DEV, batch=4).run(ds)
    for it in range(5):
        _, rows = runner.PairRunner(prm, device=DEV, batch=4).run(ds)
        assert rows.tobytes() == ref.tobytes(), "run %d" % it
