"""GPU parity of SURVEY 8(f) rank 2 (csrc/preprocess.hip through the C ABI) against the numpy restatement: bit-exact
(same fp32 formula, no fused multiply-add on either side)."""
import numpy as np
import pytest
import torch

from oracle import numpy_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape,size,bgr", [((480, 640), None, False), ((600, 800), 512, True), ((333, 517), (480, 640), True),
                                             ((1024, 768), 512, False), ((64, 96), (128, 192), False)])
def test_preprocess_matches_oracle(shape, size, bgr):
    from keypoint_bench_amd.utils.preprocess import to_tensor_resized
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    got = to_tensor_resized(img, size, bgr, DEV)
    want = numpy_ref.preprocess(img, size, bgr)
    assert got.shape[1:] == want.shape and got.dtype == torch.float32
    np.testing.assert_array_equal(got[0].cpu().numpy(), want)


def test_preprocess_batch_feeds_the_net():
    from keypoint_bench_amd.utils.preprocess import to_tensor_resized
    from keypoint_bench_amd.models.ALike import alike_t
    rng = np.random.default_rng(3)
    imgs = rng.integers(0, 256, (3, 500, 700, 3), dtype=np.uint8)
    x = to_tensor_resized(imgs, 512, True, DEV)
    assert x.shape == (3, 3, 512, 512)
    for b in range(3):
        np.testing.assert_array_equal(x[b].cpu().numpy(), numpy_ref.preprocess(imgs[b], 512, True))
    score, desc = alike_t().eval()(x)
    assert score.shape == (3, 1, 512, 512) and torch.isfinite(score).all()
