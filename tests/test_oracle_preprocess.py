"""The input-transform restatement (oracle.numpy_ref.preprocess).  cv2 is absent from the image, so the resize is
"parity unpinned"; what can be pinned without it is checked here: the ToTensor case against torch/numpy arithmetic, and
two exact consequences of INTER_LINEAR's half-pixel rule."""
import numpy as np

from oracle import numpy_ref


def test_no_resize_is_to_tensor():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    out = numpy_ref.preprocess(img)
    want = (img.astype(np.float32) / np.float32(255)).transpose(2, 0, 1)          # transforms.ToTensor / hpatches.py:59-60
    assert out.dtype == np.float32 and np.array_equal(out, want)
    assert np.array_equal(numpy_ref.preprocess(img, bgr=True), want[::-1])


def test_halving_is_a_box_average_and_identity_size_is_identity():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    f = (img.astype(np.float32) / np.float32(255)).transpose(2, 0, 1)
    half = numpy_ref.preprocess(img, size=(32, 48))
    box = (f[:, 0::2, 0::2] * 0.5 + f[:, 0::2, 1::2] * 0.5) * 0.5 + (f[:, 1::2, 0::2] * 0.5 + f[:, 1::2, 1::2] * 0.5) * 0.5
    assert np.array_equal(half, box.astype(np.float32))        # (d + 0.5) * 2 - 0.5 = 2d + 0.5: taps 2d, 2d+1, weights 1/2
    assert np.array_equal(numpy_ref.preprocess(img, size=(64, 96)), f)
    up = numpy_ref.preprocess(img, size=(128, 192))
    assert up.shape == (3, 128, 192) and up.min() >= 0 and up.max() <= 1
    assert np.array_equal(up[:, 0, 0], f[:, 0, 0]) and np.array_equal(up[:, -1, -1], f[:, -1, -1])    # clamped ends
