"""M2 edge semantics of the oracle against scipy.cdist + skimage's documented glue (tests/golden/skimage_standin.py):
NaN descriptors (numpy.argmin: the first NaN wins), infinite distances, and the `max_distance < inf` guard."""
import os
import sys

import numpy as np
import pytest

import oracle
from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import skimage_standin  # noqa: E402


def cases():
    rng = np.random.default_rng(21)
    a = rng.normal(size=(40, 8)).astype(np.float32)
    b = (a[rng.permutation(40)][:33] + 0.05 * rng.normal(size=(33, 8))).astype(np.float32)
    yield "plain", a, b
    a2, b2 = a.copy(), b.copy()
    a2[[3, 17]] = np.nan
    b2[5] = np.nan
    yield "nan_rows", a2, b2
    a3, b3 = a.copy(), b.copy()
    a3[7, 0] = np.inf
    yield "inf_row", a3, b3


@pytest.mark.parametrize("maxd", [np.inf, 1.0])
@pytest.mark.parametrize("cc", [True, False])
def test_match_edges_equal_scipy_glue(maxd, cc):
    for name, a, b in cases():
        with np.errstate(invalid="ignore"):
            want = skimage_standin.match_descriptors(a, b, metric="euclidean", max_distance=maxd, cross_check=cc)
            wd = skimage_standin.distances_of(a, b, want)
        got, gd = oracle.match(a, b, maxd, cc)
        assert np.array_equal(got, want), (name, maxd, cc, got.tolist(), want.tolist())
        assert np.array_equal(gd, wd, equal_nan=True), (name, maxd, cc)
