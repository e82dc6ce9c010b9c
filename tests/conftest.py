import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def params_from(arr):
    return dict(nms_dist=int(arr[0]), threshold=float(arr[1]), border_dist=int(arr[2]), top_k=int(arr[3]),
                min_score=float(arr[4]))


def assert_kps_equal(got, want, top_k, what=""):
    """Bit-exact comparison of detection outputs, with the documented tie rule:
    the reference's argsort leaves equal scores in arbitrary order, so rows whose score is tied are
    compared as sets and a tie group cut by top_k only has to agree in size and score."""
    got = np.asarray(got, np.float32).reshape(-1, 3)
    want = np.asarray(want, np.float32).reshape(-1, 3)
    assert got.shape == want.shape, "%s: %s vs %s" % (what, got.shape, want.shape)
    if got.shape[0] == 0:
        return
    if np.array_equal(got.view(np.uint32), want.view(np.uint32)):
        return
    # not bitwise equal row by row: only legal if the differing rows belong to tie groups
    assert np.array_equal(np.sort(got[:, 2]).view(np.uint32), np.sort(want[:, 2]).view(np.uint32)), what + ": score multiset differs"
    vals, counts = np.unique(want[:, 2], return_counts=True)
    tied = set(vals[counts > 1].tolist())
    cut = want[:, 2].min() if want.shape[0] == top_k else None   # group that top_k may have cut
    gs = {tuple(r) for r in got.view(np.uint32).tolist()}
    ws = {tuple(r) for r in want.view(np.uint32).tolist()}
    for r in (gs ^ ws):
        s = np.array([r[2]], np.uint32).view(np.float32)[0]
        assert cut is not None and s == cut, "%s: row with score %r differs outside the cut tie group" % (what, s)
    diff = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
    for i in diff:
        assert want[i, 2] in tied or (cut is not None and want[i, 2] == cut), "%s: row %d differs and is not tied" % (what, i)
