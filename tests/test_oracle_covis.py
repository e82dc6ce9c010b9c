"""The covisibility oracle (oracle.warp_homography / val_key_points) against fixtures the reference produced
(tests/golden/covis.npz, made by tests/golden/make_golden_covis.py).  Bit-exact: these are fp32 formulas."""
import os

import numpy as np
import pytest

import oracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "covis.npz"))
CASES = range(int(G["n_cases"]))


def warps(c):
    p = "c%d_" % c
    w0, h0 = G[p + "wh0"]; w1, h1 = G[p + "wh1"]
    w01 = dict(homography_matrix=G[p + "hm"], width=int(w1), height=int(h1))
    w10 = dict(homography_matrix=G[p + "hinv"], width=int(w0), height=int(h0))
    if int(G[p + "resize"]):
        w01["resize"] = w10["resize"] = int(G[p + "resize"])
    return p, w01, w10


@pytest.mark.parametrize("c", CASES)
def test_warp_homography_matches_reference(c):
    p, w01, w10 = warps(c)
    for kps, w, names in ((G[p + "kps0"], w01, ("k0v", "k01v", "ids", "ids_out")), (G[p + "kps1"], w10, ("k1v", "k10v", "ids1", "ids1_out"))):
        a, b, ids, ids_out = oracle.warp_homography(kps[:, :2], w["homography_matrix"], w["width"], w["height"])
        assert np.array_equal(ids, G[p + names[2]]) and np.array_equal(ids_out, G[p + names[3]])
        assert np.array_equal(a, G[p + names[0]])
        assert np.array_equal(b, G[p + names[1]])


@pytest.mark.parametrize("c", CASES)
def test_val_key_points_matches_reference(c):
    p, w01, w10 = warps(c)
    r = oracle.val_key_points(G[p + "kps0"], G[p + "kps1"], w01, w10, th=3)
    assert r["num_feat"] == int(G[p + "num_feat"])
    assert np.array_equal(r["pairs"], G[p + "pairs"])
    assert np.array_equal(r["dist"], G[p + "dist"])
    assert np.array_equal(r["errors"], G[p + "errors"])
    assert np.float32(r["repeatability"]) == G[p + "repeatability"]
    assert np.array_equal(np.float32(r["mean_error"]), G[p + "mean_error"], equal_nan=True)
