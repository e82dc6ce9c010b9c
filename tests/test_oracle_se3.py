"""The depth-warp oracle (oracle.warp_se3) against fixtures produced by the reference's utils/projection.py
(tests/golden/se3.npz, made by tests/golden/make_golden_se3.py).  Bit-exact on all cases, including the ones small enough
for torch's unfused matmul path."""
import os
import sys

import numpy as np
import pytest

import oracle

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_golden_se3 import scene          # the deterministic synthetic scene builder (numpy only)

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "se3.npz"))


def se3_case(c):
    k = "c%d_" % c
    seed, H0, W0, H1, W1, planar = (int(v) for v in G[k + "scene"])
    d0, d1, k0, k1, pose, b0, b1 = scene(seed, H0, W0, H1, W1, bool(planar))
    return dict(kps=G[k + "kps"], depth0=d0, depth1=d1, k0=k0, kinv0=G[k + "kinv0"], k1=k1, pose=pose, bbox0=b0, bbox1=b1,
                want=(G[k + "k0v"], G[k + "k01v"], G[k + "ids"], G[k + "ids_out"]))


@pytest.mark.parametrize("c", range(int(G["n_cases"])))
def test_warp_se3_oracle_matches_reference(c):
    s = se3_case(c)
    a, b, ids, out = oracle.warp_se3(s["kps"], s["depth0"], s["depth1"], s["kinv0"], s["k1"], s["pose"], s["bbox0"], s["bbox1"])
    for got, want in zip((a, b, ids, out), s["want"]):
        np.testing.assert_array_equal(got, want)
