"""Pins the oracle's FundamentalMatrix chain (BASELINE configs[3]) on fixtures the reference's own
tasks/FundamentalMatrix.py produced (tests/golden/make_golden_fund.py)."""
import numpy as np

import oracle
from conftest import load_golden


def params_of(prm):
    nms, border, top_k, maxd, th, _ = prm
    return {"extractor_params": dict(nms_dist=int(nms), threshold=0.0, border_dist=int(border), top_k=int(top_k), min_score=0.0),
            "matcher_params": {"type": "brute_force", "brute_force_params": dict(metric="euclidean", max_distance=float(maxd), cross_check=True)},
            "FundamentalMatrix_params": {"th": float(th)}}


def test_fundamental_matrix_chain_against_reference():
    g = load_golden("fund.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        prm = params_of(g[p + "prm"])
        mean, ratio, num, m0, _ = oracle.fundamental_matrix(g[p + "score0"], g[p + "score1"], g[p + "desc0"][0].astype(np.float32),
                                                            g[p + "desc1"][0].astype(np.float32), g[p + "F"], prm)
        want = g[p + "result"]
        assert len(m0) == len(g[p + "pairs"]), p
        assert num == int(want[2]) and ratio == want[1], p
        np.testing.assert_allclose(mean, want[0], rtol=2e-6, err_msg=p)
