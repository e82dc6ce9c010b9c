"""The oracle chain FROM PIXELS against what the reference itself produced (tests/golden/e2e.npz, make_golden_e2e.py): full-size
translated and viewpoint-warped pairs through oracle/alike_ref (torch fp32, BN folded) + the C detection / covisibility /
sampling / match + the numpy RANSAC restatement must give the reference's keypoint pixel sets, match pixel pairs, repeatability
and MHA flags.  This is what entitles tests/test_gpu_metric_from_pixels.py to use the oracle chain on 64 more pairs."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle import alike_ref
from conftest import load_golden
from keypoint_bench_amd import synthetic, weights

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
H, W = 480, 640
EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
BF = dict(metric="euclidean", max_distance=5, cross_check=True)
CASES = [str(c) for c in load_golden("e2e.npz")["cases"]]


@pytest.mark.parametrize("case", CASES)          # 8 translated + 16 warped pairs, ~1 s each
def test_oracle_chain_from_pixels_equals_the_reference(case):
    import metric_sweep
    g = load_golden("e2e.npz")
    i = int(case[5:]) if case.startswith("shift") else int(case[4:])
    if case.startswith("shift"):
        v0, v1 = synthetic.image_pair(i)
    else:
        v0, v1, h01 = synthetic.warped_pair(i, H, W, *synthetic.viewpoint_case(i))
        assert np.array_equal(h01, g[case + ".h01"])
    assert [synthetic.checksum(v0), synthetic.checksum(v1)] == [str(s) for s in g[case + ".img.sum"]]
    h01 = g[case + ".h01"]
    h10 = np.linalg.inv(h01.astype(np.float64)).astype(np.float32)
    t = {k: torch.from_numpy(v) for k, v in weights.load_alike_t().items()}
    ks, ds = [], []
    for v, key in ((v0, "0"), (v1, "1")):
        with torch.no_grad():
            s, d = alike_ref.alnet_forward(torch.from_numpy(v)[None], t)
        k, idx = oracle.detection(s[0, 0].numpy(), EP)
        want = g[case + ".idx" + key].astype(np.int64)
        assert set(np.asarray(idx).tolist()) == set(want.tolist())
        og, ow = np.argsort(np.asarray(idx), kind="stable"), np.argsort(want, kind="stable")
        np.testing.assert_allclose(k[og, 2], g[case + ".score" + key][ow], rtol=0, atol=5e-6)
        ks.append(k), ds.append(d[0].numpy())
    # brute-force matches on ALL keypoints (utils/matcher.py:206-234)
    pairs, _ = oracle.match(oracle.sample(ds[0], ks[0]), oracle.sample(ds[1], ks[1]), BF["max_distance"], BF["cross_check"])
    i0, i1 = metric_sweep.flat_index(ks[0]), metric_sweep.flat_index(ks[1])
    wp = g[case + ".pairs"].astype(np.int64)
    assert set(zip(i0[pairs[:, 0]].tolist(), i1[pairs[:, 1]].tolist())) == \
        set(zip(g[case + ".idx0"].astype(np.int64)[wp[:, 0]].tolist(), g[case + ".idx1"].astype(np.int64)[wp[:, 1]].tolist()))
    row = metric_sweep.cpu_chain(ks[0], ks[1], ds[0], ds[1], h01, h10)
    wr = g[case + ".rep"]
    assert row["num_feat"] == int(wr[0]) and np.float32(row["rep"]) == np.float32(wr[1])
    assert abs(row["rep_err"] - wr[2]) <= 1e-5
    assert row["flags"] == g[case + ".mha"].tolist()
    if case + ".mha_n" in g:
        assert row["matches"] == int(g[case + ".mha_n"])
