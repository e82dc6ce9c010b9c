"""FundamentalMatrix task (BASELINE configs[3]) on the device against fixtures the reference's own
tasks/FundamentalMatrix.py produced, and the epipolar kernel's three kps1 modes against the oracle formula."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden
from test_oracle_fund import params_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("case", range(4))
def test_fundamental_matrix_against_reference(case):
    from keypoint_bench_amd.tasks.FundamentalMatrix import fundamental_matrix
    g = load_golden("fund.npz")
    p = "c%d_" % case
    prm = params_of(g[p + "prm"])
    if g[p + "prm"][5] == 1:
        prm["matcher_params"]["type"] = "light_glue"            # matcher None: FundamentalMatrix.py:124-126
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    batch = {"fundamental": t(g[p + "F"])[None]}
    res = fundamental_matrix(case, None, batch, t(g[p + "score0"])[None, None], t(g[p + "score1"])[None, None], t(g[p + "desc0"]),
                             t(g[p + "desc1"]), None, prm)
    want = g[p + "result"]
    assert res["fundamental_num"] == int(want[2]) and res["fundamental_radio"] == want[1]
    np.testing.assert_allclose(float(res["fundamental_error"]), want[0], rtol=2e-6)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_epipolar_modes_against_oracle(mode):
    from keypoint_bench_amd.tasks.FundamentalMatrix import epipolar_error
    rng = np.random.default_rng(3 + mode)
    B, K, W, H = 3, 257, 640, 480
    k0 = rng.random((B, K, 3)).astype(np.float32)
    k1 = rng.random((B, K, 3 if mode < 2 else 2)).astype(np.float32)
    if mode == 2:
        k1 *= np.array([W - 1, H - 1], np.float32)
    F = rng.normal(size=(B, 3, 3)).astype(np.float32)
    kk = np.array([K, 100, 0], np.int32)
    err, stats = epipolar_error(torch.from_numpy(k0).to(DEV), torch.from_numpy(k1).to(DEV), torch.from_numpy(F).to(DEV), W, H, mode, 0.05,
                                k_dev=torch.from_numpy(kk).to(DEV))
    err, stats = err.cpu().numpy(), stats.cpu().numpy()
    for b in range(2):
        want = oracle.epipolar_error(k0[b, :kk[b]], k1[b, :kk[b]], F[b], W, H, mode)
        np.testing.assert_allclose(err[b, :kk[b]], want, rtol=2e-5, atol=1e-6)      # fp32 3-term dots of random signs: cancellation
        np.testing.assert_allclose(stats[b, 0], want.mean(dtype=np.float64), rtol=1e-6)
        assert stats[b, 2] == (err[b, :kk[b]] < 0.05).sum()
    assert np.isnan(stats[2, 0]) and stats[2, 2] == 0
