"""GPU parity of SURVEY 8(f) rank 4 (csrc/lk.hip through the C ABI): the tensor Lucas-Kanade tracker against the fixtures
the reference's OpticalFlow produced and against the oracle.

Tolerance 1e-3 px: window sums are butterfly reductions over lanes here, sequential in the oracle, einsum in the
reference; the iteration is contractive, so rounding differences stay at the 1e-4 px level (observed)."""
import numpy as np
import pytest
import torch

import oracle
from keypoint_bench_amd import synthetic
from test_oracle_lk import G, lk_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL_PX = 1e-3


@pytest.mark.parametrize("c", range(int(G["n_cases"])))
def test_lk_matches_reference_and_oracle(c):
    from keypoint_bench_amd.utils.matcher import OpticalFlow
    v0, v1, pts, unit, prm, want, want_err = lk_case(c)
    angle = torch.atan2(torch.from_numpy(unit[:, 1]), torch.from_numpy(unit[:, 0])).to(DEV)
    t = lambda a: torch.from_numpy(a).to(DEV)
    got, err = OpticalFlow(prm)(t(v0)[None], t(v1)[None], t(pts), t(pts), random_angle=angle)
    assert got.shape == (1, len(pts), 2) and err.shape == (1, len(pts))
    got, err = got[0].cpu().numpy(), err[0].cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=ATOL_PX)
    np.testing.assert_allclose(err, want_err, rtol=0, atol=ATOL_PX)
    # the unit vectors went through atan2/cos/sin on the way in: compare with the oracle on exactly what the kernel saw
    u = torch.stack([torch.cos(angle), torch.sin(angle)], 1).cpu().numpy()
    exp, exp_err = oracle.lk_track(v0, v1, pts, pts, u, prm["distance"], prm["win_size"], prm["levels"], prm["interation"])
    np.testing.assert_allclose(got, exp, rtol=0, atol=ATOL_PX)


def test_lk_full_size_finds_the_shift():
    """480x640, the configured 21x21 window on 3 levels, 1000 points: view1 is view0 shifted by (3, 2) px, so points that
    converge must report that flow (with a 10 px random start and the reference's update rule about two thirds do; the
    reference's own fixture has 53 of 60); also checks the oracle on a sample of them."""
    from keypoint_bench_amd.utils.matcher import optical_flow_tensor
    v0, v1 = synthetic.image_pair(5)
    rng = np.random.default_rng(9)
    pts = np.stack([rng.uniform(0.1, 0.9, 1000), rng.uniform(0.1, 0.9, 1000)], 1).astype(np.float32)
    prm = dict(distance=10, win_size=21, levels=3, interation=40, gray=False)       # config/config_fund.yaml:72-77
    t = lambda a: torch.from_numpy(a).to(DEV)
    torch.manual_seed(0)
    out = optical_flow_tensor(t(pts), t(pts), t(v0)[None], t(v1)[None], prm)[0].cpu().numpy()
    flow = out - pts * np.array([639, 479], np.float32)
    good = np.abs(flow - np.array([-3, -2])).max(1) < 0.25
    assert good.mean() > 0.5, good.mean()
    ang = rng.normal(size=32).astype(np.float32) * 6.28
    u = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    from keypoint_bench_amd.utils.matcher import OpticalFlow
    got, _ = OpticalFlow(prm)(t(v0)[None], t(v1)[None], t(pts[:32]), t(pts[:32]), random_angle=t(ang))
    u_dev = torch.stack([torch.cos(t(ang)), torch.sin(t(ang))], 1).cpu().numpy()
    exp, _ = oracle.lk_track(v0, v1, pts[:32], pts[:32], u_dev, 10, 21, 3, 40)
    np.testing.assert_allclose(got[0].cpu().numpy(), exp, rtol=0, atol=ATOL_PX)


def test_lk_rejects_what_the_reference_cannot_run():
    from keypoint_bench_amd.utils.matcher import OpticalFlow
    img = torch.zeros((1, 64, 96, 128), device=DEV)
    with pytest.raises(ValueError):
        OpticalFlow()(img, img, torch.zeros((4, 2), device=DEV), torch.zeros((4, 2), device=DEV))
