"""GPU parity of M1..M3 (csrc/match.hip through the C ABI) against the goldens and the oracle."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_sampling_matches_reference_goldens():
    from keypoint_bench_amd.utils.matcher import sample_descriptors
    g = load_golden("match.npz")
    for name in g["cases"]:
        for side in "01":
            dm = torch.from_numpy(g[name + ".dm" + side]).to(DEV)
            p = torch.from_numpy(g[name + ".p" + side]).to(DEV)
            for mem in (dm, dm.contiguous(memory_format=torch.channels_last)):
                got = sample_descriptors(p, mem).cpu().numpy()
                # tolerance 1e-4 is north_star's descriptor bound; observed error is a few ulp
                np.testing.assert_allclose(got, g[name + ".sdesc" + side], rtol=0, atol=2e-6, err_msg=name)
                np.testing.assert_array_equal(got, oracle.sample(g[name + ".dm" + side][0], g[name + ".p" + side]))


def test_match_goldens_bit_exact():
    from keypoint_bench_amd.utils.matcher import match_descriptors
    g = load_golden("match.npz")
    for name in g["cases"]:
        maxd, cc = g[name + ".prm"]
        d0 = torch.from_numpy(g[name + ".sdesc0"]).to(DEV)
        d1 = torch.from_numpy(g[name + ".sdesc1"]).to(DEV)
        pairs, dist = match_descriptors(d0, d1, max_distance=float(maxd), cross_check=bool(cc), return_distance=True)
        np.testing.assert_array_equal(pairs.cpu().numpy(), g[name + ".pairs"], err_msg=name)
        np.testing.assert_array_equal(dist.cpu().numpy(), g[name + ".dist"], err_msg=name)


def test_brute_force_matcher_goldens():
    from keypoint_bench_amd.utils.matcher import brute_force_matcher
    g = load_golden("match.npz")
    for name in g["cases"]:
        maxd, cc = g[name + ".prm"]
        prm = {"metric": "euclidean", "max_distance": float(maxd), "cross_check": bool(cc)}
        m0, m1 = brute_force_matcher(torch.from_numpy(g[name + ".p0"]).to(DEV), torch.from_numpy(g[name + ".p1"]).to(DEV),
                                     torch.from_numpy(g[name + ".dm0"]).to(DEV), torch.from_numpy(g[name + ".dm1"]).to(DEV), prm)
        np.testing.assert_array_equal(m0.cpu().numpy(), g[name + ".m0"], err_msg=name)
        np.testing.assert_array_equal(m1.cpu().numpy(), g[name + ".m1"], err_msg=name)


def test_alike_pair_goldens():
    from keypoint_bench_amd.utils.matcher import brute_force_matcher, match_descriptors
    g = load_golden("alike_t.npz")
    d0 = torch.from_numpy(g["full.sdesc0"]).to(DEV)
    d1 = torch.from_numpy(g["full.sdesc1"]).to(DEV)
    pairs, dist = match_descriptors(d0, d1, max_distance=5, cross_check=True, return_distance=True)
    np.testing.assert_array_equal(pairs.cpu().numpy(), g["full.pairs"])
    np.testing.assert_array_equal(dist.cpu().numpy(), g["full.dist"])
    prm = {"metric": "euclidean", "max_distance": 5, "cross_check": True}
    m0, m1 = brute_force_matcher(torch.from_numpy(g["small.kps0"]).to(DEV), torch.from_numpy(g["small.kps1"]).to(DEV),
                                 torch.from_numpy(g["small.desc0"])[None].to(DEV), torch.from_numpy(g["small.desc1"])[None].to(DEV), prm)
    np.testing.assert_array_equal(m0.cpu().numpy(), g["small.m0"])
    np.testing.assert_array_equal(m1.cpu().numpy(), g["small.m1"])


@pytest.mark.parametrize("n,m,C", [(1, 1, 3), (65, 63, 7), (200, 333, 64), (1000, 1000, 64), (700, 1000, 256), (130, 5, 128)])
def test_match_random_vs_oracle(n, m, C):
    from keypoint_bench_amd.utils.matcher import match_descriptors
    rng = np.random.default_rng(n * 7 + m)
    a = rng.normal(size=(n, C)).astype(np.float32)
    b = rng.normal(size=(m, C)).astype(np.float32)
    k = min(n, m) // 2
    b[:k] = a[rng.permutation(n)[:k]] + 0.05 * rng.normal(size=(k, C)).astype(np.float32)
    if C <= 7:  # force exact ties
        a = np.round(a)
        b = np.round(b)
    for cc in (True, False):
        for maxd in (np.inf, float(np.sqrt(C)) * 0.5):
            want_p, want_d = oracle.match(a, b, maxd, cc)
            got_p, got_d = match_descriptors(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), max_distance=maxd,
                                             cross_check=cc, return_distance=True)
            np.testing.assert_array_equal(got_p.cpu().numpy(), want_p)
            np.testing.assert_array_equal(got_d.cpu().numpy(), want_d)


def test_empty_inputs():
    from keypoint_bench_amd.utils.matcher import brute_force_matcher
    dm = torch.rand(1, 16, 8, 8, device=DEV)
    prm = {"metric": "euclidean", "max_distance": 5, "cross_check": True}
    m0, m1 = brute_force_matcher(torch.zeros(0, 3, device=DEV), torch.rand(5, 3, device=DEV), dm, dm, prm)
    assert m0.shape == (0, 3) and m1.shape == (0, 3)


def test_match_symmetry_property_full_size():
    """Size-independent property at BASELINE size: with cross_check the match set of (A,B) is the
    transpose of the match set of (B,A)."""
    from keypoint_bench_amd.utils.matcher import match_descriptors
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn(1000, 64, generator=g).to(DEV)
    b = (a[torch.randperm(1000, generator=g)] + 0.1 * torch.randn(1000, 64, generator=g).to("cpu")).to(DEV) if False else (a.cpu()[torch.randperm(1000, generator=g)] + 0.1 * torch.randn(1000, 64, generator=g)).to(DEV)
    p = match_descriptors(a, b, max_distance=5.0, cross_check=True).cpu().numpy()
    q = match_descriptors(b, a, max_distance=5.0, cross_check=True).cpu().numpy()
    assert p.shape[0] > 900
    assert set(map(tuple, p.tolist())) == set((j, i) for i, j in q.tolist())


@pytest.mark.parametrize("maxd", [np.inf, 1.0])
@pytest.mark.parametrize("cc", [True, False])
def test_nan_and_inf_descriptors_follow_numpy_argmin(maxd, cc):
    """A NaN descriptor row makes every distance of that row/column NaN; numpy.argmin then returns the FIRST NaN, and
    skimage filters on distance only `if max_distance < np.inf`.  The kernel must never emit an out-of-range index."""
    from keypoint_bench_amd.utils.matcher import match_descriptors
    from test_oracle_match_edge import cases
    for name, a, b in cases():
        want, wd = oracle.match(a, b, maxd, cc)
        got, gd = match_descriptors(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), max_distance=maxd, cross_check=cc,
                                    return_distance=True)
        got, gd = got.cpu().numpy(), gd.cpu().numpy()
        assert got.size == 0 or (got[:, 1].max() < len(b) and got.min() >= 0), name
        assert np.array_equal(got, want), (name, maxd, cc)
        assert np.array_equal(gd, wd, equal_nan=True), (name, maxd, cc)
