"""oracle/geometry_ref.py (RANSAC homography, PARITY UNPINNED against cv2) validated on ANALYTIC ground truth, and the
MHA chain around the estimator pinned on fixtures the reference's own tasks/MHA.py produced
(tests/golden/make_golden_mha.py: the estimator call inside it was answered by this restatement)."""
import numpy as np
import pytest

import oracle
from conftest import load_golden
from oracle import geometry_ref as g


def synth(n, inlier_share, noise, seed):
    rng = np.random.default_rng(seed)
    H = np.eye(3) + rng.normal(0, 0.05, (3, 3)) * np.array([[1, 1, 30], [1, 1, 30], [1e-4, 1e-4, 0]])
    src = rng.random((n, 2)) * [639, 479]
    q = (H @ np.c_[src, np.ones(n)].T).T
    dst = q[:, :2] / q[:, 2:] + rng.normal(0, noise, (n, 2))
    out = rng.random(n) > inlier_share
    dst[out] = rng.random((int(out.sum()), 2)) * [639, 479]
    return src.astype(np.float32).astype(np.float64), dst.astype(np.float32).astype(np.float64), H / H[2, 2], ~out


@pytest.mark.parametrize("n,share,noise,tol", [(700, 0.7, 0.5, 0.5), (300, 0.4, 1.0, 1.5), (50, 0.5, 0.2, 0.6), (1000, 0.9, 0.0, 1e-3),
                                               (5, 1.0, 0.0, 1e-3), (4, 1.0, 0.0, 1e-3)])
def test_ransac_homography_recovers_ground_truth(n, share, noise, tol):
    src, dst, H, inl = synth(n, share, noise, n)
    He, mask, info = g.find_homography_ransac(src, dst, seed=3)
    assert He is not None and abs(He[2, 2] - 1) < 1e-12
    assert g.mha_corner_error(He, H, 480, 640, 480, 640) < tol          # the MHA metric itself: mean corner distance in pixels
    agree = (mask.astype(bool) == inl).mean()
    assert agree > 0.93 and info["inliers"] >= 0.8 * inl.sum()


def test_ransac_degenerate_inputs():
    src = np.stack([np.arange(40.0), 2 * np.arange(40.0) + 1], 1)          # all collinear: every sample is degenerate
    H, mask, info = g.find_homography_ransac(src, src + 5, seed=0)
    assert H is None and mask.sum() == 0 and info["iters"] == 0       # getSubset gives up in iteration 0: OpenCV returns false
    H, mask, _ = g.find_homography_ransac(src[:3], src[:3], seed=0)
    assert H is None
    # pure outliers: a "model" may be found but it explains (almost) nothing
    rng = np.random.default_rng(0)
    H, mask, info = g.find_homography_ransac(rng.random((200, 2)) * 640, rng.random((200, 2)) * 640, seed=0)
    assert info["inliers"] < 12


def test_cv_rng_and_get_subset_follow_opencvs_published_definitions():
    """cv::RNG: state = (unsigned)state * 4164903690 + (state >> 32), the low word is the draw; uniform(a, b) = next() % (b - a) + a.
    RANSAC seeds it with (uint64)-1 at every call.  getSubset: distinct indices, a duplicate draw is repeated at once; a subset that
    checkSubset rejects is redrawn whole.  (Restated from OpenCV 4.9's sources; no cv2 here to compare with: parity unpinned.)"""
    r = g.CvRNG(g.rng_state(0))
    assert r.state == 0xFFFFFFFFFFFFFFFF
    st = r.state
    draws = []
    for _ in range(5):
        st = ((st & 0xFFFFFFFF) * 4164903690 + (st >> 32)) & 0xFFFFFFFFFFFFFFFF
        draws.append(st & 0xFFFFFFFF)
    r2 = g.CvRNG(g.rng_state(0))
    assert [r2.next() for _ in range(5)] == draws
    assert draws[0] == (0xFFFFFFFF * 4164903690 + 0xFFFFFFFF) & 0xFFFFFFFF      # first draw from the all-ones state
    r3 = g.CvRNG(g.rng_state(0))
    u = [r3.uniform(0, 333) for _ in range(4000)]
    assert min(u) == 0 and max(u) == 332 and abs(np.mean(u) - 166) < 8
    assert g.CvRNG(0).state == 0xFFFFFFFF                                        # RNG(0) -> 0xffffffff, as OpenCV's constructor
    assert g.rng_state(5) != g.rng_state(6) and g.rng_state(5) != 0
    # getSubset: distinct indices; with n == m every subset is a permutation of 0..m-1
    rng = g.CvRNG(g.rng_state(0))
    pts = np.random.default_rng(0).random((50, 2)) * 100
    for _ in range(200):
        sub = g.get_subset(rng, 50, 4, None, pts, pts)
        assert len(set(sub)) == 4 and min(sub) >= 0 and max(sub) < 50
    for _ in range(50):
        assert sorted(g.get_subset(rng, 4, 4, None, pts, pts)) == [0, 1, 2, 3]
    # a rejecting checkSubset costs draws, not iterations: the stream of ACCEPTED subsets skips the rejected ones
    a, b = g.CvRNG(g.rng_state(3)), g.CvRNG(g.rng_state(3))
    acc = [g.get_subset(a, 50, 4, lambda s, d: int(round(s[0, 0] * 1000)) % 2 == 0, pts, pts) for _ in range(20)]
    alls = []
    while len([x for x in alls if int(round(pts[x[0], 0] * 1000)) % 2 == 0]) < 20:
        alls.append(g.get_subset(b, 50, 4, None, pts, pts))
    assert acc == [x for x in alls if int(round(pts[x[0], 0] * 1000)) % 2 == 0]


def test_have_collinear_tests_only_the_last_point():
    p = np.array([[0, 0], [1, 1], [2, 2], [5, 1]], np.float64)          # the first three are collinear, the last is not on their line
    assert not g.have_collinear_last(p)
    assert g.have_collinear_last(p[[0, 3, 1, 2]])                        # now the last one closes a collinear triple
    assert g.have_collinear_last(np.array([[0, 0], [3, 1], [0, 0]], np.float64))       # a repeated point counts as collinear
    sq = np.array([[0, 0], [10, 0], [10, 10], [0, 10]], np.float64)
    assert g.check_subset_homography(sq, sq * 2 + 1)
    assert not g.check_subset_homography(sq, sq[[0, 1, 3, 2]])           # a bow-tie: orientation of some triangles flips


def mha_params(prm, th):
    nms, border, top_k, maxd = prm
    return {"MHA_params": {"th": list(th)}, "extractor_params": dict(nms_dist=int(nms), threshold=0.0, border_dist=int(border), top_k=int(top_k), min_score=0.0),
            "matcher_params": {"type": "brute_force", "brute_force_params": dict(metric="euclidean", max_distance=float(maxd), cross_check=True)}}


@pytest.mark.parametrize("case", range(4))
def test_mha_chain_around_the_estimator_against_reference(case):
    f = load_golden("mha.npz")
    p = "c%d_" % case
    prm = mha_params(f[p + "prm"], f["th"])
    h, w = (int(v) for v in f[p + "hw"])
    real_H = f[p + "real_H"]
    k0, _ = oracle.detection(f[p + "score0"], prm["extractor_params"])
    k1, _ = oracle.detection(f[p + "score1"], prm["extractor_params"])
    inv = np.linalg.inv(real_H.astype(np.float64)).astype(np.float32)
    c0, _, _, _ = oracle.warp_homography(k0[:, :2], real_H, w, h)
    c1, _, _, _ = oracle.warp_homography(k1[:, :2], inv, w, h)
    m0, m1 = oracle.brute_force_matcher(c0, c1, f[p + "desc0"][0].astype(np.float32), f[p + "desc1"][0].astype(np.float32),
                                        prm["matcher_params"]["brute_force_params"])
    px = np.array([w - 1, h - 1], np.float32)
    assert np.array_equal(m0[:, :2] * px, f[p + "p0"]) and np.array_equal(m1[:, :2] * px, f[p + "p1"])      # MHA.py:40-44 bit for bit
    H, _, _ = g.find_homography_ransac(f[p + "p0"], f[p + "p1"], seed=0)
    np.testing.assert_allclose(H, f[p + "H"], rtol=0, atol=1e-12)
    Hs, Ws = f[p + "score0"].shape
    d = g.mha_corner_error(H, real_H, np.asarray(h), np.asarray(w), Hs, Ws)
    assert [float(d <= t) for t in f["th"]] == f[p + "flags"].tolist()
    # the product's host half (keypoint_bench_amd/tasks/MHA.py corner_hits) on the same numbers
    from keypoint_bench_amd.tasks.MHA import corner_hits
    hits, d2 = corner_hits(f[p + "H"], real_H, np.asarray(h), np.asarray(w), Hs, Ws, f["th"])
    assert hits == f[p + "flags"].tolist() and d2 == d


# ------------------------------------------------------------------------------------------------ essential matrix / AUC
def scene(n, inlier_share, noise_px, seed, f=500.0):
    rng = np.random.default_rng(seed)
    ax = rng.normal(0, 0.15, 3)
    th = np.linalg.norm(ax)
    k = ax / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = rng.normal(0, 1, 3)
    t /= np.linalg.norm(t)
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(3, 8, n)]
    x1 = X[:, :2] / X[:, 2:]
    Y = X @ R.T + t
    x2 = Y[:, :2] / Y[:, 2:]
    x1 = x1 + rng.normal(0, noise_px / f, x1.shape)
    x2 = x2 + rng.normal(0, noise_px / f, x2.shape)
    out = rng.random(n) > inlier_share
    x2[out] = rng.uniform(-0.5, 0.5, (int(out.sum()), 2))
    return x1, x2, R, t, ~out


def test_five_point_solver_contains_the_true_essential_matrix():
    x1, x2, R, t, _ = scene(400, 1.0, 0.0, 11)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Egt = tx @ R
    Egt /= np.linalg.norm(Egt)
    rng = np.random.default_rng(3)
    idx = np.stack([rng.choice(400, 5, replace=False) for _ in range(200)])
    E, valid = g.essential_5pt(x1[idx], x2[idx])
    d = np.minimum(np.abs(E - Egt).max((2, 3)), np.abs(E + Egt).max((2, 3)))
    best = np.where(valid, d, 9.0).min(1)
    assert (best < 1e-6).mean() > 0.7 and np.median(best) < 1e-8          # the rest are ill-conditioned samples: RANSAC's business
    # every returned candidate is an essential matrix: two equal singular values and a zero one, and it fits its sample
    sv = np.linalg.svd(E[valid], compute_uv=False)
    ok = (np.abs(sv[:, 0] - sv[:, 1]) < 1e-5) & (sv[:, 2] < 1e-5)
    assert ok.mean() > 0.85
    resid = np.abs(np.einsum("tsij,tnj,tni->tsn", E, np.concatenate([x1[idx], np.ones((200, 5, 1))], 2), np.concatenate([x2[idx], np.ones((200, 5, 1))], 2)))
    assert np.median(resid.max(2)[valid]) < 1e-9


def test_aberth_roots_equal_numpy_roots():
    rng = np.random.default_rng(1)
    c = rng.normal(size=(300, 11)) * np.exp(rng.normal(0, 2, (300, 11)))
    r = g.aberth_roots(c)
    for i in range(300):
        ref = np.roots(c[i, ::-1])
        d = np.abs(ref[:, None] - r[i][None, :]).min(1) / (1 + np.abs(ref))
        assert d.max() < 1e-9, i


@pytest.mark.parametrize("n,share,noise,tol_t,tol_R", [(800, 0.7, 0.5, 4.0, 1.5), (300, 0.5, 0.5, 5.0, 2.0), (1000, 0.9, 0.0, 3.0, 1.0), (6, 1.0, 0.0, 1e-3, 1e-3)])
def test_essential_ransac_and_recover_pose_on_ground_truth(n, share, noise, tol_t, tol_R):
    x1, x2, R, t, inl = scene(n, share, noise, n)
    E, mask, info = g.find_essential_ransac(x1, x2, seed=0, threshold=1.0 / 500)      # seed 0: the state OpenCV starts every call with
    assert E is not None and abs(np.linalg.norm(E) - 1) < 1e-12
    nn, Rr, tt, mnew = g.recover_pose(E, x1, x2, mask)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    et, eR = g.compute_pose_error(T, Rr, tt)
    assert et < tol_t and eR < tol_R, (et, eR)          # plain RANSAC, the winning five-point model unrefined (as cv2): degrees, not arc seconds
    assert ((mask > 0) == inl).mean() > 0.8 and nn >= 0.9 * mask.sum() and abs(np.linalg.det(Rr) - 1) < 1e-9


def auc_params():
    return {"extractor_params": dict(nms_dist=2, threshold=0.0, border_dist=4, top_k=1000, min_score=0.0),
            "matcher_params": {"type": "brute_force", "brute_force_params": dict(metric="euclidean", max_distance=1.0, cross_check=True)},
            "AUC_params": {"th": [5, 10, 20]}}


def test_auc_helpers_against_reference_known_answers():
    f = load_golden("auc.npz")
    from keypoint_bench_amd.tasks import AUC as prod
    from keypoint_bench_amd.runner import pose_auc
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = f["h_Rb"], f["h_vb"]
    for mod in (g, prod):
        assert mod.angle_error_mat(f["h_Ra"], f["h_Rb"]) == f["h_angle_mat"]
        assert mod.angle_error_vec(f["h_va"], f["h_vb"]) == f["h_angle_vec"]
        assert np.array_equal(np.array(mod.compute_pose_error(T, f["h_Ra"], f["h_va"])), f["h_pose_err"])
    np.testing.assert_allclose(pose_auc(f["h_errs"], [5, 10, 20]), f["h_pose_auc"], rtol=1e-14)


@pytest.mark.parametrize("case", range(4))
def test_auc_chain_around_the_estimator_against_reference(case):
    f = load_golden("auc.npz")
    p = "c%d_" % case
    prm = auc_params()
    k0, _ = oracle.detection(f[p + "score0"], prm["extractor_params"])
    k1, _ = oracle.detection(f[p + "score1"], prm["extractor_params"])
    m0, m1 = oracle.brute_force_matcher(k0, k1, f[p + "desc0"][0].astype(np.float32), f[p + "desc1"][0].astype(np.float32), prm["matcher_params"]["brute_force_params"])
    H, W = f[p + "score0"].shape
    K = f[p + "K"]                         # float32, as datasets/megadepth.py:341-342 hands it over: numpy normalises in float32
    want = f[p + "result"]
    if len(m0) < 5:
        assert want.tolist() == [180.0, 0.0]
        return
    px0 = m0[:, :2] * np.array([W - 1, H - 1], np.float32)                 # AUC.py:125-126 (fp32 products)
    px1 = m1[:, :2] * np.array([W - 1, H - 1], np.float32)
    n0 = (px0 - K[[0, 1], [2, 2]][None]) / K[[0, 1], [0, 1]][None]        # 47-48
    n1 = (px1 - K[[0, 1], [2, 2]][None]) / K[[0, 1], [0, 1]][None]
    assert np.array_equal(n0, f[p + "k0"]) and np.array_equal(n1, f[p + "k1"])
    assert f[p + "thr"] == 1.0 / np.mean([K[0, 0], K[1, 1], K[0, 0], K[1, 1]])
    res = g.estimate_pose(px0, px1, K, K, 1.0, seed=0)
    R, t, inl = res
    np.testing.assert_allclose(R, f[p + "R"], atol=1e-12)
    np.testing.assert_allclose(t, f[p + "t"], atol=1e-12)
    et, eR = g.compute_pose_error(f[p + "T01"], R, t)            # float32 pose, as the dataset hands it over (norms of t_gt in float32)
    assert [max(et, eR), float(inl.sum())] == want.tolist()


# ------------------------------------------------------------------------------------------------ fundamental matrix (7 points)
def fscene(n, share, noise, seed, f=500.0):
    """Pixel correspondences of a two-view scene, the true F (F[2,2] = 1) and the inlier flags."""
    x1, x2, R, t, inl = scene(n, share, noise, seed, f)
    c = np.array([319.5, 239.5])
    p1 = (x1 * f + c).astype(np.float32).astype(np.float64)
    p2 = (x2 * f + c).astype(np.float32).astype(np.float64)
    Kc = np.array([[f, 0, c[0]], [0, f, c[1]], [0, 0, 1.0]])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(Kc).T @ tx @ R @ np.linalg.inv(Kc)
    return p1, p2, F / F[2, 2], inl


def test_seven_point_solver_contains_the_true_fundamental_matrix():
    p1, p2, F, _ = fscene(300, 1.0, 0.0, 21)
    rng = np.random.default_rng(0)
    idx = np.stack([rng.choice(300, 7, replace=False) for _ in range(64)])
    Fs, valid = g.fundamental_7pt(p1[idx], p2[idx])
    assert valid.any(1).all() and (valid.sum(1) >= 1).all() and set(valid.sum(1).tolist()) <= {1, 3}
    # every model satisfies the seven constraints and is singular; one of them is the scene's matrix
    for s in range(64):
        best = np.inf
        for k in range(3):
            if not valid[s, k]:
                continue
            Fk = Fs[s, k]
            res = np.einsum("ni,ij,nj->n", np.c_[p2[idx[s]], np.ones(7)], Fk, np.c_[p1[idx[s]], np.ones(7)])
            assert np.abs(res).max() < 1e-6 * np.abs(Fk).max() * 640 * 640
            assert abs(np.linalg.det(Fk / np.linalg.norm(Fk))) < 1e-9
            best = min(best, np.abs(Fk - F).max() / np.abs(F).max())
        assert best < 1e-3, (s, best)                     # float32-rounded pixels: the exact matrix up to that rounding


def test_cubic_roots_against_numpy():
    rng = np.random.default_rng(1)
    c = rng.normal(size=(500, 4))
    r, v = g.solve_cubic(c[:, 0], c[:, 1], c[:, 2], c[:, 3])
    for i in range(500):
        want = np.roots(c[i])
        want = np.sort(want[np.abs(want.imag) < 1e-7].real)
        np.testing.assert_allclose(np.sort(r[i][v[i]]), want, rtol=1e-7, atol=1e-9)
    r, v = g.solve_cubic(np.zeros(3), np.ones(3), np.ones(3), np.ones(3))
    assert not v.any()                                    # a vanishing leading coefficient voids the sample


@pytest.mark.parametrize("n,share,noise", [(800, 0.7, 0.4), (300, 0.5, 0.5), (1000, 0.9, 0.0), (60, 0.6, 0.3), (8, 1.0, 0.0)])
def test_ransac_fundamental_recovers_ground_truth(n, share, noise):
    p1, p2, F, inl = fscene(n, share, noise, 40 + n)
    Fe, mask, info = g.find_fundamental_ransac(p1, p2, seed=9)
    assert Fe is not None and abs(Fe[2, 2] - 1) < 1e-12 and info["inliers"] == mask.sum()
    # true correspondences lie on their epipolar lines of the estimate (a minimal 7-point model: a few pixels at most)
    err = g.fm_error(Fe, p1[inl], p2[inl])
    assert np.median(np.sqrt(err)) < max(2.5 * noise, 1e-3) + 0.5
    assert mask[inl].mean() > 0.85 and mask[~inl].mean() < 0.2 if (~inl).any() else True


def test_ransac_fundamental_small_and_degenerate_inputs():
    p1, p2, _, _ = fscene(7, 1.0, 0.0, 3)
    F, mask, info = g.find_fundamental_ransac(p1, p2, seed=0)          # below 8: utils/mvg.py never calls cv2
    assert F is None and mask.sum() == 0 and info["iters"] == 0
    line = np.stack([np.arange(30.0) * 7, np.arange(30.0) * 3 + 5], 1)  # collinear in both images: every sample is void
    F, mask, info = g.find_fundamental_ransac(line, line + 2, seed=0)
    assert F is None and info["iters"] == 0          # getSubset gives up in iteration 0: OpenCV returns false
    a, b, c = g.fundamental_estimate(p1, p2)
    assert a is None and len(b) == 7 and len(c) == 7
    with pytest.raises(AttributeError):
        g.fundamental_estimate(line, line + 2)


def test_fundamental_below_15_points_takes_the_least_median_branch():
    """cv::findFundamentalMat(FM_RANSAC) with 8 <= n < 15 points runs OpenCV's LMedS registrator (300 iterations at the defaults).
    With 7-point samples on so few points the criterion is weak (the sample itself supplies 7 zero errors), so what is checked is the
    rule itself: the iteration count, the kept model's median against every other hypothesis of the stream, and the sigma rule."""
    assert g.update_iters(0.99, g.LMEDS_OUTLIER_RATIO, 7, 1000) == 300
    for n in (9, 12, 14):
        p1, p2, F, _ = fscene(n, 0.8, 0.5, 40 + n)
        Fe, mask, info = g.find_fundamental_ransac(p1, p2, seed=0)
        assert Fe is not None and info["iters"] == 300
        err = g.fm_error(Fe, p1, p2)
        med = float(np.sort(err)[n // 2])
        sigma = max(2.5 * 1.4826 * (1 + 5.0 / (n - 7)) * np.sqrt(med), 0.001)
        assert np.array_equal(mask, (err <= np.float32(sigma * sigma)).astype(np.uint8)) and mask.sum() == info["inliers"] >= 7
        # no hypothesis of the same stream has a smaller median
        rng = g.CvRNG(g.rng_state(0))
        for _ in range(300):
            idx = g.get_subset(rng, n, 7, g.check_subset_fundamental, p1, p2, g.LMEDS_ATTEMPTS)
            Fs, valid = g.fundamental_7pt(p1[idx][None], p2[idx][None])
            for k in range(3):
                if valid[0, k]:
                    assert float(np.sort(g.fm_error(Fs[0, k], p1, p2))[n // 2]) >= med
    p1, p2, F, _ = fscene(15, 1.0, 0.0, 77)
    _, _, info = g.find_fundamental_ransac(p1, p2, seed=0)
    assert info["iters"] < 300          # 15 points: RANSAC, which stops as soon as every point is an inlier
