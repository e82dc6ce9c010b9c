"""GPU parity of SURVEY 8(f) rank 1 (csrc/covis.hip through the C ABI): warp_homography and val_key_points against the
fixtures the reference produced (tests/golden/covis.npz) and against the oracle.

Bit-exact bar.  One documented exception: the reference's einsum (projection.py:146) runs torch's unfused small-matrix
loop when it has fewer than 45 points and its BLAS (fused multiply-add) from 45 up; libkpb always evaluates the fused
form, so reference fixtures with n < 45 are compared within 2 ulp and the oracle's fused form bit-exactly."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "covis.npz"))
CASES = range(int(G["n_cases"]))


def warps(c, device=None):
    p = "c%d_" % c
    w0, h0 = G[p + "wh0"]; w1, h1 = G[p + "wh1"]
    conv = (lambda v: torch.as_tensor(v).to(device)) if device else (lambda v: v)
    w01 = dict(mode="homo", homography_matrix=conv(G[p + "hm"]), width=conv(int(w1)), height=conv(int(h1)))
    w10 = dict(mode="homo", homography_matrix=conv(G[p + "hinv"]), width=conv(int(w0)), height=conv(int(h0)))
    if int(G[p + "resize"]):
        w01["resize"] = w10["resize"] = conv(int(G[p + "resize"]))
    return p, w01, w10


@pytest.mark.parametrize("c", CASES)
def test_warp_homography(c):
    from keypoint_bench_amd.utils.projection import warp
    p, w01, w10 = warps(c, DEV)
    _, o01, o10 = warps(c)
    for kps, w, wo, names in ((G[p + "kps0"], w01, o01, ("k0v", "k01v", "ids", "ids_out")),
                              (G[p + "kps1"], w10, o10, ("k1v", "k10v", "ids1", "ids1_out"))):
        a, b, ids, ids_out = warp(torch.from_numpy(kps).to(DEV), w)
        assert ids.dtype == torch.int64 and a.shape == b.shape == (len(ids), 2)
        ea, eb, eids, eout = oracle.warp_homography(kps[:, :2], wo["homography_matrix"], wo["width"], wo["height"], fused=1)
        np.testing.assert_array_equal(ids.cpu().numpy(), eids)
        np.testing.assert_array_equal(ids_out.cpu().numpy(), eout)
        np.testing.assert_array_equal(a.cpu().numpy(), ea)
        np.testing.assert_array_equal(b.cpu().numpy(), eb)
        # the reference itself
        np.testing.assert_array_equal(ids.cpu().numpy(), G[p + names[2]])
        np.testing.assert_array_equal(ids_out.cpu().numpy(), G[p + names[3]])
        np.testing.assert_array_equal(a.cpu().numpy(), G[p + names[0]])
        if len(kps) >= 45:
            np.testing.assert_array_equal(b.cpu().numpy(), G[p + names[1]])
        else:
            np.testing.assert_allclose(b.cpu().numpy(), G[p + names[1]], rtol=2.4e-7, atol=0)


@pytest.mark.parametrize("c", CASES)
def test_val_key_points(c, covis_form):
    from keypoint_bench_amd.tasks.repeatability import val_key_points
    p, w01, w10 = warps(c, DEV)
    _, o01, o10 = warps(c)
    k0, k1 = G[p + "kps0"], G[p + "kps1"]
    got = val_key_points(torch.from_numpy(k0).to(DEV), torch.from_numpy(k1).to(DEV), w01, w10, th=3)
    exp = oracle.val_key_points(k0, k1, o01, o10, th=3, fused=1)
    assert got["num_feat"] == exp["num_feat"] == int(G[p + "num_feat"])
    np.testing.assert_array_equal(got["errors"].cpu().numpy(), exp["errors"])
    assert np.float32(got["repeatability"]) == np.float32(exp["repeatability"])
    np.testing.assert_array_equal(np.float32(got["mean_error"]), np.float32(exp["mean_error"]))
    if min(len(k0), len(k1)) >= 45:      # the reference ran the fused form on both sets
        np.testing.assert_array_equal(got["errors"].cpu().numpy(), G[p + "errors"])
        assert np.float32(got["repeatability"]) == G[p + "repeatability"]
        np.testing.assert_array_equal(np.float32(got["mean_error"]), G[p + "mean_error"])
    else:
        np.testing.assert_allclose(got["errors"].cpu().numpy(), G[p + "errors"], rtol=1e-5)
        assert np.float32(got["repeatability"]) == G[p + "repeatability"]


@pytest.fixture(params=["cells stored", "cells evaluated in place"])
def covis_form(request):
    """kpb_val_keypoints has two forms: the M x N cells written once by the row pass (the default when they fit
    KPB_OPT_COVIS_STORE_BYTES, 4 GiB) and every pass evaluating its own cells (beyond that, or when the workspace cannot grow).
    Default shapes only ever reach the first: the limit is an option of the context (include/kpb.h), so the
    second runs here on the same fixtures with the limit at 0 -- and must give the same bits."""
    from keypoint_bench_amd._lib import Context
    ctx = Context.get(torch.device(DEV))
    ctx.set_option(Context.OPT_COVIS_STORE_BYTES, (4 << 30) if request.param == "cells stored" else 0)
    yield request.param
    ctx.set_option(Context.OPT_COVIS_STORE_BYTES, 4 << 30)


def test_context_options_refuse_what_they_do_not_know():
    from keypoint_bench_amd._lib import Context, KpbError
    ctx = Context.get(torch.device(DEV))
    for opt, val in ((99, 1), (Context.OPT_COVIS_STORE_BYTES, -1)):
        with pytest.raises(KpbError) as e:
            ctx.set_option(opt, val)
        assert e.value.code == -1       # KPB_E_INVALID


@pytest.mark.parametrize("c", CASES)
def test_gt_mutual_pairs(c, covis_form):
    from keypoint_bench_amd.tasks.repeatability import gt_mutual
    p = "c%d_" % c
    t = lambda k: torch.from_numpy(G[p + k]).to(DEV)
    s = float(G[p + "resize"]) or float(G[p + "wh1"][0])
    s10 = float(G[p + "resize"]) or float(G[p + "wh0"][0])
    # the reference's own covisible sets in, so this checks lines 69-85 alone: bit-exact for every case
    pairs, dist, errors, gt = gt_mutual(t("k0v"), t("k01v"), t("k1v"), t("k10v"), s, s10, th=3.0, cap=8)   # cap=8 forces the regrow path
    np.testing.assert_array_equal(pairs.cpu().numpy(), G[p + "pairs"])
    np.testing.assert_array_equal(dist.cpu().numpy(), G[p + "dist"])
    np.testing.assert_array_equal(errors.cpu().numpy(), G[p + "errors"])
    assert gt == int((G[p + "dist"] <= 3).sum())


def test_device_counts_and_batch_through_the_abi(covis_form):
    """Two pairs in one call, ragged by device-side counts, against the single-pair results."""
    from keypoint_bench_amd._lib import Context, ptr
    ctx = Context.get(torch.device(DEV))
    cs = (1, 2)
    mm = max(len(G["c%d_k0v" % c]) for c in cs); nn = max(len(G["c%d_k1v" % c]) for c in cs)
    k0 = torch.zeros((2, mm, 2), device=DEV); k01 = torch.zeros_like(k0)
    k1 = torch.zeros((2, nn, 2), device=DEV); k10 = torch.zeros_like(k1)
    m = torch.zeros(2, dtype=torch.int32, device=DEV); n = torch.zeros_like(m)
    for b, c in enumerate(cs):
        p = "c%d_" % c
        M, N = len(G[p + "k0v"]), len(G[p + "k1v"])
        k0[b, :M] = torch.from_numpy(G[p + "k0v"]); k01[b, :M] = torch.from_numpy(G[p + "k01v"])
        k1[b, :N] = torch.from_numpy(G[p + "k1v"]); k10[b, :N] = torch.from_numpy(G[p + "k10v"])
        m[b], n[b] = M, N
    cap = 4096
    scale = torch.full((2, 2), 512.0, device=DEV)
    pairs = torch.full((2, cap, 2), -1, dtype=torch.int32, device=DEV); dist = torch.zeros((2, cap), device=DEV)
    errors = torch.zeros((2, mm), device=DEV); counts = torch.zeros((2, 2), dtype=torch.int32, device=DEV)
    ctx.check(ctx.lib.kpb_val_keypoints(ctx.handle, ptr(k0), ptr(k01), ptr(k1), ptr(k10), 2, mm, nn, ptr(m), ptr(n), ptr(scale),
                                        3.0, ptr(pairs), ptr(dist), cap, ptr(errors), ptr(counts)))
    for b, c in enumerate(cs):
        p = "c%d_" % c
        K = len(G[p + "pairs"])
        assert counts[b].tolist() == [K, int((G[p + "dist"] <= 3).sum())]
        np.testing.assert_array_equal(pairs[b, :K].cpu().numpy(), G[p + "pairs"])
        np.testing.assert_array_equal(dist[b, :K].cpu().numpy(), G[p + "dist"])
        np.testing.assert_array_equal(errors[b, : int(m[b])].cpu().numpy(), G[p + "errors"])


def test_warp_batch_with_device_counts():
    """Two images in one kpb_warp_homography call, ragged by device-side counts, against the single-image results."""
    from keypoint_bench_amd._lib import Context, ptr
    ctx = Context.get(torch.device(DEV))
    cs = (2, 1)
    n_max = max(len(G["c%d_kps0" % c]) for c in cs)
    kps = torch.zeros((2, n_max, 3), device=DEV)
    n = torch.zeros(2, dtype=torch.int32, device=DEV)
    hm = torch.zeros((2, 9), device=DEV); wh = torch.zeros((2, 2), dtype=torch.int32, device=DEV)
    for b, c in enumerate(cs):
        p = "c%d_" % c
        k = torch.from_numpy(G[p + "kps0"])
        kps[b, : len(k)] = k; n[b] = len(k)
        hm[b] = torch.from_numpy(G[p + "hm"]).reshape(9); wh[b] = torch.from_numpy(G[p + "wh1"].astype(np.int32))
    a = torch.zeros((2, n_max, 2), device=DEV); bb = torch.zeros_like(a)
    ids = torch.full((2, n_max), -1, dtype=torch.int32, device=DEV); nv = torch.zeros(2, dtype=torch.int32, device=DEV)
    ctx.check(ctx.lib.kpb_warp_homography(ctx.handle, ptr(kps), 2, n_max, 3, ptr(n), ptr(hm), ptr(wh), ptr(a), ptr(bb), ptr(ids), ptr(nv)))
    for b, c in enumerate(cs):
        p = "c%d_" % c
        k = int(nv[b])
        assert k == len(G[p + "ids"])
        np.testing.assert_array_equal(ids[b, :k].cpu().numpy(), G[p + "ids"])
        np.testing.assert_array_equal(ids[b, k: int(n[b])].cpu().numpy(), G[p + "ids_out"])
        np.testing.assert_array_equal(a[b, :k].cpu().numpy(), G[p + "k0v"])
        np.testing.assert_array_equal(bb[b, :k].cpu().numpy(), G[p + "k01v"])


def test_full_size_properties():
    """8192 keypoints per side: identity keeps everything in order; a warp followed by its inverse returns the points;
    the number of mutual cells is symmetric under swapping the two images."""
    from keypoint_bench_amd.utils.projection import warp
    from keypoint_bench_amd.tasks.repeatability import gt_mutual
    rng = np.random.default_rng(5)
    n = 8192
    k = torch.from_numpy(rng.random((n, 3)).astype(np.float32)).to(DEV)
    ident = dict(mode="homo", homography_matrix=torch.eye(3, device=DEV), width=640, height=480)
    a, b, ids, out = warp(k, ident)
    assert len(out) == 0 and torch.equal(ids, torch.arange(n, device=DEV))
    assert torch.allclose(a, k[:, :2], atol=1e-6) and torch.allclose(a, b, atol=1e-6)
    hm = torch.tensor([[0.95, 0.04, 12.0], [-0.03, 1.02, -7.0], [2e-5, -1e-5, 1.0]], device=DEV)
    fwd = dict(mode="homo", homography_matrix=hm, width=640, height=480)
    bwd = dict(mode="homo", homography_matrix=torch.linalg.inv(hm.double()).float(), width=640, height=480)
    a, b, ids, out = warp(k, fwd)
    assert len(ids) + len(out) == n and len(torch.unique(torch.cat([ids, out]))) == n
    a2, b2, ids2, _ = warp(b, bwd)
    assert len(ids2) >= len(ids) - 8          # points on the frame may fall out by rounding
    assert torch.allclose(b2, a[ids2], atol=2e-5)
    k1 = torch.from_numpy(rng.random((n, 2)).astype(np.float32)).to(DEV)
    a1, b1, _, _ = warp(k1, bwd)
    p01, d01, e01, g01 = gt_mutual(a, b, a1, b1, 512.0, 512.0)
    p10, d10, e10, g10 = gt_mutual(a1, b1, a, b, 512.0, 512.0)
    assert len(p01) == len(p10) and g01 == g10
    swapped = p10[:, [1, 0]]
    order = torch.argsort(swapped[:, 0] * 100000 + swapped[:, 1])
    assert torch.equal(swapped[order], p01)


def test_stored_and_in_place_forms_agree_bit_for_bit_at_full_size():
    """2 048 x 2 048 cells (beyond every fixture's size), ties included (keypoints on a coarse grid): pairs, distances, errors and the
    count within th of the evaluate-in-place form equal those of the stored-cells form."""
    from keypoint_bench_amd._lib import Context
    from keypoint_bench_amd.utils.projection import warp
    from keypoint_bench_amd.tasks.repeatability import gt_mutual
    ctx = Context.get(torch.device(DEV))
    rng = np.random.default_rng(11)
    n = 2048
    grid = lambda: torch.from_numpy((rng.integers(0, 200, (n, 2)).astype(np.float32) + 0.5) / 200.0).to(DEV)      # many exact ties
    hm = torch.tensor([[1.0, 0.0, 3.0], [0.0, 1.0, -2.0], [0.0, 0.0, 1.0]], device=DEV)
    fwd = dict(mode="homo", homography_matrix=hm, width=640, height=480)
    bwd = dict(mode="homo", homography_matrix=torch.linalg.inv(hm.double()).float(), width=640, height=480)
    a, b, _, _ = warp(grid(), fwd)
    a1, b1, _, _ = warp(grid(), bwd)
    out = []
    try:
        for limit in (4 << 30, 0):
            ctx.set_option(Context.OPT_COVIS_STORE_BYTES, limit)
            out.append(gt_mutual(a, b, a1, b1, 512.0, 512.0))
    finally:
        ctx.set_option(Context.OPT_COVIS_STORE_BYTES, 4 << 30)
    (p0, d0, e0, g0), (p1, d1, e1, g1) = out
    assert len(p0) > 100 and g0 == g1
    assert torch.equal(p0, p1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32)) and torch.equal(e0.view(torch.int32), e1.view(torch.int32))
