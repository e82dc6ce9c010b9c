"""GPU parity of A1..A5 (csrc/detect.hip through the C ABI) against the oracle and the reference goldens."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, params_from, assert_kps_equal
from keypoint_bench_amd import synthetic

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def test_library_loaded_and_version():
    from keypoint_bench_amd import _lib
    assert _lib.load().kpb_version() == 1


def test_detection_golden_small():
    from keypoint_bench_amd.utils.extracter import detection
    g = load_golden("det_small.npz")
    for name in g["cases"]:
        p = params_from(g[name + ".params"])
        s = torch.from_numpy(g[name + ".score"])[None, None].to(_dev())
        before = s.clone()
        kps = detection(s, p).cpu().numpy()
        assert torch.equal(s, before), "detection must not mutate its input"
        assert_kps_equal(kps, g[name + ".kps"], p["top_k"], name)
        okps, _ = oracle.detection(g[name + ".score"], p)
        np.testing.assert_array_equal(kps.view(np.uint32), okps.view(np.uint32), err_msg=name)  # same tie rule


def test_fast_nms_maps_golden():
    from keypoint_bench_amd.utils.extracter import fast_nms
    g = load_golden("det_small.npz")
    for name in g["cases"]:
        if name + ".nms" not in g.files:
            continue
        p = params_from(g[name + ".params"])
        s = torch.from_numpy(g[name + ".score"])[None, None].to(_dev())
        out = fast_nms(s, nms_dist=p["nms_dist"])[0, 0].cpu().numpy()
        np.testing.assert_array_equal(out.view(np.uint32), g[name + ".nms"].view(np.uint32), err_msg=name)


def test_detection_default_params_none():
    from keypoint_bench_amd.utils.extracter import detection
    g = load_golden("det_small.npz")
    s = torch.from_numpy(g["defaults_none.score"])[None, None].to(_dev())
    kps = detection(s, None).cpu().numpy()
    assert_kps_equal(kps, g["defaults_none.kps"], 300, "defaults")


def test_detection_full_size_golden_and_batched():
    from keypoint_bench_amd.utils.extracter import detection, detection_batch
    g = load_golden("det_full.npz")
    gens = dict(uniform=synthetic.score_uniform, smooth=synthetic.score_smooth)
    maps, wants = [], []
    for name in g["cases"]:
        fam, seed = g[name + ".gen"]
        smap = gens[str(fam)](int(seed), 480, 640)
        p = params_from(g[name + ".params"])
        kps = detection(torch.from_numpy(smap)[None, None].to(_dev()), p).cpu().numpy()
        assert_kps_equal(kps, g[name + ".kps"], p["top_k"], name)
        if p["nms_dist"] == 6:
            maps.append(smap)
            wants.append(g[name + ".kps"])
    # the batch entry point must give the same rows per image
    p = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
    kps, idx, n = detection_batch(torch.from_numpy(np.stack(maps))[:, None].to(_dev()), p)
    for b, want in enumerate(wants):
        nb = int(n[b])
        assert_kps_equal(kps[b, :nb].cpu().numpy(), want, 1000, "batch %d" % b)
        ii = idx[b, :nb].cpu().numpy()
        np.testing.assert_array_equal(kps[b, :nb, 2].cpu().numpy(), maps[b].ravel()[ii])


@pytest.mark.parametrize("top_k,thr", [(1000, 0.0), (100000, 0.0), (50, 0.9)])
def test_selection_forms_agree_one_workgroup_per_image_and_two_phase(top_k, thr):
    """Below 64 images the raster-order candidate scan runs as (chunks x images) workgroups (select_scan) and one workgroup per image
    only closes the chunk lists up and selects; from 64 images on that workgroup scans the map itself.  The same 72 maps through
    both forms (one call of 72, calls of 9): identical rows, counts and flat indices; N <= top_k (raster order) and N > top_k."""
    from keypoint_bench_amd.utils.extracter import detection_batch
    H, W = 96, 416                                          # 39 936 pixels: three chunks of 16 384, the last one short
    maps = np.stack([synthetic.score_uniform(700 + i, H, W) if i % 3 else synthetic.score_smooth(700 + i, H, W) for i in range(72)])
    p = dict(nms_dist=3, threshold=thr, border_dist=5, top_k=top_k, min_score=0.0)
    t = torch.from_numpy(maps)[:, None].to(_dev())
    k1, i1, n1 = detection_batch(t, p)                      # >= 64 images: one workgroup per image
    for b0 in range(0, 72, 9):
        k2, i2, n2 = detection_batch(t[b0:b0 + 9], p)       # < 64: two-phase
        assert torch.equal(n1[b0:b0 + 9], n2)
        for b in range(9):
            nb = int(n2[b])
            assert nb > 0
            assert torch.equal(k1[b0 + b, :nb], k2[b, :nb]) and torch.equal(i1[b0 + b, :nb], i2[b, :nb])
    import oracle
    for b in (0, 1, 71):
        want, _ = oracle.detection(maps[b], p)
        assert_kps_equal(k1[b, : int(n1[b])].cpu().numpy(), want, min(top_k, H * W), "image %d" % b)


def test_detection_on_reference_alike_score_maps():
    from keypoint_bench_amd.utils.extracter import detection
    g = load_golden("alike_t.npz")
    p = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
    for s, k in (("full.score0", "full.kps0"), ("full.score1", "full.kps1")):
        kps = detection(torch.from_numpy(g[s])[None, None].to(_dev()), p).cpu().numpy()
        assert_kps_equal(kps, g[k], 1000, s)


@pytest.mark.parametrize("nms", [1, 3, 6, 11, 16])
def test_detection_random_vs_oracle(nms):
    from keypoint_bench_amd.utils.extracter import detection
    rng = np.random.default_rng(1000 + nms)
    for (H, W, topk, border, thr, ms) in ((70, 130, 64, 5, 0.0, 0.0), (33, 65, 500, 0, 0.3, 0.0), (128, 96, 17, 9, 0.0, 0.5)):
        for quant in (0, 16):
            s = rng.random((H, W), dtype=np.float32)
            if quant:
                s = np.floor(s * quant).astype(np.float32) / quant
            p = dict(nms_dist=nms, threshold=thr, border_dist=border, top_k=topk, min_score=ms)
            want, _ = oracle.detection(s, p)
            got = detection(torch.from_numpy(s)[None, None].to(_dev()), p).cpu().numpy()
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32),
                                          err_msg="nms=%d %dx%d quant=%d" % (nms, H, W, quant))


def test_negative_scores_fail_loudly():
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd._lib import KpbError
    s = torch.rand(1, 1, 64, 64, device=_dev()) - 0.5
    with pytest.raises(KpbError):
        detection(s, dict(nms_dist=4, threshold=0.0, border_dist=4, top_k=100, min_score=0.0))


def test_full_size_properties():
    """Size-independent properties at the BASELINE size: survivors are >= nms_dist+1 apart, every
    survivor is a local maximum of the input, and detection is idempotent on its own NMS map."""
    from keypoint_bench_amd.utils.extracter import detection_batch, fast_nms
    s = torch.from_numpy(np.stack([synthetic.score_uniform(77, 480, 640), synthetic.score_smooth(78, 480, 640)]))[:, None].to(_dev())
    m = fast_nms(s, nms_dist=6)
    m2 = fast_nms(m, nms_dist=6)
    assert torch.equal(m, m2)
    alive = (m > 0)
    pooled = torch.nn.functional.max_pool2d(alive.float(), 13, 1, 6) * 0 + torch.nn.functional.avg_pool2d(alive.float(), 13, 1, 6, divisor_override=1)
    assert float((pooled * alive).max()) == 1.0          # exactly one survivor in each survivor's window
    wmax = torch.nn.functional.max_pool2d(s, 13, 1, 6)
    assert bool(((m == s) | (m == 0)).all())
    kept_vals = m[alive]
    assert bool((kept_vals <= wmax[alive]).all())


@pytest.mark.parametrize("mode", ["tiled-only", "tail-cut-short", "tail-list-overflow"])
def test_nms_schedules_agree(mode, monkeypatch):
    """The sparse tail, the tiled sweeps alone, and the hand-over from a tail that gives up (round limit) or whose list
    overflowed (heavy ties) must all end in the same fixed point (the rules are monotone: any schedule does)."""
    from keypoint_bench_amd.utils.extracter import fast_nms
    if mode == "tail-list-overflow":
        rng = np.random.default_rng(3)
        m = (np.floor(rng.random((2, 1, 480, 640)) * 4) / 4 + 0.25).astype(np.float32)     # plateaus: almost nothing settles in sweep 0
    else:
        m = np.stack([synthetic.score_smooth(21, 480, 640), synthetic.score_uniform(22, 480, 640)])[:, None]
    s = torch.from_numpy(m).to(_dev())
    monkeypatch.setenv("KPB_NMS_TILED", "0")            # small batches default to the tiled sweeps: force the tail
    want = fast_nms(s, 6).cpu().numpy()
    if mode == "tiled-only":
        monkeypatch.setenv("KPB_NMS_TILED", "1")
    elif mode == "tail-cut-short":
        monkeypatch.setenv("KPB_NMS_TAIL_ROUNDS", "1")
    got = fast_nms(s, 6).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    for b in range(2):
        exp, _ = oracle.fast_nms(m[b, 0], 6)
        np.testing.assert_array_equal(got[b, 0], exp)


@pytest.mark.parametrize("r", [1, 2, 3, 5, 7, 8])
def test_nms_tail_every_radius(r, monkeypatch):
    """The sparse tail is specialised per radius (r = 8 needs a 17th window column): force it on small batches and check
    every radius against the oracle's literal rounds."""
    from keypoint_bench_amd.utils.extracter import fast_nms
    m = np.stack([synthetic.score_smooth(30 + r, 240, 320), synthetic.score_uniform(40 + r, 240, 320)])[:, None]
    monkeypatch.setenv("KPB_NMS_TILED", "0")
    got = fast_nms(torch.from_numpy(m).to(_dev()), r).cpu().numpy()
    for b in range(2):
        exp, _ = oracle.fast_nms(m[b, 0], r)
        np.testing.assert_array_equal(got[b, 0], exp)


@pytest.mark.parametrize("prune", ["1", "0"])
def test_topk_pruned_tail_is_exact(prune, monkeypatch):
    """nms_tail drops undecided pixels that score below the (top_k+1)-th best maximum sweep 0 already confirmed (they cannot
    reach the output).  Forced on small batches; every reference golden, and the oracle on maps built to stress the bound:
    exact score ties around the cut, small top_k, min_score / threshold / border interplay, top_k close to the survivor count."""
    from keypoint_bench_amd.utils.extracter import detection
    monkeypatch.setenv("KPB_NMS_TILED", "0")            # small batches default to the tiled sweeps: force sweep 0 + tail
    monkeypatch.setenv("KPB_NMS_PRUNE", prune)
    g = load_golden("det_small.npz")
    for name in g["cases"]:
        p = params_from(g[name + ".params"])
        kps = detection(torch.from_numpy(g[name + ".score"])[None, None].to(_dev()), p).cpu().numpy()
        assert_kps_equal(kps, g[name + ".kps"], p["top_k"], name)
    gf = load_golden("det_full.npz")
    gens = dict(uniform=synthetic.score_uniform, smooth=synthetic.score_smooth)
    for name in gf["cases"]:
        fam, seed = gf[name + ".gen"]
        p = params_from(gf[name + ".params"])
        kps = detection(torch.from_numpy(gens[str(fam)](int(seed), 480, 640))[None, None].to(_dev()), p).cpu().numpy()
        assert_kps_equal(kps, gf[name + ".kps"], p["top_k"], name)
    rng = np.random.default_rng(12)
    maps = {"uniform": synthetic.score_uniform(91, 240, 320), "smooth": synthetic.score_smooth(92, 240, 320),
            "q16": (np.floor(rng.random((240, 320)) * 16) / 16 + 1 / 32).astype(np.float32),          # sixteen plateaus: ties everywhere
            "q256": (np.floor(rng.random((240, 320)) * 256) / 256).astype(np.float32)}
    for mname, m in maps.items():
        for nms, top_k, border, thr, ms in ((6, 1000, 8, 0.0, 0.0), (3, 50, 4, 0.0, 0.0), (2, 700, 0, 0.3, 0.0), (4, 200, 16, 0.0, 0.8),
                                            (6, 5, 8, 0.0, 0.0), (1, 3000, 2, 0.0, 0.0)):
            p = dict(nms_dist=nms, threshold=thr, border_dist=border, top_k=top_k, min_score=ms)
            got = detection(torch.from_numpy(m)[None, None].to(_dev()), p).cpu().numpy()
            want, _ = oracle.detection(m, p)
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg="%s %r" % (mname, p))


def test_nms_tail_handoff_bitmap_and_histogram_edges(monkeypatch):
    """r06: sweep 0 hands the tail a bitmap of undecided pixels (one byte per eight pixels, rows padded to four bytes) and a 4 096-bin histogram of
    the confirmed scores.  The corners of that hand-off: an odd batch (the histograms must still start on 16 bytes), a width that fills neither the
    last byte nor the last word of a bitmap row, a band of plateaus dense enough that a wave's bits overflow its LDS strip (the direct-store
    fallback) while the image stays under the list capacity, and scores far outside [2^-62, 4) (the histogram's end bins: a bound of 0 or a floor
    below every score -- still exact)."""
    from keypoint_bench_amd.utils.extracter import detection_batch, fast_nms
    monkeypatch.setenv("KPB_NMS_TILED", "0")            # small batches default to the tiled sweeps: force sweep 0 + tail
    rng = np.random.default_rng(61)
    H, W = 237, 301
    band = synthetic.score_smooth(62, H, W).copy()
    band[:20] = (np.floor(rng.random((20, W)) * 3) / 4 + 0.25).astype(np.float32)      # 20 of 237 rows: ~6 000 undecided pixels, capacity 8 917
    maps = np.stack([synthetic.score_uniform(63, H, W), band, synthetic.score_smooth(64, H, W)])[:, None]
    for r in (2, 6):
        got = fast_nms(torch.from_numpy(maps).to(_dev()), r).cpu().numpy()
        for b in range(3):
            exp, _ = oracle.fast_nms(maps[b, 0], r)
            np.testing.assert_array_equal(got[b, 0], exp, err_msg="r %d image %d" % (r, b))
    for scale in (1.0, 2.0 ** -70, 37.5):
        m = (maps * np.float32(scale)).astype(np.float32)
        for nms, top_k, border in ((2, 300, 3), (6, 40, 0)):
            p = dict(nms_dist=nms, threshold=0.0, border_dist=border, top_k=top_k, min_score=0.0)
            kps, _, n = detection_batch(torch.from_numpy(m).to(_dev()), p)
            kps, n = kps.cpu().numpy(), n.cpu().numpy()
            for b in range(3):
                want, _ = oracle.detection(m[b, 0], p)
                assert int(n[b]) == len(want), (scale, b, p)
                np.testing.assert_array_equal(kps[b, : n[b]].view(np.uint32), want.view(np.uint32), err_msg="scale %g image %d %r" % (scale, b, p))


@pytest.mark.parametrize("shape", [(480, 640), (237, 301), (96, 80)])
def test_selection_from_the_bitmap_of_confirmed_maxima_equals_the_map_scan(shape, monkeypatch):
    """r06: right after sweep 0 + tail, select_topk<BITS> reads the NMS's bitmap of confirmed maxima (a byte per eight pixels from sweep 0's owner threads, bits set
    by the tail for what it confirms) instead of the map.  It runs for batches of 64 images and more (or one-chunk maps) when the threshold is <= 0; a threshold
    of 1e-37 -- below every score -- sends the same detection through the map-reading form.  Both must give the same rows, bit for bit: 64 images, three
    bitmap rounds per image at 480 x 640, a ragged width, a map smaller than one round; a few images also against the oracle."""
    from keypoint_bench_amd.utils.extracter import detection_batch
    monkeypatch.setenv("KPB_NMS_TILED", "0")            # (64 images are below the batch at which the tail is the default)
    H, W = shape
    maps = np.stack([(synthetic.score_smooth if b % 2 else synthetic.score_uniform)(700 + b, H, W) for b in range(64)])[:, None]
    maps[5, 0, : H // 3] = (np.floor(maps[5, 0, : H // 3] * 4) / 4 + 0.25).astype(np.float32)         # plateaus: the tail has work to do, or gives up
    x = torch.from_numpy(maps).to(_dev())
    for nms, top_k, border in ((6, 1000, 8), (2, 300, 0)):
        p = dict(nms_dist=nms, threshold=0.0, border_dist=border, top_k=min(top_k, H * W), min_score=0.0)
        k0, i0, n0 = (t.cpu().numpy() for t in detection_batch(x, p))
        k1, i1, n1 = (t.cpu().numpy() for t in detection_batch(x, dict(p, threshold=1e-37)))
        np.testing.assert_array_equal(n0, n1)
        for b in range(64):
            np.testing.assert_array_equal(k0[b, : n0[b]].view(np.uint32), k1[b, : n1[b]].view(np.uint32), err_msg="image %d %r" % (b, p))
            np.testing.assert_array_equal(i0[b, : n0[b]], i1[b, : n1[b]])
        for b in (0, 5, 63):
            want, _ = oracle.detection(maps[b, 0], p)
            np.testing.assert_array_equal(k0[b, : n0[b]].view(np.uint32), want.view(np.uint32), err_msg="oracle, image %d %r" % (b, p))
