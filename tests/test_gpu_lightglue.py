"""GPU parity of N5 (LightGlue, csrc/lightglue.hip through the C ABI) against the reference's own outputs on its fp32
CPU path with seeded stand-in weights (the LightGlue checkpoints are absent from the reference tree), covering the
full-depth, early-stop and point-pruning branches, and against the torch-fp32 oracle."""
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, GOLDEN
from keypoint_bench_amd import weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
sys.path.insert(0, GOLDEN)
import make_golden_lightglue as mk   # noqa: E402  (only its deterministic input synthesiser is used here)


def _matcher(seed, dim, scale, variant, **kw):
    from keypoint_bench_amd.models.lightglue import LightGlue
    m = LightGlue(features=None, desc_scale=scale, **kw)
    m.load_state_dict(weights.random_lightglue_state_dict(seed, dim, variant))
    return m


@pytest.mark.parametrize("name", ["sp_plain", "sp_stop", "sp_prune", "disk_plain", "disk_prune", "disk_n1000", "sp_n1000",
                                  "disk_n1536_th1024", "disk_n2048_th1536"])
def test_lightglue_against_reference_golden(name):
    """The last two cases run ABOVE the keypoint counts at which the reference's CUDA paths start to prune
    (pruning_keypoint_thresholds 1024 / 1536, lightglue.py:352-357, 574-589): 1536 and 2048 keypoints, the matcher built with
    the same `prune_min_kpts` the reference instance was configured with, pruning until a side drops below it."""
    g = load_golden("lightglue.npz")
    dim, scale, seed, n0, n1 = (int(v) for v in g[name + ".cfg"])
    dm0, dm1, p0, p1 = mk.inputs(seed, dim, scale, n0=n0, n1=n1)
    th = int(g[name + ".prune_th"]) if name + ".prune_th" in g else -1
    m = _matcher(seed, dim, scale, str(g[name + ".variant"]), prune_min_kpts=th)
    T = lambda a: torch.from_numpy(a).to(DEV)
    pairs, scores, stop = m.match_indices(T(p0), T(p1), T(dm0), T(dm1), {"w": 320, "h": 240})
    assert stop == int(g[name + ".stop"])
    want = g[name + ".matches"]
    got = pairs.cpu().numpy()
    # 9 fp32 transformer layers: a match whose score sits within 1e-3 of the 0.1 threshold may flip; everything else is exact
    ws = {tuple(r): s for r, s in zip(want.tolist(), g[name + ".scores"].tolist())}
    gs = {tuple(r): s for r, s in zip(got.tolist(), scores.cpu().numpy().tolist())}
    for r in set(ws) ^ set(gs):
        s = ws.get(r, gs.get(r))
        assert abs(s - 0.1) < 1e-3, "%s: match %s (score %.4f) differs and is not at the threshold" % (name, r, s)
    common = sorted(set(ws) & set(gs))
    assert len(common) >= 0.98 * len(ws)
    np.testing.assert_allclose([gs[r] for r in common], [ws[r] for r in common], rtol=2e-3, atol=1e-5)
    assert np.all(np.diff(got[:, 0]) > 0)          # ascending in the first index, as torch.where leaves them
    a, b = m.match(T(p0), T(p1), T(dm0), T(dm1), {"w": 320, "h": 240})
    np.testing.assert_array_equal(a.cpu().numpy(), p0[got[:, 0]])
    np.testing.assert_array_equal(b.cpu().numpy(), p1[got[:, 1]])


@pytest.mark.parametrize("name", ["sp_plain", "disk_plain", "disk_prune", "disk_n1000"])
def test_lightglue_f16_attention_against_the_reference_fed_half_operands(name):
    """attention="f16" (r05): what the reference runs on a GPU -- lightglue.py:129-134 hands q.half(), k.half(), v.half() to
    scaled_dot_product_attention and casts the half result back.  Fixtures `<case>_f16` are the reference CLASS with exactly those three
    statements taken on the CPU (tests/golden/make_golden_lightglue.py:half_attention).  Half-precision rounding is not reproducible
    across implementations (accumulation order), so the bar is the one the fp32 cases use, a little wider: the same matches except at
    the 0.1 threshold, scores within 3e-3 relative (measured: 1e-3 at worst on 1000 keypoints, 3e-5 on average)."""
    g = load_golden("lightglue.npz")
    fx = name + "_f16"
    dim, scale, seed, n0, n1 = (int(v) for v in g[fx + ".cfg"])
    dm0, dm1, p0, p1 = mk.inputs(seed, dim, scale, n0=n0, n1=n1)
    T = lambda a: torch.from_numpy(a).to(DEV)
    out = {}
    for mode in ("fp32", "f16"):
        m = _matcher(seed, dim, scale, str(g[fx + ".variant"]), attention=mode)
        pairs, scores, stop = m.match_indices(T(p0), T(p1), T(dm0), T(dm1), {"w": 320, "h": 240})
        out[mode] = ({tuple(r): s for r, s in zip(pairs.cpu().numpy().tolist(), scores.cpu().numpy().tolist())}, stop)
    gs, stop = out["f16"]
    assert stop == int(g[fx + ".stop"])
    ws = {tuple(r): s for r, s in zip(g[fx + ".matches"].tolist(), g[fx + ".scores"].tolist())}
    for r in set(ws) ^ set(gs):
        s = ws.get(r, gs.get(r))
        assert abs(s - 0.1) < 2e-3, "%s: match %s (score %.4f) differs and is not at the threshold" % (fx, r, s)
    common = sorted(set(ws) & set(gs))
    assert len(common) >= 0.98 * len(ws)
    np.testing.assert_allclose([gs[r] for r in common], [ws[r] for r in common], rtol=3e-3, atol=2e-5)
    # the knob does something: the two arithmetics agree to half precision, not to the bit
    both = sorted(set(out["fp32"][0]) & set(gs))
    d = np.array([abs(out["fp32"][0][r] - gs[r]) for r in both])
    assert d.max() > 0 and d.max() < 2e-3, d.max()
    with pytest.raises(ValueError):
        _matcher(seed, dim, scale, "plain", attention="bf16")


def test_lightglue_cuda_pruning_threshold_and_empty():
    """prune_min_kpts=1024 (what the reference does on CUDA) disables pruning below 1024 points; empty inputs give no matches."""
    from oracle import lightglue_ref as R
    dim, scale, seed = 256, 8, 23
    dm0, dm1, p0, p1 = mk.inputs(seed, dim, scale)
    sd = weights.random_lightglue_state_dict(seed, dim, "prune")
    m = _matcher(seed, dim, scale, "prune", prune_min_kpts=1024)
    T = lambda a: torch.from_numpy(a).to(DEV)
    pairs, scores, stop = m.match_indices(T(p0), T(p1), T(dm0), T(dm1), {"w": 320, "h": 240})
    t = {k: torch.from_numpy(v) for k, v in weights.tensors_lightglue(sd).items()}
    with torch.no_grad():
        _, _, out = R.match(t, torch.from_numpy(p0), torch.from_numpy(p1), torch.from_numpy(dm0), torch.from_numpy(dm1), {"w": 320, "h": 240}, scale,
                            pruning_th=1024)
    want = set(map(tuple, out["matches"].numpy().tolist()))
    got = set(map(tuple, pairs.cpu().numpy().tolist()))
    assert len(want ^ got) <= max(2, len(want) // 50) and stop == out["stop"]
    e0, e1 = m.match(torch.zeros(0, 3, device=DEV), T(p1), T(dm0), T(dm1), {"w": 320, "h": 240})
    assert e0.shape == (0, 3) and e1.shape == (0, 3)


def test_lightglue_batch_with_pairs_that_stop_at_different_layers_equals_the_single_pair_runs():
    """r06: the assignment stage runs once after the last layer for every finished pair, each with the index map of the layer IT stopped at.  Four pairs in
    one kpb_lg_match call -- inputs chosen so that they stop at different layers, odd and even (the two index buffers alternate by layer), one of them an
    empty pair -- must give exactly what four single-pair calls give (same kernels, same per-pair arithmetic: bit for bit)."""
    import ctypes
    from keypoint_bench_amd._lib import LgParams, ptr
    g = load_golden("lightglue.npz")
    dim, scale, seed, _, _ = (int(v) for v in g["sp_stop.cfg"])
    m = _matcher(seed, dim, scale, str(g["sp_stop.variant"]))
    T = lambda a: torch.from_numpy(a).to(DEV)
    cases = []
    for s2, n0g, n0, n1 in ((seed, int(g["sp_stop.cfg"][3]), int(g["sp_stop.cfg"][3]), int(g["sp_stop.cfg"][4])), (seed + 1, 300, 300, 280), (seed + 2, 300, 0, 120),
                            (seed + 3, 512, 512, 400)):
        dm0, dm1, p0, p1 = mk.inputs(s2, dim, scale, n0=n0g, n1=n1)
        cases.append((dm0, dm1, p0[:n0], p1))
    singles = [m.match_indices(T(p0), T(p1), T(d0), T(d1), {"w": 320, "h": 240}) for d0, d1, p0, p1 in cases]
    stops = [s[2] for s in singles]
    assert len(set(stops)) >= 2, stops                   # the point of the test: different stop layers in one batch
    B, K = len(cases), max(max(len(c[2]), len(c[3])) for c in cases)
    dev = torch.device(DEV)
    P0 = torch.zeros((B, K, 3), device=dev); P1 = torch.zeros((B, K, 3), device=dev)
    for b, (_, _, p0, p1) in enumerate(cases):
        P0[b, : len(p0)] = T(p0); P1[b, : len(p1)] = T(p1)
    D0 = torch.stack([T(c[0])[0] for c in cases]).contiguous(); D1 = torch.stack([T(c[1])[0] for c in cases]).contiguous()
    n0 = torch.tensor([len(c[2]) for c in cases], dtype=torch.int32, device=dev); n1 = torch.tensor([len(c[3]) for c in cases], dtype=torch.int32, device=dev)
    pairs = torch.empty((B, K, 2), dtype=torch.int32, device=dev); scores = torch.empty((B, K), dtype=torch.float32, device=dev)
    k = torch.zeros((B,), dtype=torch.int32, device=dev); stop = torch.zeros((B,), dtype=torch.int32, device=dev)
    _, C, Hd, Wd = D0.shape
    sb, sc, sh, sw = D0.stride()
    prm = LgParams(float(m.conf["depth_confidence"]), float(m.conf["width_confidence"]), float(m.conf["filter_threshold"]), m.prune_min_kpts)
    ctx = m._ctx
    ctx.check(ctx.lib.kpb_lg_match(m._handle, ptr(P0), ptr(P1), ptr(n0), ptr(n1), B, K, ptr(D0), ptr(D1), C, Hd, Wd, sb, sc, sh, sw, 320, 240,
                                   ctypes.byref(prm), ptr(pairs), ptr(scores), ptr(k), ptr(stop)))
    for b, (sp, ss, st) in enumerate(singles):
        kb = int(k[b])
        assert int(stop[b]) == st, (b, int(stop[b]), st)
        np.testing.assert_array_equal(pairs[b, :kb].cpu().numpy(), sp.cpu().numpy().astype(np.int32), err_msg="pair %d" % b)
        np.testing.assert_array_equal(scores[b, :kb].cpu().numpy().view(np.uint32), ss.cpu().numpy().view(np.uint32), err_msg="pair %d" % b)
