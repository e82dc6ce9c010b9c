#!/usr/bin/env python3
"""Golden fixtures for the FundamentalMatrix task (BASELINE configs[3]; tasks/FundamentalMatrix.py:89-161), produced by the
REFERENCE.  Build container only (reference mounted read-only at /root/reference); a no-op elsewhere.

The reference's tasks/FundamentalMatrix.py, utils/extracter.py and utils/matcher.py are imported as they are.  Absent
third-party modules are supplied as in the other generators: cv2 blank (untouched on this path), models.lightglue needs
nothing extra, and skimage.feature.match_descriptors = tests/golden/skimage_standin.py (scipy.cdist + skimage's documented
glue).  The fixture holds inputs (score maps, descriptor maps, fundamental matrices, parameters) and the three numbers the
reference's function returned per pair, plus the matched rows it was computed from.  No reference source is copied.

Usage:  python tests/golden/make_golden_fund.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def fundamental_from_motion(rng, w, h):
    """F = K^-T [t]x R K^-1 for a small random camera motion (datasets/tartanair.py builds its F the same way)."""
    f = 320.0
    K = np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])
    a = rng.normal(0, 0.03, 3)
    th = np.linalg.norm(a)
    k = a / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = rng.normal(0, 1, 3)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(K)
    F = Ki.T @ tx @ R @ Ki
    return (F / np.abs(F).max()).astype(np.float32)


def main():
    if not os.path.isdir(REF):
        print("reference checkout not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import skimage_standin
    from keypoint_bench_amd import synthetic
    sk, skf = types.ModuleType("skimage"), types.ModuleType("skimage.feature")
    captured = {}

    def match_descriptors(d0, d1, **kw):
        pairs = skimage_standin.match_descriptors(np.asarray(d0), np.asarray(d1), **kw)
        captured["pairs"] = pairs
        return pairs

    skf.match_descriptors = match_descriptors
    sk.feature = skf
    sys.modules.update({"cv2": types.ModuleType("cv2"), "skimage": sk, "skimage.feature": skf})
    sys.path.insert(0, REF)
    import torch
    import tasks.FundamentalMatrix as fm

    torch.set_num_threads(4)
    rng = np.random.default_rng(77)
    out = {}
    cases = [  # H, W, C, div, nms, top_k, max_distance, matcher type, th
        (96, 128, 16, 1, 3, 120, 5.0, "brute_force", 3.0),
        (96, 128, 32, 8, 2, 200, 1.5, "brute_force", 1.0),
        (64, 96, 16, 2, 4, 40, 9.0, "light_glue", 0.5),        # matcher None -> the brute-force branch of 124-126
        (128, 160, 16, 1, 6, 1000, 5.0, "brute_force", 3.0),
    ]
    out["n_cases"] = np.int64(len(cases))
    out["scipy_version"] = np.array(skimage_standin.SCIPY_VERSION)
    for c, (H, W, C, div, nms, top_k, maxd, mtype, th) in enumerate(cases):
        s0 = synthetic.score_uniform(500 + c, H, W) if c % 2 == 0 else synthetic.score_smooth(500 + c, H, W)
        s1 = np.roll(s0, (1, 2), (0, 1)).copy()
        s1 = np.clip(s1 + rng.normal(0, 0.01, s1.shape), 0, 1).astype(np.float32)
        # descriptor maps hold fp16-representable values so that the fixture can store them in half the bytes
        d0 = rng.normal(size=(1, C, H // div, W // div)).astype(np.float16).astype(np.float32)
        d1 = (np.roll(d0, (1 // div, 2 // div), (2, 3)) + 0.1 * rng.normal(size=d0.shape)).astype(np.float16).astype(np.float32)
        F = fundamental_from_motion(rng, W, H)
        params = {"extractor_params": dict(nms_dist=nms, threshold=0.0, border_dist=4, top_k=top_k, min_score=0.0),
                  "matcher_params": {"type": mtype, "brute_force_params": dict(metric="euclidean", max_distance=maxd, cross_check=True)},
                  "FundamentalMatrix_params": {"th": th}}
        batch = {"fundamental": torch.from_numpy(F)[None]}
        t = lambda a: torch.from_numpy(a)
        res = fm.fundamental_matrix(c, None, batch, t(s0)[None, None], t(s1)[None, None], t(d0), t(d1), None, params)
        p = "c%d_" % c
        out[p + "score0"], out[p + "score1"], out[p + "desc0"], out[p + "desc1"], out[p + "F"] = s0, s1, d0.astype(np.float16), d1.astype(np.float16), F
        out[p + "prm"] = np.array([nms, 4, top_k, maxd, th, 0 if mtype == "brute_force" else 1], np.float64)
        out[p + "pairs"] = captured["pairs"]
        out[p + "result"] = np.array([float(res["fundamental_error"]), float(res["fundamental_radio"]), float(res["fundamental_num"])], np.float64)
        print(p, "matches", len(captured["pairs"]), "->", out[p + "result"])
    np.savez_compressed(os.path.join(HERE, "fund.npz"), **out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
