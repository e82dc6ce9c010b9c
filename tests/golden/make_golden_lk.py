#!/usr/bin/env python3
"""Golden fixtures for SURVEY 8(f) rank 4 (tensor Lucas-Kanade tracker), produced by the REFERENCE's OpticalFlow class.

Runs only in the build container.  utils/matcher.py imports cv2 and skimage at module level; both are absent from this
image and are supplied as blank modules (OpticalFlow is pure torch and touches neither).  The reference draws its random
start offsets with torch.randn; the script seeds the generator, so the same angles can be stored next to the outputs.

Usage:  python tests/golden/make_golden_lk.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    if not os.path.isdir(REF):
        print("reference checkout not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sk = types.ModuleType("skimage"); skf = types.ModuleType("skimage.feature"); skf.match_descriptors = None
    sys.modules.setdefault("skimage", sk); sys.modules.setdefault("skimage.feature", skf)
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    import torch
    import utils.matcher as M
    from keypoint_bench_amd import synthetic

    torch.set_num_threads(4)
    cases = [  # H, W, n, params
        (96, 128, 40, dict(distance=3, win_size=3, levels=1, interation=40, gray=False)),      # the class defaults
        (96, 128, 40, dict(distance=3, win_size=7, levels=2, interation=20, gray=False)),
        (160, 224, 60, dict(distance=10, win_size=21, levels=3, interation=40, gray=False)),   # config/config_fund.yaml:72-77
    ]
    out = {"n_cases": np.int64(len(cases))}
    rng = np.random.default_rng(77)
    for c, (H, W, n, prm) in enumerate(cases):
        v0, v1 = synthetic.image_pair(300 + c, H, W)            # view1 = view0 shifted by (3, 2) px + noise
        pts = np.stack([rng.uniform(0.15, 0.85, n), rng.uniform(0.15, 0.85, n)], 1).astype(np.float32)
        seed = 1000 + c
        torch.manual_seed(seed)
        angle = torch.randn(n) * 6.28                            # what OpticalFlow.__call__ draws first (matcher.py:55)
        unit = torch.stack([torch.cos(angle), torch.sin(angle)], 1).numpy()
        torch.manual_seed(seed)
        of = M.OpticalFlow(prm)
        p, err = of(torch.from_numpy(v0)[None], torch.from_numpy(v1)[None], torch.from_numpy(pts), torch.from_numpy(pts.copy()))
        k = "c%d_" % c
        out[k + "image_pair"] = np.array([300 + c, H, W], np.int64)      # keypoint_bench_amd.synthetic.image_pair(seed, H, W)
        out[k + "pts"], out[k + "unit"] = pts, unit.astype(np.float32)
        out[k + "prm"] = np.array([prm["distance"], prm["win_size"], prm["levels"], prm["interation"]], np.int64)
        out[k + "out"], out[k + "err"] = p[0].numpy(), err[0].numpy()
        d = p[0].numpy() - pts * np.array([W - 1, H - 1], np.float32)
        print(k, "median flow", np.median(d, 0), "err<8:", int((err[0].numpy() < 8).sum()), "/", n)
    np.savez_compressed(os.path.join(HERE, "lk.npz"), **out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
