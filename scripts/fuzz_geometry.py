#!/usr/bin/env python3
"""Random two-view scenes: the RANSAC kernels (homography, fundamental matrix) against the numpy restatement, hypothesis for
hypothesis (same sampler): found flag, iteration count, inlier count, mask, model.
    python scripts/fuzz_geometry.py      (GPU box; the oracle is the checker, never the product)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from oracle import geometry_ref as g
from test_oracle_geometry import synth, fscene
from keypoint_bench_amd.utils.mvg import find_homography, find_fundamental
rng = np.random.default_rng(5)
DEV = "cuda:0"
badH = badF = 0
sc = np.array([639, 479, 639, 479], np.float32)
for case in range(40):
    n = int(rng.integers(8, 1000)); share = float(rng.uniform(0.3, 1.0)); noise = float(rng.uniform(0, 1.0)); seed = int(rng.integers(0, 2**31))
    src, dst, H, inl = synth(n, share, noise, 1000 + case)
    a = torch.from_numpy((src / sc[:2]).astype(np.float32)).to(DEV); b = torch.from_numpy((dst / sc[2:]).astype(np.float32)).to(DEV)
    Hd, md, idv = find_homography(a, b, sc, seed=seed)
    p0 = (a.cpu().numpy() * sc[:2]).astype(np.float64); p1 = (b.cpu().numpy() * sc[2:]).astype(np.float64)
    He, me, ie = g.find_homography_ransac(p0, p1, seed=seed)
    i = idv.cpu().numpy()[0]
    if He is None:
        ok = i[0] == 0
    else:
        ok = i[0] == 1 and i[1] == ie["inliers"] and i[2] == ie["iters"] and np.array_equal(md.cpu().numpy()[0], me) and np.allclose(Hd.cpu().numpy()[0], He, atol=3e-7 * np.abs(He).max())
    badH += not ok
    if not ok: print("H mismatch", case, n, share, noise, i, ie)
    q1, q2, F, finl = fscene(n, share, noise, 2000 + case)
    a = torch.from_numpy((q1 / sc[:2]).astype(np.float32)).to(DEV); b = torch.from_numpy((q2 / sc[2:]).astype(np.float32)).to(DEV)
    Fd, md, idv = find_fundamental(a, b, sc, seed=seed)
    p0 = (a.cpu().numpy() * sc[:2]).astype(np.float64); p1 = (b.cpu().numpy() * sc[2:]).astype(np.float64)
    Fe, me, ie = g.find_fundamental_ransac(p0, p1, seed=seed)
    i = idv.cpu().numpy()[0]
    if Fe is None:
        ok = i[0] == 0
    else:
        ok = i[0] == 1 and i[2] == ie["iters"] and abs(int(i[1]) - ie["inliers"]) <= 1
    badF += not ok
    if not ok: print("F mismatch", case, n, share, noise, i, ie)
print("homography mismatches", badH, "fundamental mismatches", badF)
