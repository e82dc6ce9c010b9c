#!/usr/bin/env python3
"""Random descriptor sets: kpb_match (drop-in `match_descriptors`, MFMA prefilter + exact refinement) against oracle.match,
index pairs and float64 distances bit for bit.
    python scripts/fuzz_match.py [cases] [seed]      (GPU box; the oracle is the checker, never the product)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import oracle
from keypoint_bench_amd.utils.matcher import match_descriptors

def run(n_cases=200, seed=0):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        n, m = int(rng.integers(0, 1200)), int(rng.integers(0, 1200))
        C = int(rng.choice([64, 128, 256]))
        kind = case % 4
        a = rng.normal(size=(n, C)).astype(np.float32)
        if kind == 0:
            b = rng.normal(size=(m, C)).astype(np.float32)
        elif kind == 1:      # noisy copies: the typical matching case
            idx = rng.integers(0, max(n, 1), m) if n else np.zeros(m, np.int64)
            b = (a[idx] + rng.normal(0, 0.05, (m, C))).astype(np.float32) if n else rng.normal(size=(m, C)).astype(np.float32)
        elif kind == 2:      # exact duplicates and coarse values: ties
            a = (rng.integers(-2, 3, (n, C)) / 2).astype(np.float32)
            b = (rng.integers(-2, 3, (m, C)) / 2).astype(np.float32)
            if n and m:
                b[: min(n, m) // 2] = a[: min(n, m) // 2]
        else:                # normalised, large scale differences
            b = rng.normal(size=(m, C)).astype(np.float32)
            a /= np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-6)
            b *= 100.0
        maxd = float(rng.choice([np.inf, 5.0, 0.7, 12.0]))
        cc = bool(rng.integers(0, 2))
        want_p, want_d = oracle.match(a, b, maxd, cc)
        got = match_descriptors(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), max_distance=maxd, cross_check=cc, return_distance=True)
        gp, gd = got[0].cpu().numpy(), got[1].cpu().numpy()
        if gp.shape != want_p.shape or not np.array_equal(gp, want_p) or not np.array_equal(gd.view(np.uint64), np.asarray(want_d, np.float64).view(np.uint64)):
            bad += 1
            print("MISMATCH case", case, (n, m, C), maxd, cc, gp.shape, want_p.shape)
    print("cases", n_cases, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
