#!/usr/bin/env python3
"""Run-to-run determinism of one forward at a given shape: N forwards of the same image, each compared bit for bit with the first.
Prints, per differing run, how many score / descriptor elements differ and their bounding boxes (a clue to the kernel and tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keypoint_bench_amd import synthetic
from keypoint_bench_amd.models.ALike import alike_t
H, W, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dense = (sys.argv[4] if len(sys.argv) > 4 else "dense") == "dense"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
img = torch.from_numpy(synthetic.image_pair(41, H, W)[0])[None].to("cuda:0").repeat(B, 1, 1, 1)
net = alike_t(dense_descriptors=dense).eval()
net(img)          # warm-up: allocations, lazy initialisation
s0, d0 = net(img)
s0 = s0.clone(); d0 = d0.clone() if dense else None
bad = 0
for it in range(N):
    s, d = net(img)
    ds = (s != s0)[:, 0].any(dim=0)
    if ds.any():
        bad += 1
        ys, xs = torch.nonzero(ds, as_tuple=True)
        print("run %d: score differs at %d px, rows %d..%d cols %d..%d, max |d| %.3g" % (it, ys.numel(), ys.min(), ys.max(), xs.min(), xs.max(), (s - s0).abs().max().item()))
    if dense:
        dd = (d != d0).any(dim=0).any(dim=0)
        if dd.any():
            ys, xs = torch.nonzero(dd, as_tuple=True)
            print("run %d: desc differs at %d px, rows %d..%d cols %d..%d" % (it, ys.numel(), ys.min(), ys.max(), xs.min(), xs.max()))
if dense and bad and (d != d0).any():       # where in a 64-pixel column band and a 4-row group the last run's differing pixels sit, and how many channels
    dd3 = (d != d0)
    ys, xs = torch.nonzero(dd3.any(dim=0).any(dim=0), as_tuple=True)
    hx = torch.bincount((xs % 64) // 16, minlength=4).tolist()
    hy = torch.bincount(ys % 4, minlength=4).tolist()
    nch = dd3.sum(dim=1)[dd3.any(dim=1)].float()
    print("last run: pixels by (x %% 64) // 16: %s; by y %% 4: %s; channels differing per pixel: min %d median %d max %d" % (hx, hy, nch.min(), nch.median(), nch.max()))
print("%dx%d %s batch %d: %d of %d runs differ from the first" % (H, W, "dense" if dense else "sparse", B, bad, N))
