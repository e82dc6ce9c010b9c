#!/usr/bin/env python3
"""Where a single pair's ~1 ms goes on the drop-in path: wall time per stage with a device sync after each (so the stages do not
overlap: the sum exceeds the pipelined figure of single_pair_latency.py), dense maps and keypoint-only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypoint_bench_amd import synthetic
from keypoint_bench_amd.models.ALike import alike_t
from keypoint_bench_amd.utils.extracter import detection
from keypoint_bench_amd.utils.matcher import brute_force_matcher
EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
BF = dict(metric="euclidean", max_distance=5, cross_check=True)
dev = "cuda:0"
v0, v1 = synthetic.image_pair(1)
i0, i1 = torch.from_numpy(v0)[None].to(dev), torch.from_numpy(v1)[None].to(dev)
sync = torch.cuda.synchronize
for dense in (True, False):
    net = alike_t(dense_descriptors=dense).eval()
    acc = {"forward x2": 0.0, "detection x2": 0.0, "matcher": 0.0}
    n = 60
    for it in range(n + 5):
        sync(); t0 = time.perf_counter()
        s0, d0 = net(i0); s1, d1 = net(i1)
        sync(); t1 = time.perf_counter()
        k0, k1 = detection(s0, EP), detection(s1, EP)
        sync(); t2 = time.perf_counter()
        m = brute_force_matcher(k0, k1, d0, d1, BF)
        sync(); t3 = time.perf_counter()
        if it >= 5:
            acc["forward x2"] += t1 - t0; acc["detection x2"] += t2 - t1; acc["matcher"] += t3 - t2
    print("dense" if dense else "keypoint-only", "  ".join("%s %.3f ms" % (k, v / n * 1e3) for k, v in acc.items()), " sum %.3f ms" % (sum(acc.values()) / n * 1e3))
