#!/usr/bin/env python3
"""Latency of the reference's own calling pattern (one pair at a time, model_interface.py:205-212 + tasks/MHA.py:30-39)
through the drop-ins, on cuda:0."""
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
for dense in (True, False):
    net = alike_t(dense_descriptors=dense).eval()
    def pair():
        s0, d0 = net(i0); s1, d1 = net(i1)
        k0, k1 = detection(s0, EP), detection(s1, EP)
        return brute_force_matcher(k0, k1, d0, d1, BF)
    for _ in range(5): pair()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 50
    for _ in range(n): m0, m1 = pair()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("dense" if dense else "keypoint-only", "%.2f ms/pair -> %.0f pairs/s, %d matches" % (dt * 1e3, 1 / dt, m0.shape[0]))
