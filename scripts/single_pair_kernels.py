#!/usr/bin/env python3
"""GPU time per kernel of ONE pair on the drop-in path (HIP events on the launch stream, kpb_prof_*): what the ~0.7 ms of
single_pair_latency.py is made of on the device side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keypoint_bench_amd import synthetic, _lib
from keypoint_bench_amd.models.ALike import alike_t
from keypoint_bench_amd.utils.extracter import detection
from keypoint_bench_amd.utils.matcher import brute_force_matcher
EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
BF = dict(metric="euclidean", max_distance=5, cross_check=True)
dev = "cuda:0"
v0, v1 = synthetic.image_pair(1)
i0, i1 = torch.from_numpy(v0)[None].to(dev), torch.from_numpy(v1)[None].to(dev)
net = alike_t(dense_descriptors=True).eval()
def pair():
    s0, d0 = net(i0); s1, d1 = net(i1)
    k0, k1 = detection(s0, EP), detection(s1, EP)
    return brute_force_matcher(k0, k1, d0, d1, BF)
for _ in range(5): pair()
ctx = _lib.Context.get(torch.device(dev))
ctx.prof_enable(True)
n = 20
for _ in range(n): pair()
torch.cuda.synchronize()
rep = ctx.prof_report()
ctx.prof_enable(False)
tot = 0.0
for name, (calls, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print("%-24s %5.1f launches/pair  %7.1f us/pair  (%5.1f us each)" % (name, calls / n, ms / n * 1e3, ms / calls * 1e3))
    tot += ms / n
print("device total %.3f ms/pair, %d launches/pair" % (tot, sum(c for c, _ in rep.values()) / n))
