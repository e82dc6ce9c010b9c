#!/usr/bin/env python3
"""Throughput when every step's images start in pinned host memory (the reference's DataLoader hands over host tensors):
copy then compute on one stream, and copy of step i+1 on a second stream under the compute of step i."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keypoint_bench_amd import synthetic
from keypoint_bench_amd.models.ALike import alike_t
from keypoint_bench_amd.pipeline import PairPipeline
EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
BF = dict(metric="euclidean", max_distance=5, cross_check=True)
dev = torch.device("cuda:0")
B, H, W, steps = 256, 480, 640, 10
pairs = [synthetic.image_pair(i, H, W) for i in range(8)]
host = torch.from_numpy(np.stack([pairs[i % 8][0] for i in range(B)] + [pairs[i % 8][1] for i in range(B)])).pin_memory()
pipe = PairPipeline(alike_t().eval(), EP, BF, B, H, W, dev)
bufs = [host.to(dev) for _ in range(2)]
copy_stream = torch.cuda.Stream(dev)
def serial():
    for _ in range(steps):
        bufs[0].copy_(host, non_blocking=True)
        pipe.run(bufs[0])
def overlapped():
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    with torch.cuda.stream(copy_stream):
        bufs[0].copy_(host, non_blocking=True); ev[0].record(copy_stream)
    for i in range(steps):
        cur, nxt = i & 1, (i + 1) & 1
        if i + 1 < steps:
            copy_stream.wait_stream(torch.cuda.current_stream(dev))      # the buffer about to be overwritten is no longer read
            with torch.cuda.stream(copy_stream):
                bufs[nxt].copy_(host, non_blocking=True); ev[nxt].record(copy_stream)
        torch.cuda.current_stream(dev).wait_event(ev[cur])
        pipe.run(bufs[cur])
# the same images as the decoder delivers them: uint8 HWC, a quarter of the bytes; kpb_preprocess makes the fp32 CHW tensor on the device
from keypoint_bench_amd.utils.preprocess import to_tensor_resized
host8 = (host.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).contiguous().pin_memory()
bufs8 = [host8.to(dev) for _ in range(2)]
def overlapped_u8():
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    with torch.cuda.stream(copy_stream):
        bufs8[0].copy_(host8, non_blocking=True); ev[0].record(copy_stream)
    for i in range(steps):
        cur, nxt = i & 1, (i + 1) & 1
        if i + 1 < steps:
            copy_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(copy_stream):
                bufs8[nxt].copy_(host8, non_blocking=True); ev[nxt].record(copy_stream)
        torch.cuda.current_stream(dev).wait_event(ev[cur])
        pipe.run(to_tensor_resized(bufs8[cur], None, False, dev))
for name, fn in (("uint8 + kpb_preprocess", overlapped_u8), ("resident (bench.py)", lambda: [pipe.run(bufs[0]) for _ in range(steps)]), ("copy, then compute", serial), ("copy under compute", overlapped)):
    fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    nbytes = host8.numel() if "uint8" in name else (0 if "resident" in name else host.numel() * 4)
    print("%-22s %.2f ms/step -> %.0f pairs/s  (%.1f GB/s host->device)" % (name, dt * 1e3, B / dt, nbytes / dt / 1e9))
