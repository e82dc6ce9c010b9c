#!/usr/bin/env python3
"""Probe: does running the two halves of a step's images (forward + detection) on two contexts / streams concurrently fill the
gaps the under-filled kernels leave?  Prints ms per 512 images for one lane of 512 and two lanes of 256."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from keypoint_bench_amd import _lib, synthetic, weights
from keypoint_bench_amd._lib import DetectParams, c_void_p, ptr

L = _lib.load()
dev = torch.device("cuda:0")
H, W, K = 480, 640, 1000
blob = open(os.path.join(os.path.dirname(_lib.SO_PATH), "weights", "alike-t.kpbw"), "rb").read()
prm = DetectParams(6, 0.0, 8, K, 0.0)


class Lane:
    def __init__(self, B):
        self.stream = torch.cuda.Stream(dev)
        self.ctx = c_void_p()
        assert L.kpb_ctx_create(0, c_void_p(self.stream.cuda_stream), ctypes.byref(self.ctx)) == 0
        self.net = c_void_p()
        assert L.kpb_net_create(self.ctx, weights.ARCH_ALIKE, blob, len(blob), ctypes.byref(self.net)) == 0
        self.B = B
        f32, i32 = torch.float32, torch.int32
        self.score = torch.empty((B, 1, H, W), dtype=f32, device=dev)
        self.desc = torch.empty((B, H, W, 64), dtype=f32, device=dev)
        self.kps = torch.empty((B, K, 3), dtype=f32, device=dev)
        self.idx = torch.empty((B, K), dtype=i32, device=dev)
        self.n = torch.empty((B,), dtype=i32, device=dev)

    def enqueue(self, images):
        rc = L.kpb_net_forward(self.net, ptr(images), self.B, H, W, ptr(self.score), ptr(self.desc))
        assert rc == 0, L.kpb_last_error(self.ctx)
        rc = L.kpb_detect(self.ctx, ptr(self.score), self.B, H, W, ctypes.byref(prm), ptr(self.kps), ptr(self.idx), ptr(self.n), 0)
        assert rc == 0, L.kpb_last_error(self.ctx)

    def finish(self):
        if SPIN:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            while not ev.query():
                pass
        assert L.kpb_detect_check(self.ctx) >= 0


v = [synthetic.image_pair(i)[i & 1] for i in range(16)]
images = torch.from_numpy(np.stack([v[i % 16] for i in range(512)])).to(dev).contiguous()
SPIN = False
for lanes, SPIN in ((1, False), (1, True), (2, False), (2, True)):
    ls = [Lane(512 // lanes) for _ in range(lanes)]
    parts = [images[i * (512 // lanes):(i + 1) * (512 // lanes)] for i in range(lanes)]
    def step():
        for l, p in zip(ls, parts): l.enqueue(p)
        for l in ls: l.finish()
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 30
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("lanes %d spin %d: %.3f ms per 512 images (forward + detection)" % (lanes, SPIN, dt * 1e3), flush=True)
    ref = [l.n.clone() for l in ls]
    del ls
    torch.cuda.empty_cache()
