#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the persistent head (alike_head_f16p), from a ONE-OFF instrumented build:
    python scripts/head_stamps.py build      (build container: hipcc -DKPB_STAMPS -> scripts/_bin/libkpb_stamps.so)
    python scripts/head_stamps.py run        (GPU box: a few 512-image steps with that library, then the stamps of up to 256 sampled waves)
One row group in the middle of a workgroup's walk is stamped (group 10 of 30): shader clocks from the group's top."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "scripts", "_bin", "libkpb_stamps.so")
NAMES = {0: "group_top", 1: "requests_landed(vmcnt)", 2: "barrier_done", 3: "next_requests_issued",
         4: "t0_begin(vmcnt)", 5: "t0_x1_read+next_x1_requested", 6: "t0_taps_read+16_stores", 7: "t0_features+score", 8: "t0_split+chain0_issued",
         9: "t0_chain1+16_stores_issued", 10: "t0_accumulators_swapped",
         12: "t1_begin(vmcnt)", 13: "t1_x1_read+next_x1_requested", 14: "t1_taps_read+16_stores", 15: "t1_features+score", 16: "t1_split+chain0_issued",
         17: "t1_chain1+16_stores_issued", 18: "t1_accumulators_swapped", 20: "group_end"}


def build():
    from keypoint_bench_amd import build as kb
    objs = []
    os.makedirs(os.path.join(ROOT, "scripts", "_bin", "obj"), exist_ok=True)
    for src in kb.SOURCES:
        o = os.path.join(ROOT, "scripts", "_bin", "obj", src.replace(".hip", ".o"))
        subprocess.check_call(["hipcc"] + kb.FLAGS + ["-DKPB_STAMPS", "-c", os.path.join(kb.CSRC, src), "-o", o])
        objs.append(o)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    print(SO)


def run():
    import numpy as np
    import torch
    from keypoint_bench_amd import _lib
    _lib.SO_PATH = SO
    from keypoint_bench_amd import synthetic
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.pipeline import PairPipeline
    dev = torch.device("cuda:0")
    B = 256
    v = [synthetic.image_pair(i) for i in range(16)]
    images = torch.from_numpy(np.stack([v[i % 16][0] for i in range(B)] + [v[i % 16][1] for i in range(B)])).to(dev).contiguous()
    pipe = PairPipeline(alike_t().eval(), dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0),
                        dict(metric="euclidean", max_distance=5, cross_check=True), B, 480, 640, device=dev)
    L = _lib.load()
    buf = (ctypes.c_ulonglong * (256 * 32))()
    slots = ctypes.c_uint(0)
    L.kpb_debug_head_stamps.restype = ctypes.c_int
    for _ in range(6):
        pipe.run(images)
    L.kpb_debug_head_stamps(buf, ctypes.byref(slots))        # discard the warm-up steps' stamps (resets the slot counter)
    pipe.run(images)
    torch.cuda.synchronize()
    assert L.kpb_debug_head_stamps(buf, ctypes.byref(slots)) == 0
    n = min(int(slots.value), 256)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 32)[:n].astype(np.int64)
    print("# in-kernel s_memtime stamps of alike_head_f16p (r04 build + stamps, %d sampled waves of one 512-image launch; one row group in the" % n)
    print("# middle of a 30-group walk; clocks from the group's top; median, step, 10th / 90th percentile)")
    prev = 0
    for i in sorted(NAMES):
        d = a[:, i] - a[:, 0]
        med = int(np.median(d))
        print("%-34s med %7d  (+%6d)   p10 %7d p90 %7d" % (NAMES[i], med, med - prev, int(np.percentile(d, 10)), int(np.percentile(d, 90))))
        prev = med


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
