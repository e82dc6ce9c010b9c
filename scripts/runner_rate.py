#!/usr/bin/env python3
"""Throughput of the REAL task path: PairRunner.run over a synthetic 640x480 pair dataset (the batched PairPipeline under
the runner) for the tasks the library evaluates on the device, next to bench.py's bare pipeline loop.

    python scripts/runner_rate.py [--pairs 256] [--batch 256] [--repeat 3] [--tasks match_stats repeatability ...]

Dataset items are device-resident tensors (bench.py's convention: inputs resident in HBM) unless --host is given, in
which case they are numpy arrays as datasets/hpatches.py returns them and every image crosses PCIe inside the loop."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1024, help="pairs per run (four batches by default: the host halves of a batch run under the next batch)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--host", action="store_true")
    ap.add_argument("--u8", action="store_true", help="with --host: items are decoded uint8 [H,W,3] images (a quarter of the PCIe bytes)")
    ap.add_argument("--files", choices=["png", "jpeg", "ppm"], default=None,
                    help="items are image FILES, decoded by PIL on the runner's prefetch pool (keypoint_bench_amd/datasets.py: SURVEY 8(f)2's decode stage)")
    ap.add_argument("--decode-workers", type=int, default=None)
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--tasks", nargs="+", default=["match_stats", "repeatability", "MHA", "AUC"])
    ap.add_argument("--sequence", action="store_true", help="a sequence dataset instead (frames sliding over one canvas): FundamentalMatrix and visual_odometer tasks")
    ap.add_argument("--model", default="Alike", choices=["Alike", "XFeat"])
    args = ap.parse_args()
    import numpy as np
    import torch
    from keypoint_bench_amd import runner, synthetic
    dev = "cuda:0"
    H, W = 480, 640
    h01 = np.array([[1, 0, -3], [0, 1, -2], [0, 0, 1]], np.float32)
    views = [synthetic.image_pair(5000 + i, H, W) for i in range(args.distinct)]
    if args.u8:
        views = [tuple(np.ascontiguousarray((v.transpose(1, 2, 0) * 255.0 + 0.5).astype(np.uint8)) for v in pair) for pair in views]
    if not args.host:
        views = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in views]
    ds = []
    K = np.array([[500.0, 0, 319.5], [0, 500.0, 239.5], [0, 0, 1]], np.float32)
    T01 = np.eye(4, dtype=np.float32)
    T01[0, 3] = 1.0
    for i in range(args.pairs):
        v0, v1 = views[i % args.distinct]
        ds.append({"image0": v0, "image1": v1, "dataset": "HPatches",
                   "warp01_params": dict(mode="homo", homography_matrix=h01, width=W, height=H, intrinsics0=K, intrinsics1=K, pose01=T01),
                   "warp10_params": dict(mode="homo", homography_matrix=np.linalg.inv(h01).astype(np.float32), width=W, height=H)})
    decode_only = None
    if args.files:      # the same views as encoded files on local disk (page cache after the first read, like a warmed dataset)
        import tempfile
        from PIL import Image
        from keypoint_bench_amd import datasets
        tmp = tempfile.mkdtemp(prefix="kpb_files_")
        paths = []
        for j in range(args.distinct):
            pp = []
            for k, v in enumerate(synthetic.image_pair(5000 + j, H, W)):
                u8 = np.ascontiguousarray((v.transpose(1, 2, 0) * 255.0 + 0.5).astype(np.uint8))
                f = os.path.join(tmp, "%d_%d.%s" % (j, k, args.files))
                Image.fromarray(u8).save(f, format=args.files.upper(), quality=92)
                pp.append(f)
            paths.append(pp)
        recs = [dict(it, image0=paths[i % args.distinct][0], image1=paths[i % args.distinct][1]) for i, it in enumerate(ds)]
        ds = datasets.ImagePairFiles(recs)
        t0 = time.perf_counter()        # what the decode pool alone delivers (no device work); .ppm: the files READ into arrays (lazy_raw=False)
        with datasets.Prefetcher(datasets.ImagePairFiles(recs, lazy_raw=False), range(args.pairs), workers=args.decode_workers) as pf:
            n = sum(1 for _ in pf)
        decode_only = round(n / (time.perf_counter() - t0), 1)
    if args.sequence:
        canvas, _ = synthetic.image_pair(4242, H + 64, W + 64)
        rng = np.random.default_rng(3)
        ds = []
        for i in range(args.pairs):
            dy, dx = (i * 3) % 64, (i * 5) % 64
            fr = np.ascontiguousarray(canvas[:, dy:dy + H, dx:dx + W])
            ds.append({"image0": fr if args.host else torch.from_numpy(fr).to(dev), "dataset": "TartanAir",
                       "fundamental": rng.normal(size=(3, 3)).astype(np.float32), "fx": 500.0, "fy": 500.0, "cx": 319.5, "cy": 239.5,
                       "ground_truth": np.array([0.05 * i, 0, 0, 0, 0, 0, 1], np.float32),
                       "last_ground_truth": np.array([0.05 * max(i - 1, 0), 0, 0, 0, 0, 0, 1], np.float32)})
        if args.tasks == ["match_stats", "repeatability", "MHA", "AUC"]:
            args.tasks = ["FundamentalMatrix", "visual_odometer"]
    out = {"pairs": args.pairs, "batch": args.batch, "items": ("%s files %s (%s threads)" % (args.files, "read straight into the pinned staging ring (datasets.RawImage; HPatches' own format, hpatches.py:36)" if args.files == "ppm" else "decoded by PIL", args.decode_workers or "one per core, <= 16")) if args.files else
           ("host uint8 HWC" if args.u8 else "host numpy fp32 CHW") if args.host else "device tensors",
           "decode_pool_alone_pairs_per_s": decode_only,
           "descriptors": "dense-map" if args.dense else "keypoint-only"}
    for task in args.tasks:
        params = {"model_type": args.model, "task_type": task, "XFeat_params": {}, "FundamentalMatrix_params": {"th": 3.0}, "Alike_params": dict(c1=8, c2=16, c3=32, c4=64, dim=64),
                  "extractor_params": dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0),
                  "matcher_params": {"type": "brute_force", "brute_force_params": dict(metric="euclidean", max_distance=5, cross_check=True)},
                  "repeatability_params": {"th": 3}, "MHA_params": {"th": [3, 5, 7]}, "AUC_params": {"th": [5, 10, 20]}}
        model = None
        if args.model == "XFeat":       # its checkpoint is not in the reference tree: seeded stand-in weights, as in bench.py
            from keypoint_bench_amd.models.XFeat import xfeat_random
            model = xfeat_random(9).eval()
        r = runner.PairRunner(params, model=model, device=dev, batch=args.batch, dense_descriptors=args.dense)
        r.decode_workers = args.decode_workers
        try:
            agg, _ = r.run(ds)            # warm-up: allocations, first-shape workspaces
        except (ImportError, NotImplementedError) as e:
            out[task] = "unavailable: %s" % e
            continue
        best = 1e9
        for _ in range(args.repeat):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            agg, rows = r.run(ds)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out[task] = {"pairs_per_s": round(args.pairs / best, 1), "ms_per_pair": round(1e3 * best / args.pairs, 4), "batched_pairs": r.batched_pairs,
                     "aggregate": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in agg.items() if not hasattr(v, "shape")}}
        del r
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
