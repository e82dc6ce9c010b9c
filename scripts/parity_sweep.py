#!/usr/bin/env python3
"""How often does the whole GPU path (split-f16 MFMA net -> NMS / top-K -> sampling -> fp64 match) return exactly what the
fp32 CPU chain (oracle/: torch-fp32 ALIKE-t restatement + C detection / sampling / match) returns on full-size pairs?
The stages are bit-exact on equal inputs; the net's score map differs from the CPU's by ~1e-6, which can flip an NMS decision
between near-equal neighbours.  Prints, over `pairs` synthetic 640x480 pairs: max score / descriptor differences, the
number of images whose keypoint index sets are identical, and the number of pairs whose match sets are identical.
    python scripts/parity_sweep.py [pairs] [out.json] [--also-fp32]    (GPU box; the oracle is the checker, never the product)
--also-fp32 repeats the sweep in a child process with KPB_FP32_MATRIX=1 (the strict-fp32 kernels; the library reads the knob once
per process) and records its figures under "strict_fp32": how much of the score difference is the split-f16 products and how
much is summation order.  tests/test_gpu_parity_sweep.py runs `sweep(8)` as a test."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
BF = dict(metric="euclidean", max_distance=5, cross_check=True)


def cpu_pair(i):
    torch.set_num_threads(1)
    import oracle
    from oracle import alike_ref
    from keypoint_bench_amd import synthetic, weights
    t = {k: torch.from_numpy(v) for k, v in weights.load_alike_t().items()}
    out = []
    for v in synthetic.image_pair(i, 480, 640):
        with torch.no_grad():
            s, d = alike_ref.alnet_forward(torch.from_numpy(v)[None], t)
        k, idx = oracle.detection(s[0, 0].numpy(), EP)
        out.append((s[0, 0].numpy(), k, idx, oracle.sample(d[0].numpy(), k)))
    pairs, _ = oracle.match(out[0][3], out[1][3], BF["max_distance"], BF["cross_check"])
    return out, pairs


def sweep(n, first=0, workers=None):
    """Compares pairs first .. first + n - 1; returns the figures as a dict."""
    import multiprocessing as mp
    import oracle
    oracle.build()
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers or min(16, len(os.sched_getaffinity(0)))) as pool:
        ref = pool.map(cpu_pair, range(first, first + n))
    cpu_s = time.time() - t0
    from keypoint_bench_amd import synthetic
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import match_descriptors, sample_descriptors
    net = alike_t().eval()
    same_kps = same_matches = same_order = 0
    diffm = 0
    ds = dd = 0.0
    nk = nm = diffk = 0
    for i in range(n):
        views = synthetic.image_pair(first + i, 480, 640)
        feats, gidx, widx = [], [], []
        for j, v in enumerate(views):
            s, d = net(torch.from_numpy(v)[None].cuda())
            k = detection(s, EP)
            want_s, want_k, want_idx, want_f = ref[i][0][j]
            ds = max(ds, float(np.abs(s[0, 0].cpu().numpy() - want_s).max()))
            got_idx = (np.round(k[:, 1].cpu().numpy() * 480 - 0.5).astype(np.int64) * 640 + np.round(k[:, 0].cpu().numpy() * 640 - 0.5).astype(np.int64))
            a, b = set(got_idx.tolist()), set(np.asarray(want_idx).tolist())
            same_kps += a == b
            diffk += len(a ^ b)
            nk += len(b)
            f = sample_descriptors(k, d)
            if a == b:      # descriptors compared keypoint by keypoint (rows of near-equal score may be permuted)
                order_g, order_w = np.argsort(got_idx, kind="stable"), np.argsort(np.asarray(want_idx), kind="stable")
                dd = max(dd, float(np.abs(f.cpu().numpy()[order_g] - want_f[order_w]).max()))
            feats.append(f)
            gidx.append(got_idx)
            widx.append(np.asarray(want_idx))
            same_order += np.array_equal(got_idx, np.asarray(want_idx))
        pairs = match_descriptors(feats[0], feats[1], max_distance=BF["max_distance"], cross_check=BF["cross_check"]).cpu().numpy()
        # rows are ordered by score: a 1e-6 score difference may permute near-equal rows, so matches are compared as pixel pairs
        got_m = set(zip(gidx[0][pairs[:, 0]].tolist(), gidx[1][pairs[:, 1]].tolist()))
        want_m = set(zip(widx[0][ref[i][1][:, 0]].tolist(), widx[1][ref[i][1][:, 1]].tolist()))
        same_matches += got_m == want_m
        diffm += len(got_m ^ want_m)
        nm += len(ref[i][1])
    return {"pairs": n, "first_pair_seed": first, "images": 2 * n, "size": "640x480", "extractor": EP, "matcher": BF,
            "arithmetic": "strict fp32 (KPB_FP32_MATRIX=1)" if os.environ.get("KPB_FP32_MATRIX", "0") not in ("", "0") else "split-f16 MFMA triples",
            "max_abs_score_diff": ds, "max_abs_descriptor_diff": dd,
            "images_with_identical_keypoint_sets": int(same_kps), "images_in_identical_row_order": int(same_order),
            "keypoints": int(nk), "keypoints_differing": int(diffk // 2),
            "pairs_with_identical_match_sets": int(same_matches), "matches": int(nm), "matches_differing": int(diffm // 2),
            "cpu_chain_seconds": round(cpu_s, 1)}


def main():
    import json
    import subprocess
    import tempfile
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(argv[0]) if argv else 32
    r = sweep(n)
    if "--also-fp32" in sys.argv:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "fp32.json")
            subprocess.run([sys.executable, os.path.abspath(__file__), str(n), out], env=dict(os.environ, KPB_FP32_MATRIX="1"), check=True,
                           stdout=subprocess.DEVNULL)
            with open(out) as f:
                r["strict_fp32"] = {k: v for k, v in json.load(f).items() if k not in ("extractor", "matcher", "size")}
    print(json.dumps(r))
    if len(argv) > 1:
        with open(argv[1], "w") as f:
            json.dump(r, f, indent=1)


if __name__ == "__main__":
    main()
