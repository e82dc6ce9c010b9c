#!/usr/bin/env python3
"""north_star's metric bar FROM PIXELS on non-trivial geometry: "repeatability / MHA within +-0.001".

N pairs of the graded viewpoint family (`synthetic.warped_pair(i, *synthetic.viewpoint_case(i))`: a seeded HPatches-like
homography at four strengths x four noise levels, ground-truth homography known) go through

  GPU   net -> detection -> [repeatability: val_key_points] -> covisibility filter -> brute-force match -> RANSAC homography ->
        corner error -> hit@3/5/7                      (the drop-ins of tasks/repeatability.py:95-122 and tasks/MHA.py:11-72)
  CPU   the same chain from oracle/ (torch-fp32 ALIKE-t restatement, C detection / covisibility / sampling / match, the numpy
        restatement of OpenCV's RANSAC homography)

and the per-pair rows and their means are compared.  `detection` returns score-descending rows and OpenCV-style RANSAC draws
its samples by row index, so two chains with IDENTICAL keypoint sets whose scores differ by 1e-6 may permute near-equal rows and
follow different hypothesis streams: the CPU chain is therefore run a second time with its own keypoints put in the GPU's row
order, which separates "flips caused by row order alone" from every other cause.

    python scripts/metric_sweep.py [pairs] [out.json]      (GPU box; the oracle is the checker, never the product)
    python scripts/metric_sweep.py [pairs] [out.json] --auc        the AUC leg: essential-matrix RANSAC + recoverPose on synthetic.pose_pair (below)
tests/test_gpu_metric_from_pixels.py runs `sweep(64)` as a test."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

H, W = 480, 640
EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
BF = dict(metric="euclidean", max_distance=5, cross_check=True)
TH = [3, 5, 7]


def flat_index(k):
    k = np.asarray(k)
    return np.round(k[:, 1] * H - 0.5).astype(np.int64) * W + np.round(k[:, 0] * W - 0.5).astype(np.int64)


def pair_inputs(i):
    from keypoint_bench_amd import synthetic
    v0, v1, h01 = synthetic.warped_pair(i, H, W, *synthetic.viewpoint_case(i))
    h10 = np.linalg.inv(h01.astype(np.float64)).astype(np.float32)          # datasets/hpatches.py:80-82
    return v0, v1, h01, h10


def cpu_chain(k0, k1, d0, d1, h01, h10):
    """Everything after detection, from oracle/ pieces; returns the row of figures the sweep compares."""
    import oracle
    from oracle import geometry_ref
    w01, w10 = dict(homography_matrix=h01, width=W, height=H), dict(homography_matrix=h10, width=W, height=H)
    rep = oracle.val_key_points(k0, k1, w01, w10, th=3)
    ids0 = oracle.warp_homography(k0[:, :2], h01, W, H)[2]                  # tasks/MHA.py:33-34
    ids1 = oracle.warp_homography(k1[:, :2], h10, W, H)[2]
    c0, c1 = k0[ids0], k1[ids1]
    flags, mean_dist, nm, inl, mset = [0.0] * len(TH), float("nan"), 0, 0, set()
    if len(c0) and len(c1):
        f0, f1 = oracle.sample(d0, c0), oracle.sample(d1, c1)
        pairs, _ = oracle.match(f0, f1, BF["max_distance"], BF["cross_check"])
        nm = len(pairs)
        i0, i1 = flat_index(c0), flat_index(c1)
        mset = set(zip(i0[pairs[:, 0]].tolist(), i1[pairs[:, 1]].tolist()))
        if nm >= 4:
            sc = np.array([W - 1, H - 1], np.float32)                       # MHA.py:40-44, float32 products
            Hm, _, info = geometry_ref.find_homography_ransac(c0[pairs[:, 0], :2] * sc, c1[pairs[:, 1], :2] * sc, seed=0)
            if Hm is not None:
                mean_dist = float(geometry_ref.mha_corner_error(Hm, h01, H, W, H, W))
                flags = [float(mean_dist <= t) for t in TH]
                inl = int(info["inliers"])
    return dict(rep=float(rep["repeatability"]), rep_err=float(rep["mean_error"]), num_feat=int(rep["num_feat"]), flags=flags,
                mean_dist=mean_dist, matches=nm, inliers=inl, mset=mset)


def cpu_pair(args):
    i, order0, order1 = args
    torch.set_num_threads(1)
    import oracle
    from oracle import alike_ref
    from keypoint_bench_amd import weights
    t = {k: torch.from_numpy(v) for k, v in weights.load_alike_t().items()}
    v0, v1, h01, h10 = pair_inputs(i)
    ks, ds, idxs = [], [], []
    for v in (v0, v1):
        with torch.no_grad():
            s, d = alike_ref.alnet_forward(torch.from_numpy(v)[None], t)
        k, idx = oracle.detection(s[0, 0].numpy(), EP)
        ks.append(k), ds.append(d[0].numpy()), idxs.append(np.asarray(idx, np.int64))
    own = cpu_chain(ks[0], ks[1], ds[0], ds[1], h01, h10)
    reordered = None
    if all(o is not None and len(o) == len(x) and set(o.tolist()) == set(x.tolist()) for o, x in zip((order0, order1), idxs)):
        perm = []
        for o, x in zip((order0, order1), idxs):
            pos = {int(p): r for r, p in enumerate(x)}
            perm.append(np.array([pos[int(p)] for p in o]))
        reordered = cpu_chain(ks[0][perm[0]], ks[1][perm[1]], ds[0], ds[1], h01, h10)
    return dict(idx=idxs, own=own, reordered=reordered)


def gpu_pair(net, i):
    """The drop-in path main.py takes, one pair at a time.  Returns (row of figures, keypoint pixel orders)."""
    from keypoint_bench_amd.tasks.MHA import corner_hits, mha
    from keypoint_bench_amd.tasks.repeatability import val_key_points
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import brute_force_matcher
    from keypoint_bench_amd.utils.mvg import find_homography
    from keypoint_bench_amd.utils.projection import warp
    dev = "cuda:0"
    v0, v1, h01, h10 = pair_inputs(i)
    t = lambda a: torch.from_numpy(a).to(dev)
    img0, img1 = t(v0)[None], t(v1)[None]
    s0, d0 = net(img0)
    s1, d1 = net(img1)
    k0, k1 = detection(s0, EP), detection(s1, EP)
    w01 = dict(mode="homo", homography_matrix=t(h01), width=torch.tensor(W), height=torch.tensor(H))
    w10 = dict(mode="homo", homography_matrix=t(h10), width=torch.tensor(W), height=torch.tensor(H))
    rep = val_key_points(k0, k1, w01, w10, th=3)
    c0, c1 = warp(k0, w01)[0], warp(k1, w10)[0]
    flags, mean_dist, nm, inl, mset = [0.0] * len(TH), float("nan"), 0, 0, set()
    if c0.shape[0] and c1.shape[0]:
        m0, m1 = brute_force_matcher(c0, c1, d0, d1, BF)
        nm = int(m0.shape[0])
        mset = set(zip(flat_index(m0.cpu().numpy()).tolist(), flat_index(m1.cpu().numpy()).tolist()))
        if nm >= 4:
            Hm, _, info = find_homography(m0[:, 0:2], m1[:, 0:2], [W - 1, H - 1, W - 1, H - 1], seed=0)
            if int(info[0, 0]):
                flags, mean_dist = corner_hits(Hm[0].cpu().numpy(), h01, np.asarray(H), np.asarray(W), H, W, TH)
                mean_dist, inl = float(mean_dist), int(info[0, 1])
    params = {"MHA_params": {"th": TH}, "extractor_params": EP, "matcher_params": {"brute_force_params": BF}}
    task_flags = [float(f) for f in mha(i, img0, s0, d0, img1, s1, d1, w01, w10, params)]
    assert task_flags == [float(f) for f in flags], (i, task_flags, flags)          # the task itself = its pieces
    row = dict(rep=float(rep["repeatability"]), rep_err=float(rep["mean_error"]), num_feat=int(rep["num_feat"]), flags=[float(f) for f in flags],
               mean_dist=mean_dist, matches=nm, inliers=inl, mset=mset)
    return row, (flat_index(k0.cpu().numpy()), flat_index(k1.cpu().numpy()))


def _means(rows):
    rep = float(np.mean([r["rep"] for r in rows]))
    err = float(np.nanmean([r["rep_err"] for r in rows]))
    mha = [float(np.mean([r["flags"][j] for r in rows])) for j in range(len(TH))]
    return rep, err, mha


def sweep(n, first=0, workers=None):
    import multiprocessing as mp
    import oracle
    oracle.build()
    from keypoint_bench_amd.models.ALike import alike_t
    net = alike_t().eval()
    t0 = time.time()
    gpu, orders = [], []
    for i in range(first, first + n):
        row, order = gpu_pair(net, i)
        gpu.append(row), orders.append(order)
    torch.cuda.synchronize()
    gpu_s = time.time() - t0
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers or min(16, len(os.sched_getaffinity(0)))) as pool:
        cpu = pool.map(cpu_pair, [(first + j, orders[j][0], orders[j][1]) for j in range(n)])
    cpu_s = time.time() - t0

    same_sets = sum(all(set(o.tolist()) == set(x.tolist()) for o, x in zip(orders[j], cpu[j]["idx"])) for j in range(n))
    same_order = sum(all(np.array_equal(o, x) for o, x in zip(orders[j], cpu[j]["idx"])) for j in range(n))
    own = [c["own"] for c in cpu]
    reo = [c["reordered"] for c in cpu]
    flips_total = [sum(g["flags"][t] != o["flags"][t] for g, o in zip(gpu, own)) for t in range(len(TH))]
    flips_order = [sum(r is not None and r["flags"][t] != o["flags"][t] for r, o in zip(reo, own)) for t in range(len(TH))]
    flips_other = [sum(r is not None and g["flags"][t] != r["flags"][t] for g, r in zip(gpu, reo)) for t in range(len(TH))]
    g_rep, g_err, g_mha = _means(gpu)
    c_rep, c_err, c_mha = _means(own)
    per_pair = []
    for j in range(n):
        g, o, r = gpu[j], own[j], reo[j]
        per_pair.append({"pair": first + j, "case": list(__import__("keypoint_bench_amd.synthetic", fromlist=["x"]).viewpoint_case(first + j)),
                         "rep_gpu": g["rep"], "rep_cpu": o["rep"], "matches_gpu": g["matches"], "matches_cpu": o["matches"],
                         "same_match_set": g["mset"] == o["mset"], "inliers_gpu": g["inliers"], "inliers_cpu": o["inliers"],
                         "mean_dist_gpu": g["mean_dist"], "mean_dist_cpu": o["mean_dist"],
                         "mean_dist_cpu_in_gpu_row_order": None if r is None else r["mean_dist"],
                         "flags_gpu": g["flags"], "flags_cpu": o["flags"]})
    md = np.array([[p["mean_dist_gpu"], p["mean_dist_cpu"], np.nan if p["mean_dist_cpu_in_gpu_row_order"] is None else p["mean_dist_cpu_in_gpu_row_order"]] for p in per_pair])
    fin = np.isfinite(md).all(1) & (md[:, :2].max(1) < 50)
    return {"pairs": n, "first_pair": first, "size": "640x480", "family": "synthetic.warped_pair x viewpoint_case", "extractor": EP, "matcher": BF, "th": TH,
            "repeatability_gpu": g_rep, "repeatability_cpu": c_rep, "abs_diff_repeatability": abs(g_rep - c_rep),
            "rep_mean_error_gpu": g_err, "rep_mean_error_cpu": c_err, "abs_diff_rep_mean_error": abs(g_err - c_err),
            "pairs_with_different_repeatability": int(sum(g["rep"] != o["rep"] for g, o in zip(gpu, own))),
            "mha_gpu": g_mha, "mha_cpu": c_mha, "abs_diff_mha": [abs(a - b) for a, b in zip(g_mha, c_mha)],
            "pairs_with_identical_keypoint_sets": int(same_sets), "pairs_in_identical_row_order": int(same_order),
            "pairs_with_identical_match_sets": int(sum(g["mset"] == o["mset"] for g, o in zip(gpu, own))),
            "mha_flag_flips_gpu_vs_cpu": flips_total, "mha_flag_flips_from_row_order_alone": flips_order,
            "mha_flag_flips_gpu_vs_cpu_in_gpu_row_order": flips_other,
            "max_abs_mean_dist_diff_gpu_vs_cpu_px": float(np.abs(md[fin, 0] - md[fin, 1]).max()) if fin.any() else None,
            "max_abs_mean_dist_diff_gpu_vs_cpu_in_gpu_row_order_px": float(np.abs(md[fin, 0] - md[fin, 2]).max()) if fin.any() else None,
            "gpu_chain_seconds": round(gpu_s, 1), "cpu_chain_seconds": round(cpu_s, 1), "per_pair": per_pair}


# ------------------------------------------------------------------------------------------------ AUC leg (r05)
# tasks/AUC.py:101-154 on the POSE family (synthetic.pose_pair: two planes at different depths, seeded camera motion, intrinsics in the
# task's pixel convention): detection, brute-force match on ALL keypoints, cv2.findEssentialMat(RANSAC) + recoverPose restated,
# max(translation angle error, rotation angle error) per pair, pose_auc at 5 / 10 / 20 degrees.  OpenCV's sampler draws by ROW INDEX, so
# the oracle chain runs twice -- its own rows in its own order and in the GPU's order -- which separates what row order alone does to
# the essential-matrix RANSAC (BASELINE configs[2]'s estimator, 1 500 MegaDepth pairs) from every other cause.
AUC_TH = [5, 10, 20]


def pose_inputs(i):
    from keypoint_bench_amd import synthetic
    return synthetic.pose_pair(i, H, W, *synthetic.pose_case(i))


def cpu_pose_chain(k0, k1, d0, d1, K, pose):
    import oracle
    from oracle import geometry_ref
    f0, f1 = oracle.sample(d0, k0), oracle.sample(d1, k1)
    pairs, _ = oracle.match(f0, f1, BF["max_distance"], BF["cross_check"])
    i0, i1 = flat_index(k0), flat_index(k1)
    mset = set(zip(i0[pairs[:, 0]].tolist(), i1[pairs[:, 1]].tolist()))
    err, inl = 180.0, 0
    if len(pairs) >= 5:
        sc = np.array([W - 1, H - 1], np.float32)                           # AUC.py:125-126: float32 keypoints x float32 sizes
        ret = geometry_ref.estimate_pose(k0[pairs[:, 0], :2] * sc, k1[pairs[:, 1], :2] * sc, K, K, 1.0, seed=0)
        if ret is not None:
            R, t, mask = ret
            et, er = geometry_ref.compute_pose_error(pose, R, t)
            err, inl = float(max(et, er)), int(mask.sum())
    return dict(err=err, inliers=inl, matches=len(pairs), mset=mset)


def cpu_pose_pair(args):
    i, order0, order1 = args
    torch.set_num_threads(1)
    import oracle
    from oracle import alike_ref
    from keypoint_bench_amd import weights
    t = {k: torch.from_numpy(v) for k, v in weights.load_alike_t().items()}
    v0, v1, K, pose = pose_inputs(i)
    ks, ds, idxs = [], [], []
    for v in (v0, v1):
        with torch.no_grad():
            s, d = alike_ref.alnet_forward(torch.from_numpy(v)[None], t)
        k, idx = oracle.detection(s[0, 0].numpy(), EP)
        ks.append(k), ds.append(d[0].numpy()), idxs.append(np.asarray(idx, np.int64))
    own = cpu_pose_chain(ks[0], ks[1], ds[0], ds[1], K, pose)
    reordered = None
    if all(o is not None and len(o) == len(x) and set(o.tolist()) == set(x.tolist()) for o, x in zip((order0, order1), idxs)):
        perm = []
        for o, x in zip((order0, order1), idxs):
            pos = {int(p): r for r, p in enumerate(x)}
            perm.append(np.array([pos[int(p)] for p in o]))
        reordered = cpu_pose_chain(ks[0][perm[0]], ks[1][perm[1]], ds[0], ds[1], K, pose)
    return dict(idx=idxs, own=own, reordered=reordered)


def gpu_pose_pair(net, i):
    """tasks/AUC.py's `auc` through the drop-ins, one pair at a time; also its pieces, to have the match set."""
    from keypoint_bench_amd.tasks.AUC import auc
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.utils.matcher import brute_force_matcher
    dev = "cuda:0"
    v0, v1, K, pose = pose_inputs(i)
    t = lambda a: torch.from_numpy(a).to(dev)
    img0, img1 = t(v0)[None], t(v1)[None]
    s0, d0 = net(img0)
    s1, d1 = net(img1)
    k0, k1 = detection(s0, EP), detection(s1, EP)
    m0, m1 = brute_force_matcher(k0, k1, d0, d1, BF)
    mset = set(zip(flat_index(m0.cpu().numpy()).tolist(), flat_index(m1.cpu().numpy()).tolist()))
    w01 = dict(mode="se3", intrinsics0=torch.from_numpy(K), intrinsics1=torch.from_numpy(K), pose01=torch.from_numpy(pose))
    params = {"AUC_params": {"th": AUC_TH}, "extractor_params": EP, "matcher_params": {"brute_force_params": BF}}
    r = auc(i, img0, s0, d0, img1, s1, d1, w01, None, params)
    row = dict(err=float(r["AUC"]), inliers=int(r["inliers"]), matches=int(m0.shape[0]), mset=mset)
    return row, (flat_index(k0.cpu().numpy()), flat_index(k1.cpu().numpy()))


def sweep_auc(n, first=0, workers=None):
    import multiprocessing as mp
    import oracle
    oracle.build()
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.runner import pose_auc
    net = alike_t().eval()
    t0 = time.time()
    gpu, orders = [], []
    for i in range(first, first + n):
        row, order = gpu_pose_pair(net, i)
        gpu.append(row), orders.append(order)
    torch.cuda.synchronize()
    gpu_s = time.time() - t0
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers or min(16, len(os.sched_getaffinity(0)))) as pool:
        cpu = pool.map(cpu_pose_pair, [(first + j, orders[j][0], orders[j][1]) for j in range(n)])
    cpu_s = time.time() - t0
    own, reo = [c["own"] for c in cpu], [c["reordered"] for c in cpu]
    same_sets = sum(all(set(o.tolist()) == set(x.tolist()) for o, x in zip(orders[j], cpu[j]["idx"])) for j in range(n))
    same_order = sum(all(np.array_equal(o, x) for o, x in zip(orders[j], cpu[j]["idx"])) for j in range(n))
    eg, eo = np.array([g["err"] for g in gpu]), np.array([o["err"] for o in own])
    er = np.array([np.nan if r is None else r["err"] for r in reo])
    a_g, a_o = pose_auc(eg, AUC_TH), pose_auc(eo, AUC_TH)
    have = np.isfinite(er)
    a_r = pose_auc(np.where(have, er, eo), AUC_TH)
    cross = lambda a, b: [int(((a <= t) != (b <= t)).sum()) for t in AUC_TH]
    from keypoint_bench_amd import synthetic
    per_pair = [{"pair": first + j, "case": list(synthetic.pose_case(first + j)), "err_gpu": gpu[j]["err"], "err_cpu": own[j]["err"],
                 "err_cpu_in_gpu_row_order": None if reo[j] is None else reo[j]["err"], "matches_gpu": gpu[j]["matches"], "matches_cpu": own[j]["matches"],
                 "same_match_set": gpu[j]["mset"] == own[j]["mset"], "inliers_gpu": gpu[j]["inliers"], "inliers_cpu": own[j]["inliers"],
                 "inliers_cpu_in_gpu_row_order": None if reo[j] is None else reo[j]["inliers"]} for j in range(n)]
    return {"pairs": n, "first_pair": first, "size": "640x480", "family": "synthetic.pose_pair x pose_case", "extractor": EP, "matcher": BF, "th": AUC_TH,
            "auc_gpu": [float(a) for a in a_g], "auc_cpu": [float(a) for a in a_o], "auc_cpu_in_gpu_row_order": [float(a) for a in a_r],
            "abs_diff_auc_gpu_vs_cpu": [abs(float(a) - float(b)) for a, b in zip(a_g, a_o)],
            "abs_diff_auc_gpu_vs_cpu_in_gpu_row_order": [abs(float(a) - float(b)) for a, b in zip(a_g, a_r)],
            "abs_diff_auc_from_row_order_alone": [abs(float(a) - float(b)) for a, b in zip(a_r, a_o)],
            "pairs_with_identical_keypoint_sets": int(same_sets), "pairs_in_identical_row_order": int(same_order),
            "pairs_with_identical_match_sets": int(sum(g["mset"] == o["mset"] for g, o in zip(gpu, own))),
            "max_abs_err_diff_gpu_vs_cpu_in_gpu_row_order_deg": float(np.abs(eg[have] - er[have]).max()) if have.any() else None,
            "pairs_with_different_inlier_count_in_gpu_row_order": int(sum(r is not None and g["inliers"] != r["inliers"] for g, r in zip(gpu, reo))),
            "max_abs_err_diff_from_row_order_alone_deg": float(np.abs(er[have] - eo[have]).max()) if have.any() else None,
            "mean_abs_err_diff_from_row_order_alone_deg": float(np.abs(er[have] - eo[have]).mean()) if have.any() else None,
            "threshold_crossings_gpu_vs_cpu": cross(eg, eo), "threshold_crossings_from_row_order_alone": cross(np.where(have, er, eo), eo),
            "threshold_crossings_gpu_vs_cpu_in_gpu_row_order": cross(eg, np.where(have, er, eg)),
            "median_err_cpu_deg": float(np.median(eo)), "gpu_chain_seconds": round(gpu_s, 1), "cpu_chain_seconds": round(cpu_s, 1), "per_pair": per_pair}


def main():
    import json
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    if "--auc" in sys.argv:
        r = sweep_auc(n)
        print(json.dumps({k: v for k, v in r.items() if k != "per_pair"}))
        out = [a for a in sys.argv[2:] if not a.startswith("--")]
        if out:
            with open(out[0], "w") as f:
                json.dump(r, f, indent=1)
        return
    r = sweep(n)
    print(json.dumps({k: v for k, v in r.items() if k != "per_pair"}))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(r, f, indent=1)


if __name__ == "__main__":
    main()
