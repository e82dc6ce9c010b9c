#!/usr/bin/env python3
"""Golden fixtures for the AUC task (BASELINE configs[2] / [4]; tasks/AUC.py:101-154), produced by the REFERENCE's own `auc`,
plus known answers of its pure-numpy helpers (angle_error_mat / angle_error_vec / compute_pose_error / pose_auc).
Build container only; a no-op elsewhere.

tasks/AUC.py, utils/extracter.py and utils/matcher.py are imported as they are.  Absent third-party modules are supplied
as in the other generators: skimage.feature.match_descriptors = tests/golden/skimage_standin.py, and `cv2` is a module with
the three functions this path touches: `findEssentialMat` and `recoverPose` (the estimator calls at AUC.py:50-61) are
answered by the numpy restatement oracle/geometry_ref.py (PARITY UNPINNED for those calls) while recording what the
reference handed to them; `imwrite` discards the match plot (145-148; `plot_matches` is replaced by a no-op).  What these
fixtures pin is everything AROUND the estimator: detection, matching on all keypoints, pixel scaling per image (122-128),
intrinsics normalisation and threshold (44-48), the pose errors (66-84, 143) and the returned dict."""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def rot(ax):
    th = np.linalg.norm(ax)
    k = ax / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def main():
    if not os.path.isdir(REF):
        print("reference checkout not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.path[:0] = [HERE, ROOT]
    import skimage_standin
    from oracle import geometry_ref
    captured = {}
    sk, skf, cv2 = types.ModuleType("skimage"), types.ModuleType("skimage.feature"), types.ModuleType("cv2")
    skf.match_descriptors = lambda d0, d1, **kw: skimage_standin.match_descriptors(np.asarray(d0), np.asarray(d1), **kw)
    sk.feature = skf
    cv2.RANSAC = 8

    def findEssentialMat(k0, k1, cam, threshold=1.0, prob=0.999, method=None):
        assert method == cv2.RANSAC and np.array_equal(cam, np.eye(3))
        captured["k0"], captured["k1"], captured["thr"] = np.array(k0), np.array(k1), threshold
        E, mask, _ = geometry_ref.find_essential_ransac(k0, k1, seed=captured["seed"], threshold=threshold, prob=prob)
        captured["E"] = E
        return E, mask[:, None]

    def recoverPose(E, k0, k1, cam, dist, mask=None):
        n, R, t, mnew = geometry_ref.recover_pose(E, k0, k1, mask.ravel(), dist)
        mask[:, 0] = mnew                                   # cv2 updates the mask in place
        captured["R"], captured["t"] = R, t
        return n, R, t[:, None], mask

    cv2.findEssentialMat, cv2.recoverPose, cv2.imwrite = findEssentialMat, recoverPose, (lambda *a, **k: True)
    sys.modules.update({"cv2": cv2, "skimage": sk, "skimage.feature": skf})
    sys.path.insert(0, REF)
    import torch
    import tasks.AUC as ref_auc
    ref_auc.plot_matches = lambda *a, **k: None

    rng = np.random.default_rng(41)
    out = {"scipy_version": np.array(skimage_standin.SCIPY_VERSION)}
    # ---- known answers of the pure helpers
    Ra, Rb = rot(rng.normal(0, 0.4, 3)), rot(rng.normal(0, 0.4, 3))
    va, vb = rng.normal(size=3), rng.normal(size=3)
    T = np.eye(4); T[:3, :3] = Rb; T[:3, 3] = vb
    errs = np.abs(rng.normal(0, 12, 40))
    out["h_Ra"], out["h_Rb"], out["h_va"], out["h_vb"], out["h_errs"] = Ra, Rb, va, vb, errs
    out["h_angle_mat"] = np.float64(ref_auc.angle_error_mat(Ra, Rb))
    out["h_angle_vec"] = np.float64(ref_auc.angle_error_vec(va, vb))
    out["h_pose_err"] = np.array(ref_auc.compute_pose_error(T, Ra, va), np.float64)
    out["h_pose_auc"] = np.array(ref_auc.pose_auc(errs, [5, 10, 20]), np.float64)
    # ---- the task on synthetic two-view scenes: keypoints are projections of 3-D points, descriptors agree across views
    cases = [(96, 128, 16, 300, 0.9), (128, 160, 16, 400, 0.6), (64, 96, 8, 60, 1.0), (64, 96, 8, 4, 1.0)]      # H, W, C, points, share seen in both; last: < 5 matches
    out["n_cases"] = np.int64(len(cases))
    for c, (H, W, C, npts, share) in enumerate(cases):
        f = 0.9 * W
        K = np.array([[f, 0, (W - 1) / 2.0], [0, f, (H - 1) / 2.0], [0, 0, 1.0]], np.float32)
        R, t = rot(rng.normal(0, 0.08, 3)), rng.normal(0, 0.3, 3)
        t[2] *= 0.3
        X = np.c_[rng.uniform(-2.2, 2.2, 4 * npts), rng.uniform(-1.6, 1.6, 4 * npts), rng.uniform(4, 9, 4 * npts)]
        s0, s1 = np.zeros((H, W), np.float32), np.zeros((H, W), np.float32)
        d0, d1 = rng.normal(0, 0.05, (1, C, H, W)).astype(np.float16).astype(np.float32), rng.normal(0, 0.05, (1, C, H, W)).astype(np.float16).astype(np.float32)
        placed = 0
        for Xw in X:
            if placed >= npts:
                break
            p0 = K.astype(np.float64) @ Xw
            p1 = K.astype(np.float64) @ (R @ Xw + t)
            u0, v0, u1, v1 = p0[0] / p0[2], p0[1] / p0[2], p1[0] / p1[2], p1[1] / p1[2]
            c0, r0, c1, r1 = int(round(u0)), int(round(v0)), int(round(u1)), int(round(v1))
            if not (6 <= c0 < W - 6 and 6 <= r0 < H - 6 and 6 <= c1 < W - 6 and 6 <= r1 < H - 6):
                continue
            if s0[r0 - 3:r0 + 4, c0 - 3:c0 + 4].max() > 0 or s1[r1 - 3:r1 + 4, c1 - 3:c1 + 4].max() > 0:
                continue
            desc = rng.normal(0, 1, C).astype(np.float16).astype(np.float32)
            s0[r0, c0] = 0.5 + 0.5 * rng.random()
            d0[0, :, r0 - 1:r0 + 2, c0 - 1:c0 + 2] = desc[:, None, None]
            if rng.random() < share:
                s1[r1, c1] = 0.5 + 0.5 * rng.random()
                d1[0, :, r1 - 1:r1 + 2, c1 - 1:c1 + 2] = desc[:, None, None]
            placed += 1
        T01 = np.eye(4, dtype=np.float32); T01[:3, :3] = R; T01[:3, 3] = t
        w01 = {"intrinsics0": torch.from_numpy(K), "intrinsics1": torch.from_numpy(K), "pose01": torch.from_numpy(T01)}
        params = {"extractor_params": dict(nms_dist=2, threshold=0.0, border_dist=4, top_k=1000, min_score=0.0),
                  "matcher_params": {"brute_force_params": dict(metric="euclidean", max_distance=1.0, cross_check=True)},
                  "AUC_params": {"output": "/tmp", "th": [5, 10, 20]}}
        captured.clear()
        captured["seed"] = 0      # cv::RNG((uint64)-1): OpenCV's state at every call
        tt = torch.from_numpy
        img = torch.zeros((1, 3, H, W))
        res = ref_auc.auc(c, img, tt(s0)[None, None], tt(d0), img, tt(s1)[None, None], tt(d1), w01, {}, params)
        p = "c%d_" % c
        out[p + "score0"], out[p + "score1"], out[p + "desc0"], out[p + "desc1"] = s0, s1, d0.astype(np.float16), d1.astype(np.float16)
        out[p + "K"], out[p + "T01"] = K, T01
        out[p + "result"] = np.array([float(res["AUC"]), float(res["inliers"])], np.float64)
        if "E" in captured:
            out[p + "k0"], out[p + "k1"], out[p + "thr"], out[p + "E"] = captured["k0"], captured["k1"], np.float64(captured["thr"]), captured["E"]
            out[p + "R"], out[p + "t"] = captured["R"], captured["t"]
        print(p, "matches", len(captured.get("k0", [])), "->", out[p + "result"])
    np.savez_compressed(os.path.join(HERE, "auc.npz"), **out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
