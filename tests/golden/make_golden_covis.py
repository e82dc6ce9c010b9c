#!/usr/bin/env python3
"""Golden fixtures for SURVEY 8(f) rank 1 (covisibility warp + ground-truth mutual NN), produced by the REFERENCE.

Runs only in the build container (reference mounted read-only at /root/reference).  The reference's
utils/projection.py and tasks/repeatability.py are imported as they are; `cv2` is absent from this image and is
supplied as a blank module (neither `warp_homography` nor `val_key_points` touches it).  The fixture holds inputs and
the outputs the reference computed: no reference source is copied.

Usage:  python tests/golden/make_golden_covis.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def homography(rng, strength, w, h):
    a = rng.normal(0, strength, (3, 3))
    m = np.eye(3) + a * np.array([[1, 1, 0.15 * w], [1, 1, 0.15 * h], [1.0 / w, 1.0 / h, 0]])
    return m.astype(np.float32)


def keypoints(rng, n, size):
    flat = rng.choice(size * size, n, replace=False)
    flat.sort()
    rows, cols = flat // size, flat % size
    k = np.stack([(cols + 0.5) / size, (rows + 0.5) / size, rng.random(n)], 1).astype(np.float32)
    return k


def main():
    if not os.path.isdir(REF):
        print("reference checkout not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import torch
    import tasks.repeatability as rep
    import utils.projection as proj

    torch.set_num_threads(1)
    rng = np.random.default_rng(2024)
    out = {}
    cases = [  # n0, n1, strength, (w0,h0), (w1,h1), resize key present
        (17, 23, 0.02, (640, 480), (640, 480), True),
        (300, 280, 0.05, (800, 600), (1024, 768), True),
        (1000, 1000, 0.03, (640, 480), (700, 500), True),
        (1000, 997, 0.10, (512, 512), (512, 512), False),
        (64, 64, 0.0, (320, 240), (320, 240), True),          # identity: every keypoint meets its own copy
        (5, 400, 0.5, (640, 480), (640, 480), True),          # strong warp, few survivors
    ]
    out["n_cases"] = np.int64(len(cases))
    for c, (n0, n1, strength, (w0, h0), (w1, h1), has_resize) in enumerate(cases):
        size = 512
        hm = homography(rng, strength, w0, h0)
        hinv = np.linalg.inv(hm).astype(np.float32)           # datasets/hpatches.py:79
        kps0 = keypoints(rng, n0, size)
        if strength == 0.0:
            kps1 = kps0.copy()
        else:
            kps1 = keypoints(rng, n1, size)
            # plant true correspondences: a share of kps1 are warps of kps0 (plus sub-pixel noise)
            xy = kps0[: min(n0, n1) // 2, :2] * np.array([w1 - 1, h1 - 1], np.float32)
            q = (hm.astype(np.float64) @ np.concatenate([xy, np.ones((len(xy), 1))], 1).T).T
            q = q[:, :2] / q[:, 2:]
            q = q / np.array([w1 - 1, h1 - 1]) + rng.normal(0, 0.002, q.shape)
            ok = (q[:, 0] > 0) & (q[:, 0] < 1) & (q[:, 1] > 0) & (q[:, 1] < 1)
            kps1[: len(q)][ok, :2] = q[ok].astype(np.float32)
        w01 = {"mode": "homo", "width": torch.tensor(w1), "height": torch.tensor(h1), "homography_matrix": torch.from_numpy(hm)}
        w10 = {"mode": "homo", "width": torch.tensor(w0), "height": torch.tensor(h0), "homography_matrix": torch.from_numpy(hinv)}
        if has_resize:
            w01["resize"] = torch.tensor(size)
            w10["resize"] = torch.tensor(size)
        t0, t1 = torch.from_numpy(kps0), torch.from_numpy(kps1)
        a, b, ids, ids_out = proj.warp(t0, w01)
        a1, b1, ids1, ids1_out = proj.warp(t1, w10)
        res = rep.val_key_points(t0, t1, w01, w10, th=3)
        p = "c%d_" % c
        out[p + "kps0"], out[p + "kps1"], out[p + "hm"], out[p + "hinv"] = kps0, kps1, hm, hinv
        out[p + "wh0"], out[p + "wh1"] = np.array([w0, h0]), np.array([w1, h1])
        out[p + "resize"] = np.int64(size if has_resize else 0)
        out[p + "k0v"], out[p + "k01v"], out[p + "ids"], out[p + "ids_out"] = a.numpy(), b.numpy(), ids.numpy(), ids_out.numpy()
        out[p + "k1v"], out[p + "k10v"], out[p + "ids1"], out[p + "ids1_out"] = a1.numpy(), b1.numpy(), ids1.numpy(), ids1_out.numpy()
        out[p + "num_feat"] = np.int64(res["num_feat"])
        out[p + "repeatability"] = np.float32(res["repeatability"])
        out[p + "mean_error"] = np.float32(res["mean_error"])
        out[p + "errors"] = res["errors"].numpy() if res["errors"] is not None else np.zeros(0, np.float32)
        if len(a) and len(a1):
            # the mutual cells themselves, through the reference's own helpers in val_key_points' order (69-75)
            d = (rep.compute_keypoints_distance(a, b1) + rep.compute_keypoints_distance(a1, b).t()) / 2.
            im = torch.arange(min(d.shape))
            d[im, im] = 99999
            i, j = rep.mutual_argmin(d)
            out[p + "pairs"] = torch.stack([i, j], 1).numpy()
            out[p + "dist"] = (d[i, j] * (w01["resize"] if has_resize else w01["width"])).numpy()
        print(p, "kept", len(a), len(a1), "num_feat", res["num_feat"], "rep", float(res["repeatability"]), "err", float(res["mean_error"]))
    np.savez_compressed(os.path.join(HERE, "covis.npz"), **out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
