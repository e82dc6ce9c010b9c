#!/usr/bin/env python3
"""Golden fixtures for the depth-based covisibility warp (utils/projection.py:195-268 warp_se3), produced by the REFERENCE.

Runs only in the build container.  utils/projection.py is imported as it is (cv2, absent here, supplied as a blank module:
warp_se3 is pure torch).  Scenes are synthetic: smooth positive depth maps with holes, a small rigid motion, pinhole
intrinsics, crop offsets -- stored as seeds plus the few numbers that define them, next to the reference's outputs.

Usage:  python tests/golden/make_golden_se3.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def scene(seed, H0, W0, H1, W1, planar=False):
    """Deterministic synthetic scene (also used by the tests to rebuild the inputs from the seed).  planar: a fronto-parallel
    wall seen under a pure sideways translation, so that depths agree between the views and most points are covisible;
    otherwise two unrelated wavy depth maps, so that most points come out occluded or outside."""
    rng = np.random.default_rng(seed)
    def depth(H, W):
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        d = 4.0 + 0.004 * xx + 0.002 * yy + 0.3 * np.sin(xx / 37.0) * np.cos(yy / 29.0)
        if planar:
            d = np.full((H, W), 5.0, np.float32) + rng.uniform(-0.02, 0.02, (H, W)).astype(np.float32)
        holes = rng.random((H, W)) < 0.03
        d[holes] = 0.0
        d[H // 3: H // 3 + 25, W // 4: W // 4 + 40] = 0.0                  # a block without depth
        return d.astype(np.float32)
    d0, d1 = depth(H0, W0), depth(H1, W1)
    k0 = np.array([[520.0, 0, W0 / 2 + 3.5], [0, 515.0, H0 / 2 - 2.25], [0, 0, 1]], np.float32)
    k1 = np.array([[505.0, 0, W1 / 2 - 1.5], [0, 512.0, H1 / 2 + 4.75], [0, 0, 1]], np.float32)
    ang = rng.normal(0, 0.03, 3)
    rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
    ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
    rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
    pose = np.eye(4)
    pose[:3, :3] = rz @ ry @ rx
    pose[:3, 3] = rng.normal(0, 0.15, 3)
    if planar:
        pose = np.eye(4)
        pose[:2, 3] = rng.normal(0, 0.3, 2)
    bbox0 = np.array([12.0, 7.0], np.float32)            # (row, col) of the crop in the full image
    bbox1 = np.array([5.0, 21.0], np.float32)
    return d0, d1, k0, k1, pose.astype(np.float32), bbox0, bbox1


def main():
    if not os.path.isdir(REF):
        print("reference checkout not present; nothing to do")
        return 0
    sys.dont_write_bytecode = True
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import torch
    import utils.projection as proj
    torch.set_num_threads(1)
    cases = [(1000, 480, 640, 480, 640, True), (1000, 480, 640, 480, 640, False), (300, 384, 512, 480, 640, True),
             (60, 240, 320, 240, 320, True), (20, 240, 320, 240, 320, True), (8, 240, 320, 240, 320, False)]
    out = {"n_cases": np.int64(len(cases))}
    rng = np.random.default_rng(99)
    for c, (n, H0, W0, H1, W1, planar) in enumerate(cases):
        d0, d1, k0, k1, pose, bbox0, bbox1 = scene(500 + c, H0, W0, H1, W1, planar)
        kps = np.concatenate([rng.uniform(0.0, 1.0, (n, 2)), rng.random((n, 1))], 1).astype(np.float32)
        t = torch.from_numpy
        prm = {"mode": "se3", "pose01": t(pose), "bbox0": t(bbox0), "bbox1": t(bbox1), "depth0": t(d0), "depth1": t(d1),
               "intrinsics0": t(k0), "intrinsics1": t(k1)}
        a, b, ids, ids_out = proj.warp(t(kps), prm)
        k = "c%d_" % c
        out[k + "scene"] = np.array([500 + c, H0, W0, H1, W1, int(planar)], np.int64)
        out[k + "kps"] = kps
        out[k + "kinv0"] = torch.inverse(t(k0)).numpy()               # what unproject (43) computes
        out[k + "k0v"], out[k + "k01v"], out[k + "ids"], out[k + "ids_out"] = a.numpy(), b.numpy(), ids.numpy(), ids_out.numpy()
        print(k, "n", n, "valid", len(ids), "out", len(ids_out))
    np.savez_compressed(os.path.join(HERE, "se3.npz"), **out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
