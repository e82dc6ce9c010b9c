"""Drop-in for the reference's tasks/MHA.py `mha` (11-72), the metric of BASELINE configs[1]: detection, covisibility filter,
brute-force match, robust homography, corner error against the ground-truth homography, one hit flag per threshold.
Everything up to the 3x3 homography runs on the device; the corner error of four points is evaluated on the host in
float64 numpy exactly as the reference writes it (51-70).

The estimator is this library's RANSAC (utils/mvg.py; cv2.findHomography in the reference: PARITY UNPINNED, see there)."""
import numpy as np
import torch

from ..utils.mvg import find_homography
from ..utils.projection import _scalar


def corner_hits(H, real_H, h, w, resize_h, resize_w, th):
    """tasks/MHA.py:50-70 (the corner rows are (h-1, 0), (0, w-1) in the reference: kept)."""
    corners = np.array([[0, 0, 1], [h - 1, 0, 1], [0, w - 1, 1], [h - 1, w - 1, 1]])
    real_warped_corners = np.dot(corners, np.transpose(real_H))
    real_warped_corners = real_warped_corners[:, :2] / real_warped_corners[:, 2:]
    warped_corners = np.dot(corners, np.transpose(H))
    warped_corners = warped_corners[:, :2] / warped_corners[:, 2:]
    real_warped_corners = real_warped_corners * np.array([resize_h / h, resize_w / w])
    warped_corners = warped_corners * np.array([resize_h / h, resize_w / w])
    mean_dist = np.mean(np.linalg.norm(real_warped_corners - warped_corners, axis=1))
    return [float(mean_dist <= t) for t in th], mean_dist


def _hw(warp01):
    return np.asarray(_scalar(warp01["height"])), np.asarray(_scalar(warp01["width"]))        # 40: .cpu().numpy() 0-d arrays


def _raw_hw(img):
    """(H, W) of a dataset image as the reference's `mha` sees it: the UNCROPPED batch['image0'] (model_interface.py:249-251
    passes it, not the x32 crop the network ran on; MHA.py:59-60 reads img_0.shape[2:4]).  Decoded uint8 items are [H, W, 3]."""
    if getattr(img, "ndim", 0) == 3 and img.shape[-1] == 3 and str(getattr(img, "dtype", "")).endswith("uint8"):
        return int(img.shape[0]), int(img.shape[1])
    return int(img.shape[-2]), int(img.shape[-1])


def _real_h(warp01):
    return torch.as_tensor(warp01["homography_matrix"]).detach().cpu().numpy()


def mha(idx, img_0, score_map_0, desc_map_0, img_1, score_map_1, desc_map_1, warp01, warp10, params):
    """tasks/MHA.py:11-72.  Returns one float per MHA_params['th'] entry."""
    from ..utils.extracter import detection
    from ..utils.matcher import brute_force_matcher
    from ..utils.projection import warp
    th = params["MHA_params"]["th"]
    zeros = [0 for _ in th]
    kps0 = detection(score_map_0, params["extractor_params"])                                   # 30-31
    kps1 = detection(score_map_1, params["extractor_params"])
    kps0_cov, _, _, _ = warp(kps0, warp01)                                                     # 33-34
    kps1_cov, _, _, _ = warp(kps1, warp10)
    if kps0_cov.shape[0] == 0 or kps1_cov.shape[0] == 0:                                        # 35-36
        return zeros
    m_pts0, m_pts1 = brute_force_matcher(kps0_cov, kps1_cov, desc_map_0, desc_map_1, params["matcher_params"]["brute_force_params"])
    h, w = _hw(warp01)
    if m_pts0.shape[0] < 4:         # cv2.findHomography raises below 4 correspondences; no model -> no hit
        return zeros
    H, _, info = find_homography(m_pts0[:, 0:2], m_pts1[:, 0:2], [w - 1, h - 1, w - 1, h - 1], seed=0)   # 41-47: BOTH sides use warp01's size; seed 0 = cv::RNG((uint64)-1), OpenCV's state at every call
    if int(info[0, 0]) == 0:                                                                    # 48-49
        return zeros
    hits, _ = corner_hits(H[0].cpu().numpy(), _real_h(warp01), h, w, img_0.shape[2], img_0.shape[3], th)
    return hits


def mha_batch(pipe, items, params, indices=None):
    """`mha` for the pairs a PairPipeline run with the covisibility stage has just processed (pipe.matched(), pipe.k):
    one RANSAC launch for the batch, one read-back of the homographies.  Returns a callable that computes the rows."""
    th = params["MHA_params"]["th"]
    f, B = len(items), pipe.B
    hw = [_hw(it["warp01_params"]) for it in items]
    scale = torch.tensor([[w - 1, h - 1, w - 1, h - 1] for h, w in hw] + [[1, 1, 1, 1]] * (B - f), dtype=torch.float32)
    m0, m1 = pipe.matched()
    H, _, info = find_homography(m0, m1, scale, k_dev=pipe.k)         # seed 0 for every pair, as OpenCV restarts its RNG per call
    H, info = H.cpu().numpy(), info.cpu().numpy()
    raw = [_raw_hw(it["image0"]) for it in items]      # resize factors come from the uncropped image 0 (MHA.py:59-60)

    def rows():                 # the host half (50-70), on host copies: the runner runs it under the next batch's kernels
        out = []
        for b in range(f):
            if info[b, 0] == 0:
                out.append([0.0 for _ in th])
                continue
            h, w = hw[b]
            hits, _ = corner_hits(H[b], _real_h(items[b]["warp01_params"]), h, w, raw[b][0], raw[b][1], th)
            out.append(hits)
        return out
    return rows
