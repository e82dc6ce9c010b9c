"""Drop-in for the reference's tasks/AUC.py `auc` (101-154), the metric of BASELINE configs[2] / [4]: detection, brute-force match
on ALL keypoints (no covisibility filter, 116-120), relative pose from the matches, max(angular translation error, rotation
error) against the ground-truth pose.  Everything up to (R, t) runs on the device (utils/mvg.py `estimate_pose`: cv2's
findEssentialMat + recoverPose restated, PARITY UNPINNED); the two angle errors are evaluated on the host in float64
numpy exactly as the reference writes them (66-84).  The reference's per-pair match plot (145-148) is not produced."""
import numpy as np
import torch

from ..utils.mvg import estimate_pose


def angle_error_mat(R1, R2):
    """tasks/AUC.py:66-69."""
    cos = (np.trace(np.dot(R1.T, R2)) - 1) / 2
    cos = np.clip(cos, -1., 1.)
    return np.rad2deg(np.abs(np.arccos(cos)))


def angle_error_vec(v1, v2):
    """tasks/AUC.py:72-74."""
    n = np.linalg.norm(v1) * np.linalg.norm(v2)
    return np.rad2deg(np.arccos(np.clip(np.dot(v1, v2) / n, -1.0, 1.0)))


def compute_pose_error(T_0to1, R, t):
    """tasks/AUC.py:77-84."""
    R_gt = T_0to1[:3, :3]
    t_gt = T_0to1[:3, 3]
    error_t = angle_error_vec(t, t_gt)
    error_t = np.minimum(error_t, 180 - error_t)
    error_R = angle_error_mat(R, R_gt)
    return error_t, error_R


def _np(v):
    return torch.as_tensor(v).detach().cpu().numpy()


def _row(rt, good, mask, warp01):
    """AUC.py:135-154 from the device outputs of one pair."""
    if int(good) == 0:                                     # estimate_pose returned None (40-41, 57-64)
        return {"AUC": 180, "inliers": 0}
    R, t = rt[:9].reshape(3, 3), rt[9:]
    err_t, err_R = compute_pose_error(_np(warp01["pose01"]), R, t)          # the pose keeps its dtype (float32 from the datasets), as in the reference
    return {"AUC": np.maximum(err_t, err_R), "inliers": np.sum(mask > 0)}


def auc(idx, img_0, score_map_0, desc_map_0, img_1, score_map_1, desc_map_1, warp01, warp10, params):
    """tasks/AUC.py:101-154.  Returns the reference's dict {'AUC': max(err_t, err_R), 'inliers': count}."""
    from ..utils.extracter import detection
    from ..utils.matcher import brute_force_matcher
    kps0 = detection(score_map_0, params["extractor_params"])                                   # 116-117
    kps1 = detection(score_map_1, params["extractor_params"])
    m_pts0, m_pts1 = brute_force_matcher(kps0, kps1, desc_map_0, desc_map_1, params["matcher_params"]["brute_force_params"])
    h0, w0 = score_map_0.shape[2], score_map_0.shape[3]
    h1, w1 = score_map_1.shape[2], score_map_1.shape[3]
    if m_pts0.shape[0] < 5:                                                                     # estimate_pose: `if len(kpts0) < 5: return None`
        return {"AUC": 180, "inliers": 0}
    rt, mask, good, _ = estimate_pose(m_pts0[:, 0:2], m_pts1[:, 0:2], [w0 - 1, h0 - 1, w1 - 1, h1 - 1], _np(warp01["intrinsics0"]),
                                      _np(warp01["intrinsics1"]), thresh=1., seed=0)      # cv::RNG restarts from (uint64)-1 at every cv2 call: seed 0 = that state
    return _row(rt[0].cpu().numpy(), good[0], mask[0].cpu().numpy(), warp01)


def auc_batch(pipe, items, params, indices=None):
    """`auc` for the pairs a PairPipeline run has just processed (pipe.m0 / m1 / k): two launches and one read-back for the batch."""
    f, B = len(items), pipe.B
    pad = lambda xs: xs + [xs[-1]] * (B - f)
    K0 = np.stack(pad([_np(it["warp01_params"]["intrinsics0"]) for it in items]))
    K1 = np.stack(pad([_np(it["warp01_params"]["intrinsics1"]) for it in items]))
    rt, mask, good, _ = estimate_pose(pipe.m0, pipe.m1, [pipe.W - 1, pipe.H - 1, pipe.W - 1, pipe.H - 1], K0, K1, thresh=1., k_dev=pipe.k)       # seed 0 for every pair: OpenCV's RNG state at every call
    rt, mask, good = rt.cpu().numpy(), mask.cpu().numpy(), good.cpu().numpy()

    def rows():                 # the host half (135-154), on host copies: the runner runs it under the next batch's kernels
        out = []
        for b in range(f):
            r = _row(rt[b], good[b], mask[b], items[b]["warp01_params"])
            out.append([float(r["AUC"]), float(r["inliers"])])
        return out
    return rows
