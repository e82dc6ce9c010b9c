"""Drop-in for the reference's tasks/visual_odometer.py `visual_odometry` (10-91, task_type visual_odometer on the
sequence datasets): detection on two consecutive frames, the brute-force or LightGlue matcher, the relative pose of the
matches -- cv2.findEssentialMat(focal, pp, RANSAC, prob 0.999, threshold 1.0) + cv2.recoverPose(focal, pp) restated on
the device (utils/mvg.py `estimate_pose`; PARITY UNPINNED, OpenCV absent) -- and the chained pose update with the
ground-truth step length (79-89).  The match plot (70-73, commented out in the reference) is not produced.

SURVEY 8(e): the chain `r_est[-1]` makes the task sequential, but only in its last two lines; everything before them is
per frame.  `relative_motion` / `relative_motion_batch` return the per-frame (R, t, scale) rows, `compose` chains them
(model_interface.py:283-298 + 161-166), so that a sharded run gathers rows and composes once.

The optical-flow branch of the reference calls cv2's pyramidal LK (`optical_flow_cv`, utils/matcher.py); that branch is
outside this library's contract and is left to the reference (keypoint_bench_amd.shim routes it there)."""
import numpy as np
import torch

from ..utils.mvg import estimate_pose

PROB = 0.999            # visual_odometer.py:75
THRESHOLD = 1.0         # pixels; cv2 divides by the focal length
DISTANCE = 50.0         # cv2.recoverPose's distanceThresh when none is passed (76-77)


def _position(p):
    """First three numbers of a pose: pypose LieTensors answer .tensor(), arrays and tensors are taken as they are (81-83)."""
    t = p.tensor() if hasattr(p, "tensor") and callable(p.tensor) else torch.as_tensor(np.asarray(p) if not torch.is_tensor(p) else p)
    return t.reshape(-1)[0:3].detach().cpu().to(torch.float64)


def step_length(batch):
    """visual_odometer.py:80-83."""
    return float(torch.norm(_position(batch["ground_truth"]) - _position(batch["last_ground_truth"])))


def _scalar(v):
    return float(torch.as_tensor(v).reshape(-1)[0])


def in_contract(step, pose_R, pose_t, last_img, batch, score_map_0, score_map_1, desc_map_0, desc_map_1, matcher, params):
    if not (torch.is_tensor(score_map_0) and score_map_0.is_cuda):
        return "score maps are not on a HIP device"
    if params["matcher_params"]["type"] == "optical_flow":
        return "optical_flow_cv is cv2's tracker"
    return None


def match_branch(kps0, kps1, score_map_0, desc_map_0, desc_map_1, matcher, params):
    """visual_odometer.py:43-61."""
    from ..utils.matcher import brute_force_matcher
    mp = params["matcher_params"]
    h, w = score_map_0.shape[2], score_map_0.shape[3]
    if mp["type"] == "optical_flow":
        raise NotImplementedError("visual_odometry: the optical_flow branch is cv2's tracker (optical_flow_cv); not part of this library")
    if mp["type"] == "brute_force" or (mp["type"] == "light_glue" and matcher is None):
        return brute_force_matcher(kps0, kps1, desc_map_0, desc_map_1, mp["brute_force_params"])
    if mp["type"] == "light_glue":
        k0, k1 = matcher.match(kps0, kps1, desc_map_0, desc_map_1, {"w": w, "h": h})
        return k0, k1[:, 0:2]
    return kps0, kps1


def _camera(fx, cx, cy):
    return np.array([[fx, 0.0, cx], [0.0, fx, cy], [0.0, 0.0, 1.0]], np.float64)      # focal= / pp=: one focal length, float64


def relative_motion(step, last_img, batch, score_map_0, score_map_1, desc_map_0, desc_map_1, matcher, params, seed=None):
    """visual_odometer.py:37-78 for one frame pair: (R [3,3], t [3,1]) float64 numpy, as cv2.recoverPose returns them."""
    from ..utils.extracter import detection
    kps0 = detection(score_map_0, params["extractor_params"])
    kps1 = detection(score_map_1, params["extractor_params"])
    kps0, kps1 = match_branch(kps0, kps1, score_map_0, desc_map_0, desc_map_1, matcher, params)
    if kps0.shape[0] < 5:
        raise RuntimeError("visual_odometry: fewer than five matches (cv2.findEssentialMat raises there)")
    sc = [score_map_0.shape[3] - 1, score_map_0.shape[2] - 1, last_img.shape[3] - 1, last_img.shape[2] - 1]      # 65-67
    K = _camera(_scalar(batch["fx"]), _scalar(batch["cx"]), _scalar(batch["cy"]))
    rt, _, _, info = estimate_pose(kps0[None, :, 0:2], kps1[None, :, 0:2], sc, K, K, thresh=THRESHOLD, conf=PROB,
                                   seed=0 if seed is None else seed, recover_all=True, dist=DISTANCE)
    return _rt(rt[0].cpu().numpy(), int(info[0, 0]))


def _rt(rt, found):
    if not found:               # no model from RANSAC: the frame contributes no motion
        return np.eye(3), np.zeros((3, 1))
    return rt[:9].reshape(3, 3).copy(), rt[9:].reshape(3, 1).copy()


def update(pose_R, pose_t, R, t, scale):
    """visual_odometer.py:84-91: the pose only moves when the ground-truth step is at least a millimetre."""
    if scale >= 0.001:
        return pose_R.dot(R), pose_t + float(scale) * pose_R.dot(t)
    return pose_R, pose_t


def visual_odometry(step, pose_R, pose_t, last_img, batch, score_map_0, score_map_1, desc_map_0, desc_map_1, matcher, params, seed=None):
    """tasks/visual_odometer.py:10-91.  Returns the reference's dict {'R': R_est, 't': t_est}."""
    R, t = relative_motion(step, last_img, batch, score_map_0, score_map_1, desc_map_0, desc_map_1, matcher, params, seed)
    R_est, t_est = update(pose_R, pose_t, R, t, step_length(batch))
    return {"R": R_est, "t": t_est}


def relative_motion_batch(pipe, items, indices=None):
    """Rows [R (9), t (3), step length] of the frames a SequencePipeline run has just matched (slot j = frame pair
    (j-1, j)): two launches for the chunk."""
    f = len(items)
    F = pipe.m0.shape[0]
    pad = lambda xs: xs + [xs[-1]] * (F - f)
    K = np.stack(pad([_camera(_scalar(it["fx"]), _scalar(it["cx"]), _scalar(it["cy"])) for it in items]))
    rt, _, _, info = estimate_pose(pipe.m0, pipe.m1, [pipe.W - 1, pipe.H - 1, pipe.W - 1, pipe.H - 1], K, K, thresh=THRESHOLD, conf=PROB,
                                   k_dev=pipe.k, recover_all=True, dist=DISTANCE)
    rt, info, kk = rt.cpu().numpy(), info.cpu().numpy(), pipe.k.cpu().numpy()
    rows = []
    for j in range(f):
        if kk[j] < 5:
            raise RuntimeError("visual_odometry: fewer than five matches (cv2.findEssentialMat raises there)")
        R, t = _rt(rt[j], int(info[j, 0]))
        rows.append(list(R.reshape(9)) + list(t.reshape(3)) + [step_length(items[j])])
    return rows


def compose(rows):
    """model_interface.py:116-117 + 296-297: r_est = [I], t_est = [0]; every frame appends the updated pose.
    rows [n, 13] float64 -> (r_est [n+1, 3, 3], t_est [n+1, 3, 1])."""
    r_est, t_est = [np.eye(3)], [np.zeros((3, 1))]
    for row in np.asarray(rows, np.float64):
        R, t = update(r_est[-1], t_est[-1], row[:9].reshape(3, 3), row[9:12].reshape(3, 1), row[12])
        r_est.append(R)
        t_est.append(t)
    return np.stack(r_est), np.stack(t_est)
