"""Drop-in for the reference's tasks/FundamentalMatrix.py: `fundamental_matrix` (89-161), the task of BASELINE configs[3]
(XFeat + brute-force match on TartanAir): detection on both score maps, one of the three matcher branches, then the
epipolar residual of the matches against the ground-truth fundamental matrix -- all on the device
(csrc/geometry.hip `epipolar_error`); three numbers come back to the host per pair.  And `fundamental_matrix_ransac`
(12-86, task_type FundamentalMatrixRansac of config/long_term.yaml): the same front end, then the share of keypoints that
survive cv2.findFundamentalMat(FM_RANSAC) -- restated on the device (utils/mvg.py here; PARITY UNPINNED, OpenCV absent)."""
import torch

from .._lib import Context, ptr


def epipolar_error(kps0, kps1, fundamental, W, H, mode1, th, k_dev=None):
    """FundamentalMatrix.py:137-161 for B pairs: kps0 [B,K,c0] normalised rows, kps1 [B,K,c1] as the matcher branch
    left them (mode1 0: used as they are; 1: scaled to pixels, 1 appended; 2: pixels, 1 appended), fundamental [B,3,3].
    Returns (errors [B,K], stats [B,3] = mean error, ratio under th, count under th)."""
    dev = kps0.device
    a = kps0.detach().to(torch.float32).contiguous()
    b = kps1.detach().to(torch.float32).contiguous()
    if a.dim() == 2:
        a, b = a[None], b[None]
    B, K = a.shape[0], a.shape[1]
    f = torch.as_tensor(fundamental, dtype=torch.float32, device=dev).reshape(B, 9).contiguous()
    err = torch.empty((B, K), dtype=torch.float32, device=dev)
    stats = torch.empty((B, 3), dtype=torch.float32, device=dev)
    ctx = Context.get(dev)
    ctx.check(ctx.lib.kpb_epipolar_error(ctx.handle, ptr(a), a.shape[2], ptr(b), b.shape[2], B, K, ptr(k_dev), ptr(f), int(W), int(H),
                                         int(mode1), float(th), ptr(err), ptr(stats)))
    return err, stats


def in_contract(step, last_img, batch, score_map_0, score_map_1, desc_map_0, desc_map_1, matcher, params):
    """What keypoint_bench_amd.shim routes to the reference's own function instead."""
    if not (torch.is_tensor(score_map_0) and score_map_0.is_cuda):
        return "score maps are not on a HIP device"
    if params["matcher_params"]["type"] not in ("optical_flow", "brute_force", "light_glue"):
        return "matcher type"
    return None


def match_branch(kps0, kps1, score_map_0, desc_map_0, desc_map_1, matcher, params):
    """FundamentalMatrix.py:116-135.  Returns (kps0 rows, kps1 rows, mode1 for epipolar_error)."""
    from ..utils.matcher import brute_force_matcher, optical_flow_tensor
    mp = params["matcher_params"]
    h, w = score_map_0.shape[2], score_map_0.shape[3]
    if mp["type"] == "optical_flow":      # desc_map_* are the two images here (model_interface.py:262-267)
        kps1 = optical_flow_tensor(kps0[:, 0:2], kps0[:, 0:2], desc_map_0, desc_map_1, mp["optical_flow_params"])
        return kps0, kps1[0], 2
    if mp["type"] == "brute_force" or matcher is None:
        kps0, kps1 = brute_force_matcher(kps0, kps1, desc_map_0, desc_map_1, mp["brute_force_params"])
        return kps0, kps1, 0
    kps0, kps1 = matcher.match(kps0, kps1, desc_map_0, desc_map_1, {"w": w, "h": h})
    return kps0, kps1, 1


def fundamental_matrix(step, last_img, batch, score_map_0, score_map_1, desc_map_0, desc_map_1, matcher, params):
    """tasks/FundamentalMatrix.py:89-161.  Returns the reference's dict: fundamental_error (0-dim tensor), fundamental_radio
    (float), fundamental_num (int)."""
    from ..utils.extracter import detection
    kps0 = detection(score_map_0, params["extractor_params"])                   # 112-113
    kps1 = detection(score_map_1, params["extractor_params"])
    kps0, kps1, mode1 = match_branch(kps0, kps1, score_map_0, desc_map_0, desc_map_1, matcher, params)
    k = kps0.shape[0]
    if k == 0:                      # the reference divides by error.shape[0] (159)
        raise ZeroDivisionError("division by zero")
    f = batch["fundamental"][0]
    _, stats = epipolar_error(kps0, kps1, f[None], score_map_0.shape[3], score_map_0.shape[2], mode1,
                              params["FundamentalMatrix_params"]["th"])
    s = stats[0].cpu()
    return {"fundamental_error": s[0], "fundamental_radio": float(s[2]) / k, "fundamental_num": int(s[2])}


def _ransac_branch(kps0, kps1, score_map_0, desc_map_0, desc_map_1, matcher, params):
    """FundamentalMatrix.py:52-69: like match_branch, but every branch leaves three columns."""
    from ..utils.matcher import brute_force_matcher, optical_flow_tensor
    mp = params["matcher_params"]
    h, w = score_map_0.shape[2], score_map_0.shape[3]
    if mp["type"] == "optical_flow":
        k1 = optical_flow_tensor(kps0[:, 0:2], kps0[:, 0:2], desc_map_0, desc_map_1, mp["optical_flow_params"])
        return kps0, torch.cat([k1[0], torch.ones(k1.shape[1], 1, device=k1.device)], dim=1)
    if mp["type"] == "brute_force" or (mp["type"] == "light_glue" and matcher is None):
        return brute_force_matcher(kps0, kps1, desc_map_0, desc_map_1, mp["brute_force_params"])
    if mp["type"] == "light_glue":
        return matcher.match(kps0, kps1, desc_map_0, desc_map_1, {"w": w, "h": h})
    return kps0, kps1           # any other type: the reference matches nothing and pairs the rows as they are


def ransac_in_contract(step, image_0, image_1, score_map_0, score_map_1, desc_map_0, desc_map_1, matcher, params):
    if not (torch.is_tensor(score_map_0) and score_map_0.is_cuda):
        return "score maps are not on a HIP device"
    if params["matcher_params"].get("save_result") or params["extractor_params"].get("save_result"):
        return "save_result draws with cv2"
    return None


def fundamental_matrix_ransac(step, image_0, image_1, score_map_0, score_map_1, desc_map_0, desc_map_1, matcher, params, seed=None):
    """tasks/FundamentalMatrix.py:12-86 without its cv2 drawing (save_result must be off).  Returns the reference's dict:
    fundamental_error 0, fundamental_radio = kept keypoints / detected keypoints, fundamental_num = kept keypoints.
    `seed` drives the RANSAC sampler (default 0: the state OpenCV's RNG has at every cv2 call)."""
    from ..utils.extracter import detection
    from ..utils.mvg import find_fundamental
    h, w = score_map_0.shape[2], score_map_0.shape[3]
    kps0 = detection(score_map_0, params["extractor_params"])                   # 32, 35
    kps1 = detection(score_map_1, params["extractor_params"])
    total_size = kps0.shape[0] + kps1.shape[0]
    kps0, kps1 = _ransac_branch(kps0, kps1, score_map_0, desc_map_0, desc_map_1, matcher, params)
    n = kps0.shape[0]
    if n < 8:                       # utils/mvg.py:13-15: no estimate, every match kept
        kept = n
    else:                           # 76-78: (x, y) scaled by (w - 1, h - 1) in fp32, then cv2.findFundamentalMat(FM_RANSAC)
        _, mask, info = find_fundamental(kps0[None, :, :-1], kps1[None, :, :-1], [w - 1, h - 1, w - 1, h - 1], seed=0 if seed is None else seed)
        found, kept = (int(v) for v in info[0, :2].cpu())
        if not found:               # cv2 returns (None, None) and utils/mvg.py:17 fails on it
            raise AttributeError("'NoneType' object has no attribute 'ravel'")
    valid_size = 2 * kept
    return {"fundamental_error": 0, "fundamental_radio": valid_size / total_size, "fundamental_num": valid_size}


def fundamental_ransac_batch(pipe, items, params, indices=None):
    """The FundamentalMatrixRansac rows [error 0, ratio, num] of a whole PairPipeline batch (brute-force branch): one
    launch of the 7-point RANSAC over the batch's matches; every pair samples from OpenCV's per-call RNG state (seed 0)."""
    from ..utils.mvg import find_fundamental
    B, f = pipe.B, len(items)
    H, W = pipe.H, pipe.W
    _, _, info = find_fundamental(pipe.m0, pipe.m1, [W - 1, H - 1, W - 1, H - 1], k_dev=pipe.k)
    info = info.cpu().numpy()
    n, k = pipe.n.cpu().numpy(), pipe.k.cpu().numpy()
    rows = []
    for b in range(f):
        if k[b] >= 8 and not info[b, 0]:
            raise AttributeError("'NoneType' object has no attribute 'ravel'")
        kept = int(info[b, 1]) if k[b] >= 8 else int(k[b])
        rows.append([0.0, 2.0 * kept / float(n[b] + n[B + b]), 2.0 * kept])
    return rows
