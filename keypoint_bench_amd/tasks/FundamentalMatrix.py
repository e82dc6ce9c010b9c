"""Drop-in for the reference's tasks/FundamentalMatrix.py `fundamental_matrix` (89-161), the task of BASELINE configs[3]
(XFeat + brute-force match on TartanAir): detection on both score maps, one of the three matcher branches, then the
epipolar residual of the matches against the ground-truth fundamental matrix -- all on the device
(csrc/geometry.hip `epipolar_error`); three numbers come back to the host per pair."""
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
