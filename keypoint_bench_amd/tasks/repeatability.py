"""Drop-in for the metric core of the reference's tasks/repeatability.py: `val_key_points` (54-92) with
`compute_keypoints_distance` (39-51) and `mutual_argmax/argmin` (9-36), computed by csrc/covis.hip.

The reference materialises three M x N matrices per pair; here the pair never leaves the GPU until the handful of
numbers the metric needs are read back."""
import torch

from .._lib import Context, ptr
from ..utils.projection import _scalar, warp_homography_device, warp


def _scale(w):
    return float(_scalar(w["resize"] if "resize" in w else w["width"]))       # repeatability.py:76-81


def gt_mutual(k0, k01, k1, k10, scale01, scale10, th=3.0, m_dev=None, n_dev=None, cap=None):
    """Lines 69-85 for covisible sets already on the device.  k0/k01 [M,2], k1/k10 [N,2]; m_dev/n_dev optional
    int32[1] device counts (rows beyond them ignored).  Returns (pairs[K,2] int64, dist[K], errors[M], gt_num)."""
    dev = k0.device
    M, N = k0.shape[0], k1.shape[0]
    ctx = Context.get(dev)
    scale = torch.tensor([scale01, scale10], dtype=torch.float32, device=dev)
    errors = torch.empty((M,), dtype=torch.float32, device=dev)
    counts = torch.zeros((2,), dtype=torch.int32, device=dev)
    cap = cap or (M + N + 1024)
    while True:
        pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        dist = torch.empty((cap,), dtype=torch.float32, device=dev)
        ctx.check(ctx.lib.kpb_val_keypoints(ctx.handle, ptr(k0), ptr(k01), ptr(k1), ptr(k10), 1, M, N, ptr(m_dev),
                                            ptr(n_dev), ptr(scale), float(th), ptr(pairs), ptr(dist), cap, ptr(errors),
                                            ptr(counts)))
        total, gt = (int(v) for v in counts.tolist())
        if total <= cap:
            break
        cap = total                     # every tie is a mutual cell (mutual_argmax keeps them all): grow and redo
    return pairs[:total].to(torch.int64), dist[:total], errors, gt


def val_key_points(kps0, kps1, warp01, warp10, th: int = 3):
    """tasks/repeatability.py:54-92 (homography and se3 warps)."""
    num_feat = min(kps0.shape[0], kps1.shape[0])
    if warp01["mode"] == "homo" and warp10["mode"] == "homo":       # counts stay on the device until both warps are queued
        a, b, _, na = warp_homography_device(kps0[:, 0:2], warp01)
        a1, b1, _, nb = warp_homography_device(kps1[:, 0:2], warp10)
        M, N = int(na.item()), int(nb.item())
        a, b, a1, b1 = a[:M], b[:M], a1[:N], b1[:N]
    else:
        a, b, _, _ = warp(kps0, warp01)
        a1, b1, _, _ = warp(kps1, warp10)
        M, N = a.shape[0], a1.shape[0]
    if M == 0 or N == 0:
        return {"num_feat": 0, "repeatability": 0, "mean_error": 0, "errors": None}
    pairs, dist, errors, gt = gt_mutual(a, b, a1, b1, _scale(warp01), _scale(warp10), th)
    error = dist[dist <= th].cpu().numpy()                                   # 83
    return {
        "num_feat": num_feat,
        "repeatability": torch.tensor(gt) / num_feat,                          # 82, 89 (int64 / int -> float32)
        "mean_error": error.mean(),                                            # 84 (nan when nothing is within th)
        "errors": errors,
    }


def repeatability(idx, img_0, score_map_0, img_1, score_map_1, warp01, warp10, params):
    """tasks/repeatability.py:95-122 without its plotting (cv2.imwrite of the keypoint overlays, 114-119): detection on
    both score maps, then val_key_points.  Returns the reference's dict."""
    from ..utils.extracter import detection
    kps0 = detection(score_map_0, params["extractor_params"])
    kps1 = detection(score_map_1, params["extractor_params"])
    return val_key_points(kps0, kps1, warp01, warp10, th=params["repeatability_params"]["th"])


def homography_tables(warps, dev):
    """Stacks the 'homo' warp dicts of B pairs: (hmat [B, 9] fp32, wh [B, 2] int32 = width, height, scale [B])."""
    import numpy as np

    def mat(v):     # numpy items stay in numpy (one host array, one upload per table instead of a tensor per pair)
        return (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)).astype(np.float32, copy=False).reshape(9)
    hm = torch.from_numpy(np.stack([mat(w["homography_matrix"]) for w in warps])).to(dev)
    wh = torch.from_numpy(np.array([[_scalar(w["width"]), _scalar(w["height"])] for w in warps], np.int32)).to(dev)
    return hm.contiguous(), wh.contiguous(), [_scale(w) for w in warps]


def repeatability_batch(kps, n, warps01, warps10, th):
    """val_key_points (54-92) for B pairs in four launches and one read-back.  kps [2B, K, 3] / n [2B] as
    PairPipeline leaves them (rows 0..B-1 image 0, B..2B-1 image 1); warps01 / warps10: lists of B 'homo' warp dicts.
    Returns B rows [num_feat, repeatability, mean_error] equal to the single-pair path's."""
    import numpy as np
    dev = kps.device
    B, K = len(warps01), kps.shape[1]
    ctx = Context.get(dev)
    hm01, wh01, s01 = homography_tables(warps01, dev)
    hm10, wh10, s10 = homography_tables(warps10, dev)
    hm, wh = torch.cat([hm01, hm10]).contiguous(), torch.cat([wh01, wh10]).contiguous()
    k0 = torch.empty((2 * B, K, 2), dtype=torch.float32, device=dev)
    k01 = torch.empty((2 * B, K, 2), dtype=torch.float32, device=dev)
    ids = torch.empty((2 * B, K), dtype=torch.int32, device=dev)
    nc = torch.empty((2 * B,), dtype=torch.int32, device=dev)
    ctx.check(ctx.lib.kpb_warp_homography(ctx.handle, ptr(kps), 2 * B, K, 3, ptr(n), ptr(hm), ptr(wh), ptr(k0), ptr(k01), ptr(ids), ptr(nc)))
    scale = torch.tensor(list(zip(s01, s10)), dtype=torch.float32, device=dev).contiguous()
    cap = 2 * K + 1024
    pairs = torch.empty((B, cap, 2), dtype=torch.int32, device=dev)
    dist = torch.empty((B, cap), dtype=torch.float32, device=dev)
    errors = torch.empty((B, K), dtype=torch.float32, device=dev)
    counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    ctx.check(ctx.lib.kpb_val_keypoints(ctx.handle, ptr(k0[:B]), ptr(k01[:B]), ptr(k0[B:]), ptr(k01[B:]), B, K, K, ptr(nc[:B]), ptr(nc[B:]),
                                        ptr(scale), float(th), ptr(pairs), ptr(dist), cap, ptr(errors), ptr(counts)))
    n_h, nc_h, cnt_h, dist_h = n.cpu().numpy(), nc.cpu().numpy(), counts.cpu().numpy(), dist.cpu().numpy()
    rows = []
    for b in range(B):
        num_feat = int(min(n_h[b], n_h[B + b]))
        M, N = int(nc_h[b]), int(nc_h[B + b])
        if M == 0 or N == 0:
            rows.append([0.0, 0.0, 0.0])                                   # 61-67
            continue
        total, gt = int(cnt_h[b, 0]), int(cnt_h[b, 1])
        if total > cap:                 # more mutual ties than the shared capacity: this pair alone, with room
            _, d, _, gt = gt_mutual(k0[b, :M], k01[b, :M], k0[B + b, :N], k01[B + b, :N], s01[b], s10[b], th, cap=total)
            d = d.cpu().numpy()
        else:
            d = dist_h[b, :total]
        err = d[d <= th]
        rows.append([float(num_feat), float(np.float32(gt) / np.float32(num_feat)), float(err.mean()) if err.size else float("nan")])
    return rows
