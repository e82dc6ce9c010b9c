"""Literal numpy restatement of utils/extracter.py (TEST INFRASTRUCTURE; small inputs only).

Kept deliberately close to the reference's tensor program so that it can be read side by side with
utils/extracter.py; the C oracle (kpb_oracle.c) is cross-checked against it in tests/.
"""
import numpy as np


def fast_nms(score_hw, nms_dist):
    """utils/extracter.py:6-100.  unfold -> argmax == midpoint -> fold -> masked_fill, to a fixed count."""
    img = np.array(score_hw, dtype=np.float32, copy=True)
    if nms_dist == 0:
        return img, 0
    r = nms_dist
    ks = 2 * r + 1
    mid = (ks * ks) // 2
    H, W = img.shape
    count = None
    rounds = 0
    while True:
        pad = np.zeros((H + 2 * r, W + 2 * r), np.float32)  # F.unfold zero padding (extracter.py:54-60)
        pad[r:r + H, r:r + W] = img
        win = np.lib.stride_tricks.sliding_window_view(pad, (ks, ks)).reshape(H, W, ks * ks)
        mask = win.argmax(axis=2) == mid  # first index of the max (extracter.py:69-70)
        rounds += 1
        new_count = int(mask.sum())
        if new_count == count:  # extracter.py:77-78
            break
        count = new_count
        # F.fold of the mask expanded over all window channels but the centre (extracter.py:81-93):
        # fold[q] = number of maxima p != q with q inside p's window.
        mpad = np.zeros((H + 2 * r, W + 2 * r), np.int32)
        mpad[r:r + H, r:r + W] = mask
        mwin = np.lib.stride_tricks.sliding_window_view(mpad, (ks, ks))
        fold = mwin.sum(axis=(2, 3)) - mask
        img = np.where(fold > 0, np.float32(0.0), img)  # extracter.py:96
    return img, rounds


def detection(score_hw, params):
    """utils/extracter.py:193-221 with the documented tie rule (score desc, raster index asc)."""
    m, _ = fast_nms(score_hw, params["nms_dist"])
    b = params["border_dist"]
    H, W = m.shape
    if b > 0:  # extracter.py:177-188
        m[:, :b] = 0
        m[:, -b:] = 0
        m[:b, :] = 0
        m[-b:, :] = 0
    ys, xs = np.nonzero(m > np.float32(params["threshold"]))  # raster order (extracter.py:148-155)
    sc = m[ys, xs]
    idx = (ys * W + xs).astype(np.int64)
    if len(sc) > params["top_k"]:  # extracter.py:217-218
        order = np.lexsort((idx, -sc.astype(np.float64)))[: params["top_k"]]
        ys, xs, sc, idx = ys[order], xs[order], sc[order], idx[order]
    if params["min_score"] > 0:  # extracter.py:219-220
        keep = sc > np.float32(params["min_score"])
        ys, xs, sc, idx = ys[keep], xs[keep], sc[keep], idx[keep]
    x = (xs.astype(np.float32) + np.float32(0.5)) / np.float32(W)
    y = (ys.astype(np.float32) + np.float32(0.5)) / np.float32(H)
    return np.stack([x, y, sc.astype(np.float32)], axis=1).reshape(-1, 3), idx.astype(np.int32)


def greedy_nms(score_hw, nms_dist):
    """The closed form the HIP path relies on ('NMS fixed point'): for a non-negative
    map the fixed point of fast_nms keeps exactly the pixels chosen by greedy suppression in
    (score descending, raster index ascending) order with Chebyshev radius nms_dist."""
    img = np.asarray(score_hw, dtype=np.float32)
    H, W = img.shape
    out = np.zeros_like(img)
    if nms_dist == 0:
        return img.copy()
    flat = img.ravel()
    order = np.lexsort((np.arange(flat.size), -flat.astype(np.float64)))
    blocked = np.zeros((H, W), bool)
    r = nms_dist
    for i in order:
        if not flat[i] > 0:
            break
        y, x = divmod(int(i), W)
        if blocked[y, x]:
            continue
        out[y, x] = flat[i]
        blocked[max(0, y - r):y + r + 1, max(0, x - r):x + r + 1] = True
    return out
