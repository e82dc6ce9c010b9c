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


def preprocess(img_u8, size=None, bgr=False):
    """datasets/hpatches.py:47-69 after decoding, on one uint8 [H, W, 3] image: BGR -> RGB (59-60), astype(float32) / 255
    (59-60), cv2.resize(img, (w, h)) with the default INTER_LINEAR (66-67), HWC -> CHW (74-75).  PARITY UNPINNED: cv2 is
    not installed here; the resize restates OpenCV's INTER_LINEAR arithmetic for float32 images (resize.cpp:
    f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s, clamped; horizontal blend, then vertical blend)."""
    a = np.asarray(img_u8)
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3
    if bgr:
        a = a[:, :, ::-1]
    f = a.astype(np.float32) / np.float32(255.0)
    Hs, Ws = f.shape[:2]
    Hd, Wd = (Hs, Ws) if size is None else ((size, size) if isinstance(size, int) else size)

    def coords(n_dst, n_src):
        d = np.arange(n_dst, dtype=np.float64)
        fr = ((d + 0.5) * (float(n_src) / float(n_dst)) - 0.5).astype(np.float32)
        s = np.floor(fr).astype(np.int64)
        fr = fr - s.astype(np.float32)
        lo = s < 0
        s[lo] = 0; fr[lo] = 0
        hi = s >= n_src - 1
        s[hi] = n_src - 1; fr[hi] = 0
        return s, np.minimum(s + 1, n_src - 1), fr.astype(np.float32)

    sx, sx1, fx = coords(Wd, Ws)
    sy, sy1, fy = coords(Hd, Hs)
    a0, a1 = (np.float32(1) - fx)[None, :, None], fx[None, :, None]
    rows = f[:, sx] * a0 + f[:, sx1] * a1                                   # horizontal pass, all source rows
    b0, b1 = (np.float32(1) - fy)[:, None, None], fy[:, None, None]
    out = rows[sy] * b0 + rows[sy1] * b1                                    # vertical pass
    return np.ascontiguousarray(out.transpose(2, 0, 1).astype(np.float32))
