"""CPU parity oracle for the keypoint_bench hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``keypoint_bench_amd``) never does: it fails loudly when its HIP library is
missing instead of falling back to anything here.

Two layers:
  * ``kpb_oracle.c``  -- plain-C restatement (fast enough for 480x640), loaded through ctypes.
  * ``numpy_ref.py``  -- a literal numpy restatement used on small inputs to cross-check the C code.
  * ``alike_ref.py``  -- torch-fp32 functional restatement of models/ALike.py:136-164.
Each function cites the reference file:line it follows; pinning against the reference's own outputs
is done by tests/test_oracle_golden.py with the fixtures under tests/golden/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkpb_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "kpb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libkpb_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        dp = ctypes.POINTER(ctypes.c_double)
        L.kpbo_fast_nms.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.kpbo_fast_nms.restype = ctypes.c_int
        L.kpbo_detection.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_float, fp, ip, ctypes.c_int]
        L.kpbo_detection.restype = ctypes.c_int
        L.kpbo_sample.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_long,
                                  ctypes.c_long, fp, ctypes.c_int, ctypes.c_int, fp]
        L.kpbo_sample.restype = None
        L.kpbo_match.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                 ip, dp]
        L.kpbo_match.restype = ctypes.c_int
        L.kpbo_warp_homography.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp, ip]
        L.kpbo_warp_homography.restype = ctypes.c_int
        L.kpbo_val_keypoints.argtypes = [fp, fp, ctypes.c_int, fp, fp, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                         ip, fp, ctypes.c_long, fp]
        L.kpbo_val_keypoints.restype = ctypes.c_long
        L.kpbo_lk_track.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp, fp, ctypes.c_int, ctypes.c_float,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp]
        L.kpbo_lk_track.restype = None
        L.kpbo_warp_se3.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                    fp, fp, fp, fp, fp, ctypes.c_int, fp, fp, ip, ip, ip]
        L.kpbo_warp_se3.restype = None
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


DEFAULT_PARAMS = dict(nms_dist=4, threshold=0.0, border_dist=8, top_k=300, min_score=0.0)  # extracter.py:200-205


def fast_nms(score_hw, nms_dist):
    """utils/extracter.py:6-100 on one [H, W] map.  Returns (map, rounds)."""
    s = _f32(score_hw)
    out = np.empty_like(s)
    rounds = lib().kpbo_fast_nms(_fp(s), _fp(out), s.shape[0], s.shape[1], int(nms_dist))
    return out, rounds


def detection(score_hw, params=None):
    """utils/extracter.py:193-221 on one [H, W] map.  Returns (kps[N,3] float32, flat_idx[N] int32)."""
    p = dict(DEFAULT_PARAMS) if params is None else params
    s = _f32(score_hw)
    H, W = s.shape
    cap = H * W
    kps = np.empty((cap, 3), np.float32)
    idx = np.empty((cap,), np.int32)
    n = lib().kpbo_detection(_fp(s), H, W, int(p["nms_dist"]), float(p["threshold"]), int(p["border_dist"]),
                             int(p["top_k"]), float(p["min_score"]), _fp(kps),
                             idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), cap)
    return kps[:n].copy(), idx[:n].copy()


def sample(desc_chw, pts):
    """utils/matcher.py:221-226: desc [C, Hd, Wd] (any strides), pts [N, >=2] -> [N, C]."""
    d = np.asarray(desc_chw, dtype=np.float32)
    if not d.flags.c_contiguous:
        d = np.ascontiguousarray(d)
    C, Hd, Wd = d.shape
    p = _f32(pts)
    n = p.shape[0]
    out = np.empty((n, C), np.float32)
    if n:
        lib().kpbo_sample(_fp(d), C, Hd, Wd, Hd * Wd, Wd, 1, _fp(p), n, p.shape[1], _fp(out))
    return out


def match(d0, d1, max_distance=np.inf, cross_check=True):
    """skimage.feature.match_descriptors as called at utils/matcher.py:227-230 (restated, unpinned).
    Returns (pairs[K,2] int64, dist[K] float64)."""
    a, b = _f32(d0), _f32(d1)
    n, m = a.shape[0], b.shape[0]
    pairs = np.empty((max(min(n, m), n), 2), np.int32)
    dist = np.empty((max(n, 1),), np.float64)
    k = 0
    if n and m:
        k = lib().kpbo_match(_fp(a), n, _fp(b), m, a.shape[1], float(max_distance), int(bool(cross_check)),
                             pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                             dist.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return pairs[:k].astype(np.int64), dist[:k].copy()


def brute_force_matcher(pts0, pts1, desc0_chw, desc1_chw, params):
    """utils/matcher.py:206-234 end to end on numpy arrays."""
    assert params["metric"] == "euclidean"
    d0 = sample(desc0_chw, pts0)
    d1 = sample(desc1_chw, pts1)
    pairs, _ = match(d0, d1, params["max_distance"], params["cross_check"])
    return np.asarray(pts0)[pairs[:, 0]], np.asarray(pts1)[pairs[:, 1]]


def warp_homography(kps, hm, width, height, fused=-1):
    """utils/projection.py:137-167.  Returns (kps0_valid[K,2], kps01_valid[K,2], ids[K], ids_out[n-K]).
    fused=-1 follows the reference's torch build (BLAS/fused for n >= 45, unfused below); 1 = the form libkpb uses."""
    p = _f32(kps)
    n = p.shape[0]
    h = _f32(hm).reshape(9)
    a = np.empty((max(n, 1), 2), np.float32)
    b = np.empty((max(n, 1), 2), np.float32)
    ids = np.empty((max(n, 1),), np.int32)
    k = lib().kpbo_warp_homography(_fp(p), n, p.shape[1] if n else 2, _fp(h), int(width), int(height), int(fused), _fp(a), _fp(b),
                                   ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) if n else 0
    return a[:k].copy(), b[:k].copy(), ids[:k].astype(np.int64), ids[k:n].astype(np.int64)


def val_key_points(kps0, kps1, warp01, warp10, th=3, fused=-1):
    """tasks/repeatability.py:54-92 on numpy inputs (mode 'homo').  Returns the reference's dict, plus the mutual
    pairs and their scaled distances for the tests."""
    num_feat = min(len(kps0), len(kps1))
    k0, k01, _, _ = warp_homography(np.asarray(kps0)[:, :2], warp01["homography_matrix"], warp01["width"], warp01["height"], fused)
    k1, k10, _, _ = warp_homography(np.asarray(kps1)[:, :2], warp10["homography_matrix"], warp10["width"], warp10["height"], fused)
    if len(k0) == 0 or len(k1) == 0:
        return dict(num_feat=0, repeatability=0, mean_error=0, errors=None)
    s01 = float(warp01["resize"] if "resize" in warp01 else warp01["width"])
    s10 = float(warp10["resize"] if "resize" in warp10 else warp10["width"])
    M, N = len(k0), len(k1)
    cap = M * N
    pairs = np.empty((cap, 2), np.int32)
    dist = np.empty((cap,), np.float32)
    errors = np.empty((M,), np.float32)
    K = lib().kpbo_val_keypoints(_fp(k0), _fp(k01), M, _fp(k1), _fp(k10), N, s01, s10,
                                 pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fp(dist), cap, _fp(errors))
    dist = dist[:K]
    sel = dist[dist <= th]
    return dict(num_feat=num_feat, repeatability=np.float32(len(sel)) / np.float32(num_feat), mean_error=sel.mean() if len(sel) else np.float32("nan"),
                errors=errors, pairs=pairs[:K].astype(np.int64), dist=dist.copy())


def lk_track(img1_chw, img2_chw, pts1, pts2, unit, distance=3, win_size=3, levels=1, interation=40):
    """utils/matcher.py:7-142 OpticalFlow(params)(img1, img2, pts1, pts2) for one pair; `unit` [n,2] are the (cos, sin)
    of the reference's random angles.  Returns (pts [n,2] in pixels, error [n])."""
    a, b = _f32(img1_chw), _f32(img2_chw)
    C, H, W = a.shape
    p1, p2, u = _f32(pts1)[:, :2].copy(), _f32(pts2)[:, :2].copy(), _f32(unit)
    n = p1.shape[0]
    out = np.empty((n, 2), np.float32)
    err = np.empty((n,), np.float32)
    if n:
        lib().kpbo_lk_track(_fp(a), _fp(b), C, H, W, _fp(p1), _fp(p2), _fp(u), n, float(distance), int(win_size), int(levels),
                            int(interation), _fp(out), _fp(err))
    return out, err


def warp_se3(kps, depth0, depth1, kinv0, k1, pose01, bbox0, bbox1, fused=-1):
    """utils/projection.py:195-268.  kinv0 = torch.inverse(intrinsics0) as unproject (43) computes it.
    Returns (kpts0_valid, kpts01_valid, ids_valid, ids_out)."""
    p = _f32(kps)
    n = p.shape[0]
    d0, d1 = _f32(depth0), _f32(depth1)
    a = np.empty((max(n, 1), 2), np.float32); b = np.empty((max(n, 1), 2), np.float32)
    iv = np.empty((max(n, 1),), np.int32); io = np.empty((max(n, 1),), np.int32); cnt = np.zeros(2, np.int32)
    ipt = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    if n:
        lib().kpbo_warp_se3(_fp(p), n, p.shape[1], _fp(d0), d0.shape[0], d0.shape[1], _fp(d1), d1.shape[0], d1.shape[1],
                            _fp(_f32(kinv0).reshape(9)), _fp(_f32(k1).reshape(9)), _fp(_f32(pose01).reshape(16)), _fp(_f32(bbox0)), _fp(_f32(bbox1)),
                            int(fused), _fp(a), _fp(b), ipt(iv), ipt(io), ipt(cnt))
    return a[:cnt[0]].copy(), b[:cnt[0]].copy(), iv[:cnt[0]].astype(np.int64), io[:cnt[1]].astype(np.int64)


def epipolar_error(kps0, kps1, fmat, W, H, mode1=0):
    """tasks/FundamentalMatrix.py:137-144 in numpy fp32: |x1^T F x0| / max(|(F x0)_xy|, 1e-6) per match.
    kps0 [K, >=2] normalised rows; kps1 rows as the matcher branch leaves them (mode1 0: the (x, y, score) rows
    themselves, 120-122; 1: scaled to pixels with a 1 appended, 134-135; 2: pixels with a 1 appended, 117-119)."""
    k0 = _f32(kps0)
    k1 = _f32(kps1)
    F = _f32(fmat).reshape(3, 3)
    p0 = np.concatenate([k0[:, :2] * np.array([W - 1, H - 1], np.float32), np.ones((len(k0), 1), np.float32)], 1)
    if mode1 == 0:
        p1 = k1[:, :3]
    elif mode1 == 1:
        p1 = np.concatenate([k1[:, :2] * np.array([W - 1, H - 1], np.float32), np.ones((len(k1), 1), np.float32)], 1)
    else:
        p1 = np.concatenate([k1[:, :2], np.ones((len(k1), 1), np.float32)], 1)
    I = (F @ p0.T).astype(np.float32)                                   # 140
    err = np.abs(np.einsum("ij,ji->i", p1, I)).astype(np.float32)       # 141-142: the diagonal of kps1 @ I
    nrm = np.maximum(np.sqrt(I[0] * I[0] + I[1] * I[1]), np.float32(1e-6))   # 143
    return (err / nrm).astype(np.float32)


def fundamental_matrix(score0_hw, score1_hw, desc0_chw, desc1_chw, fmat, params):
    """tasks/FundamentalMatrix.py:89-161, brute-force branch (120-126), on numpy inputs.
    Returns (mean error, ratio under th, count under th, matched rows 0, matched rows 1)."""
    k0, _ = detection(score0_hw, params["extractor_params"])
    k1, _ = detection(score1_hw, params["extractor_params"])
    m0, m1 = brute_force_matcher(k0, k1, desc0_chw, desc1_chw, params["matcher_params"]["brute_force_params"])
    H, W = np.asarray(score0_hw).shape
    err = epipolar_error(m0, m1, fmat, W, H, 0)
    num = int((err < np.float32(params["FundamentalMatrix_params"]["th"])).sum())
    return float(err.mean(dtype=np.float32)), num / len(err), num, m0, m1
