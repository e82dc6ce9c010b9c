"""CPU restatement of the robust-geometry stage -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference calls OpenCV (`cv2.findHomography(pts0, pts1, cv2.RANSAC)` tasks/MHA.py:45-47,
`cv2.findEssentialMat` + `cv2.recoverPose` tasks/AUC.py:50-64); OpenCV is a third-party dependency (requirements.txt,
unpinned), absent from the reference tree and from this image, and its RANSAC draws from its own RNG, so not even a
present cv2 could be matched bit for bit.  What is restated here is OpenCV's published algorithm for these calls with
their default parameters:

  findHomography(RANSAC): ransacReprojThreshold 3, maxIters 2000, confidence 0.995; 4-point samples (degenerate =
  three collinear points, or a sample whose orientation flips); model from the sample; inliers = forward reprojection
  error^2 <= threshold^2; keep the model with strictly more inliers; iterations adapted with
  log(1-confidence)/log(1-w^4); then a least-squares refit on the inliers (normalised DLT) and <= 10
  Levenberg-Marquardt steps on the reprojection error; H / H[2,2].

and it is validated against ANALYTIC ground truth (tests/test_oracle_geometry.py), not against cv2.  The HIP kernels
(csrc/geometry.hip) follow the same steps with the same counter-based sample generator (`sample_index`) so that the two
can be compared hypothesis for hypothesis; the sampler is this build's own (OpenCV's RNG is not reproduced).
The exact 4-point model is the projective-basis closed form (two adjugates) rather than an 8x8 solve: same model up to
rounding.
"""
import numpy as np

ROUND = 256            # hypotheses per round: one per thread of the device workgroup
H_MAX_ITERS = 2000     # cv2.findHomography maxIters
H_CONFIDENCE = 0.995
H_THRESHOLD = 3.0
LM_ITERS = 10


def lowbias32(h):
    h = np.asarray(h, np.uint64) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x7FEB352D) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x846CA68B) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def sample_index(seed, it, draw, n):
    """Index in [0, n) of draw number `draw` of hypothesis `it` (vectorised over `it`)."""
    it = np.asarray(it, np.uint64)
    h = (np.uint64(seed) ^ ((it * 0x9E3779B1) & 0xFFFFFFFF) ^ ((np.uint64(draw) * 0x85EBCA77) & 0xFFFFFFFF)) & 0xFFFFFFFF
    return ((lowbias32(h) * np.uint64(n)) >> 32).astype(np.int64)


def draw_samples(seed, its, n, m):
    """m distinct indices per hypothesis: successive draws, a draw equal to an earlier pick of the same hypothesis is
    skipped; a hypothesis that has not found m distinct indices after 4 m draws is void (ok = False)."""
    its = np.asarray(its, np.int64)
    idx = np.zeros((len(its), m), np.int64)
    have = np.zeros(len(its), np.int64)
    for d in range(4 * m):
        c = sample_index(seed, its, d, n)
        need = have < m
        dup = np.zeros(len(its), bool)
        for j in range(m):
            dup |= (j < have) & (idx[:, j] == c)
        take = need & ~dup
        idx[take, have[take]] = c[take]
        have[take] += 1
    return idx, have == m


def _adj(a):
    """adjugate of [..., 3, 3]"""
    c = np.empty_like(a)
    c[..., 0, 0] = a[..., 1, 1] * a[..., 2, 2] - a[..., 1, 2] * a[..., 2, 1]
    c[..., 0, 1] = a[..., 0, 2] * a[..., 2, 1] - a[..., 0, 1] * a[..., 2, 2]
    c[..., 0, 2] = a[..., 0, 1] * a[..., 1, 2] - a[..., 0, 2] * a[..., 1, 1]
    c[..., 1, 0] = a[..., 1, 2] * a[..., 2, 0] - a[..., 1, 0] * a[..., 2, 2]
    c[..., 1, 1] = a[..., 0, 0] * a[..., 2, 2] - a[..., 0, 2] * a[..., 2, 0]
    c[..., 1, 2] = a[..., 0, 2] * a[..., 1, 0] - a[..., 0, 0] * a[..., 1, 2]
    c[..., 2, 0] = a[..., 1, 0] * a[..., 2, 1] - a[..., 1, 1] * a[..., 2, 0]
    c[..., 2, 1] = a[..., 0, 1] * a[..., 2, 0] - a[..., 0, 0] * a[..., 2, 1]
    c[..., 2, 2] = a[..., 0, 0] * a[..., 1, 1] - a[..., 0, 1] * a[..., 1, 0]
    return c


def homography_4pt(src, dst):
    """Exact homographies of T samples: src, dst [T, 4, 2] -> (H [T, 3, 3] with H[2,2] = 1, ok [T]).
    Projective-basis construction: S = [p1 p2 p3] diag(lambda), lambda = adj([p1 p2 p3]) p4, maps the canonical basis to
    the source points; D likewise for the destination; H = D S^-1, with S^-1 taken as diag(l2 l3, l1 l3, l1 l2) adj(A)
    (a common factor drops out).  A sample is degenerate when a lambda vanishes (three collinear points) or when the
    sample's orientation differs between the images (OpenCV's checkSubset)."""
    one = np.ones(src.shape[:-1] + (1,))
    P = np.concatenate([src, one], -1)            # [T, 4, 3]
    Q = np.concatenate([dst, one], -1)
    A = np.swapaxes(P[:, :3], 1, 2)               # columns p1 p2 p3
    B = np.swapaxes(Q[:, :3], 1, 2)
    adjA, adjB = _adj(A), _adj(B)
    lam = np.einsum("tij,tj->ti", adjA, P[:, 3])
    mu = np.einsum("tij,tj->ti", adjB, Q[:, 3])
    detA = np.einsum("ti,ti->t", A[:, 0, :], adjA[:, :, 0])
    detB = np.einsum("ti,ti->t", B[:, 0, :], adjB[:, :, 0])
    # triangle orientations of the four triples (p1 p2 p3), (p2 p3 p4), (p1 p3 p4), (p1 p2 p4) must agree in sign between
    # the images; lam / mu components are exactly those signed areas (up to the sign of det)
    sA = np.sign(np.concatenate([detA[:, None], lam], 1))
    sB = np.sign(np.concatenate([detB[:, None], mu], 1))
    scaleA = np.abs(A[:, :2]).max((1, 2)) ** 2 + 1e-300
    scaleB = np.abs(B[:, :2]).max((1, 2)) ** 2 + 1e-300
    tiny = 1e-9
    ok = (np.abs(detA) > tiny * scaleA) & (np.abs(detB) > tiny * scaleB) & (np.abs(lam) > tiny * scaleA[:, None]).all(1) & \
         (np.abs(mu) > tiny * scaleB[:, None]).all(1) & (sA * sA[:, :1] == sB * sB[:, :1]).all(1)
    w = np.stack([lam[:, 1] * lam[:, 2], lam[:, 0] * lam[:, 2], lam[:, 0] * lam[:, 1]], 1)
    H = np.einsum("tij,tj,tjk->tik", B * mu[:, None, :], w, adjA)
    h22 = H[:, 2, 2]
    ok &= np.abs(h22) > 1e-12 * np.abs(H).max((1, 2))
    H = H / np.where(ok, h22, 1.0)[:, None, None]
    return H, ok


def reproj_err2(H, src, dst):
    """Forward reprojection error^2 of every point under every model: H [T,3,3], src/dst [N,2] -> [T,N]."""
    x, y = src[:, 0][None], src[:, 1][None]
    w = H[:, 2, 0, None] * x + H[:, 2, 1, None] * y + H[:, 2, 2, None]
    w = np.where(np.abs(w) > 2.220446049250313e-16, 1.0 / np.where(w == 0, 1, w), 0.0)      # OpenCV: ww = fabs(w) > eps ? 1/w : 0
    dx = (H[:, 0, 0, None] * x + H[:, 0, 1, None] * y + H[:, 0, 2, None]) * w - dst[:, 0][None]
    dy = (H[:, 1, 0, None] * x + H[:, 1, 1, None] * y + H[:, 1, 2, None]) * w - dst[:, 1][None]
    return dx * dx + dy * dy


def update_iters(conf, outlier_ratio, m, max_iters):
    """OpenCV RANSACUpdateNumIters."""
    p = min(max(conf, 0.0), 1.0)
    ep = min(max(outlier_ratio, 0.0), 1.0)
    num = max(1.0 - p, 2.2250738585072014e-308)
    denom = 1.0 - (1.0 - ep) ** m
    if denom < 2.2250738585072014e-308:
        return 0
    num, denom = np.log(num), np.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(round(num / denom))


def dlt_inhomogeneous(src, dst):
    """Least-squares homography of the inliers: normalised coordinates (centroid, mean absolute deviation per axis, as
    OpenCV's runKernel), h33 = 1 in the normalised frame, 8x8 normal equations."""
    def norm(p):
        c = p.mean(0)
        s = np.abs(p - c).mean(0)
        s = np.where(s > 2.220446049250313e-16, 1.0 / s, 1.0)
        return (p - c) * s, c, s
    a, ca, sa = norm(src)
    b, cb, sb = norm(dst)
    n = len(a)
    M = np.zeros((2 * n, 8))
    r = np.zeros(2 * n)
    M[0::2, 0:2], M[0::2, 2] = a, 1.0
    M[0::2, 6:8] = -b[:, :1] * a
    r[0::2] = b[:, 0]
    M[1::2, 3:5], M[1::2, 5] = a, 1.0
    M[1::2, 6:8] = -b[:, 1:] * a
    r[1::2] = b[:, 1]
    try:
        h = np.linalg.solve(M.T @ M, M.T @ r)
    except np.linalg.LinAlgError:
        return None
    Hn = np.append(h, 1.0).reshape(3, 3)
    Ta = np.array([[sa[0], 0, -ca[0] * sa[0]], [0, sa[1], -ca[1] * sa[1]], [0, 0, 1]])
    Tb_inv = np.array([[1 / sb[0], 0, cb[0]], [0, 1 / sb[1], cb[1]], [0, 0, 1]])
    H = Tb_inv @ Hn @ Ta
    if abs(H[2, 2]) < 1e-300:
        return None
    return H / H[2, 2]


def lm_refine(H, src, dst, iters=LM_ITERS):
    """Levenberg-Marquardt on the forward reprojection error over h11..h32 (h33 = 1), as cv2's HomographyRefineCallback
    parametrises it.  lambda starts at 1e-3, /10 on an accepted step, x10 on a rejected one."""
    h = (H / H[2, 2]).ravel()[:8].copy()

    def resid(h):
        w = h[6] * src[:, 0] + h[7] * src[:, 1] + 1.0
        wi = np.where(np.abs(w) > 2.220446049250313e-16, 1.0 / np.where(w == 0, 1, w), 0.0)
        u = (h[0] * src[:, 0] + h[1] * src[:, 1] + h[2]) * wi
        v = (h[3] * src[:, 0] + h[4] * src[:, 1] + h[5]) * wi
        return u, v, wi

    u, v, wi = resid(h)
    err = ((u - dst[:, 0]) ** 2 + (v - dst[:, 1]) ** 2).sum()
    lam = 1e-3
    for _ in range(iters):
        x, y = src[:, 0], src[:, 1]
        J = np.zeros((2 * len(src), 8))
        J[0::2, 0], J[0::2, 1], J[0::2, 2] = x * wi, y * wi, wi
        J[0::2, 6], J[0::2, 7] = -x * wi * u, -y * wi * u
        J[1::2, 3], J[1::2, 4], J[1::2, 5] = x * wi, y * wi, wi
        J[1::2, 6], J[1::2, 7] = -x * wi * v, -y * wi * v
        r = np.empty(2 * len(src))
        r[0::2], r[1::2] = u - dst[:, 0], v - dst[:, 1]
        JtJ, Jtr = J.T @ J, J.T @ r
        improved = False
        for _try in range(6):
            try:
                step = np.linalg.solve(JtJ + lam * np.diag(np.diag(JtJ)), -Jtr)
            except np.linalg.LinAlgError:
                lam *= 10
                continue
            hn = h + step
            un, vn, win = resid(hn)
            en = ((un - dst[:, 0]) ** 2 + (vn - dst[:, 1]) ** 2).sum()
            if en < err:
                h, u, v, wi, err, lam, improved = hn, un, vn, win, en, lam / 10, True
                break
            lam *= 10
        if not improved:
            break
    return np.append(h, 1.0).reshape(3, 3)


def find_homography_ransac(src, dst, seed=0, threshold=H_THRESHOLD, max_iters=H_MAX_ITERS, confidence=H_CONFIDENCE, refine=True):
    """cv2.findHomography(src, dst, cv2.RANSAC) restated (see the module docstring).  src, dst [N, 2] pixel coordinates.
    Returns (H [3,3] float64 or None, mask [N] uint8, info dict)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    n = len(src)
    mask = np.zeros(n, np.uint8)
    if n < 4:
        return None, mask, dict(iters=0, inliers=0)
    if n == 4:                      # OpenCV: exactly the minimal set -> the model itself, every point an inlier
        H, ok = homography_4pt(src[None], dst[None])
        if not ok[0]:
            return None, mask, dict(iters=0, inliers=0)
        mask[:] = 1
        return H[0], mask, dict(iters=0, inliers=4)
    t2 = threshold * threshold
    best_cnt, best_H, niters, done = 0, None, max_iters, 0
    while done < niters:
        its = np.arange(done, done + ROUND)
        idx, ok = draw_samples(seed, its, n, 4)
        H, good = homography_4pt(src[idx], dst[idx])
        ok &= good & (its < max_iters)
        cnt = (reproj_err2(H, src, dst) <= t2).sum(1)
        cnt = np.where(ok, cnt, 0)
        j = int(np.argmax(cnt))                           # first maximum = lowest iteration number
        if cnt[j] > max(best_cnt, 3):
            best_cnt, best_H = int(cnt[j]), H[j]
        done += ROUND
        niters = update_iters(confidence, (n - best_cnt) / n, 4, max_iters) if best_cnt else max_iters
    if best_H is None:
        return None, mask, dict(iters=done, inliers=0)
    inl = reproj_err2(best_H[None], src, dst)[0] <= t2
    mask[inl] = 1
    H = best_H
    if refine:
        H0 = dlt_inhomogeneous(src[inl], dst[inl])
        if H0 is not None and np.isfinite(H0).all():
            e0 = reproj_err2(H0[None], src[inl], dst[inl]).sum()
            eb = reproj_err2(best_H[None], src[inl], dst[inl]).sum()
            H = H0 if e0 < eb else best_H
        H = lm_refine(H, src[inl], dst[inl])
    return H / H[2, 2], mask, dict(iters=done, inliers=int(best_cnt))


# --------------------------------------------------------------------------------------------- task halves (pinnable)
def mha_corner_error(H, real_H, h, w, resize_h, resize_w):
    """tasks/MHA.py:50-66: mean distance between the image corners warped by the estimated and by the true homography.
    (The corner rows are written (h-1, 0), (0, w-1) in the reference: kept.)"""
    corners = np.array([[0, 0, 1], [h - 1, 0, 1], [0, w - 1, 1], [h - 1, w - 1, 1]])
    real = np.dot(corners, np.transpose(real_H))
    real = real[:, :2] / real[:, 2:]
    est = np.dot(corners, np.transpose(H))
    est = est[:, :2] / est[:, 2:]
    real = real * np.array([resize_h / h, resize_w / w])
    est = est * np.array([resize_h / h, resize_w / w])
    return np.mean(np.linalg.norm(real - est, axis=1))
