"""CPU restatement of the robust-geometry stage -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference calls OpenCV (`cv2.findHomography(pts0, pts1, cv2.RANSAC)` tasks/MHA.py:45-47,
`cv2.findEssentialMat` + `cv2.recoverPose` tasks/AUC.py:50-64, `cv2.findFundamentalMat(.., FM_RANSAC)` utils/mvg.py:16); OpenCV is
a third-party dependency (requirements.txt:2 opencv-python~=4.9.0.80), absent from the reference tree and from this image, so
nothing here could be run against it.  What is restated is OpenCV 4.9's published algorithm for these calls with their default
parameters:

  the RANSAC driver (modules/calib3d/src/ptsetreg.cpp, RANSACPointSetRegistrator::run), r03 -- SAMPLER INCLUDED: cv::RNG
  (multiply-with-carry, state (uint64)-1 at every call), getSubset (each index redrawn until it differs from the ones already
  picked; the whole subset redrawn, up to 10 000 times, until checkSubset accepts it -- a rejected subset does not cost an
  iteration), models of an iteration scored in order, `goodCount > max(maxGoodCount, modelPoints - 1)` keeps a model and
  updates niters = RANSACUpdateNumIters(confidence, outlier ratio, modelPoints, niters), the loop ends at iter >= niters.
  The hypothesis stream is therefore OpenCV's by construction; hypotheses are EVALUATED a round of 256 at a time (one per
  thread of the device workgroup) and the keep / niters rule is then applied to them in iteration order, which gives what the
  sequential loop gives (hypotheses past the stopping iteration are discarded unseen).

  findHomography(RANSAC): ransacReprojThreshold 3, maxIters 2000, confidence 0.995; checkSubset = last point not collinear
  with two earlier ones in either image, and the four triangle orientations agree between the images; model from the sample;
  inliers = forward reprojection error^2 <= threshold^2; then a least-squares refit on the inliers (normalised DLT) and <= 10
  Levenberg-Marquardt steps on the reprojection error; H / H[2,2].

It is validated against ANALYTIC ground truth (tests/test_oracle_geometry.py), not against cv2.  The HIP kernels
(csrc/geometry.hip) follow the same steps with the same generator so that the two can be compared hypothesis for hypothesis.
Still restated rather than copied from OpenCV's arithmetic: the exact 4-point model is the projective-basis closed form (two
adjugates) rather than OpenCV's 9 x 9 eigen-decomposition, and the errors are evaluated in float64 where OpenCV's callbacks
use float32 -- same models and inlier sets up to rounding at the threshold.
"""
import numpy as np

ROUND = 256            # hypotheses per round: one per thread of the device workgroup
H_MAX_ITERS = 2000     # cv2.findHomography maxIters
H_CONFIDENCE = 0.995
H_THRESHOLD = 3.0
LM_ITERS = 10


CV_RNG_COEFF = 4164903690
GETSUBSET_ATTEMPTS = 10000     # RANSACPointSetRegistrator::run calls getSubset(.., rng, 10000)


class CvRNG:
    """cv::RNG (modules/core/include/opencv2/core/operations.hpp): multiply-with-carry on a 64-bit state."""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state if state else 0xFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * CV_RNG_COEFF + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def rng_state(seed):
    """seed 0: the state OpenCV's RANSAC starts EVERY call with, `RNG rng((uint64)-1)`.  Any other seed gives another non-zero
    64-bit state (tests that want different hypothesis streams; the device kernel maps seeds the same way)."""
    s = int(seed) & 0xFFFFFFFF
    return 0xFFFFFFFFFFFFFFFF if s == 0 else (((s ^ 0x9E3779B9) << 32) | s)


def have_collinear_last(p):
    """OpenCV's haveCollinearPoints(m, count): the LAST of the count points lies on a line through two earlier ones (or too
    close to one).  p [count, 2] float32-valued coordinates, arithmetic in float64 with FLT_EPSILON, as in OpenCV."""
    p = np.asarray(p, np.float64)
    i = len(p) - 1
    for j in range(i):
        dx1, dy1 = p[j, 0] - p[i, 0], p[j, 1] - p[i, 1]
        for k in range(j):
            dx2, dy2 = p[k, 0] - p[i, 0], p[k, 1] - p[i, 1]
            if abs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


def check_subset_homography(s, d):
    """HomographyEstimatorCallback::checkSubset for the 4-point sample: no collinear last point in either image, and the
    signed areas of the four triangles (0 1 2), (1 2 3), (0 2 3), (0 1 3) change sign in all of them or in none."""
    if have_collinear_last(s) or have_collinear_last(d):
        return False
    s, d = np.asarray(s, np.float64), np.asarray(d, np.float64)
    negative = 0
    for t in ((0, 1, 2), (1, 2, 3), (0, 2, 3), (0, 1, 3)):
        A = np.array([[s[t[0], 0], s[t[0], 1], 1.0], [s[t[1], 0], s[t[1], 1], 1.0], [s[t[2], 0], s[t[2], 1], 1.0]])
        B = np.array([[d[t[0], 0], d[t[0], 1], 1.0], [d[t[1], 0], d[t[1], 1], 1.0], [d[t[2], 0], d[t[2], 1], 1.0]])
        negative += _det3x3(A) * _det3x3(B) < 0
    return negative == 0 or negative == 4


def _det3x3(m):
    """cv::determinant(Matx33d): cofactor expansion along the first row."""
    return (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
            + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))


def check_subset_fundamental(s, d):
    """FMEstimatorCallback::checkSubset."""
    return not have_collinear_last(s) and not have_collinear_last(d)


def get_subset(rng, n, m, check, p0, p1, max_attempts=GETSUBSET_ATTEMPTS):
    """RANSACPointSetRegistrator::getSubset: m distinct indices in [0, n), redrawn as a whole until `check` accepts them."""
    for _ in range(max_attempts):
        idx = []
        for i in range(m):
            c = rng.uniform(0, n)
            while c in idx:
                c = rng.uniform(0, n)
            idx.append(c)
        if check is None or check(p0[idx], p1[idx]):
            return idx
    return None


def subsets_for_round(rng, n, m, check, p0, p1, count):
    """The next `count` subsets of the stream.  Returns (idx [count, m], ok [count]); ok turns False at a getSubset failure and
    stays False (the sequential loop ends there)."""
    idx = np.zeros((count, m), np.int64)
    ok = np.zeros(count, bool)
    for j in range(count):
        sub = get_subset(rng, n, m, check, p0, p1)
        if sub is None:
            break
        idx[j], ok[j] = sub, True
    return idx, ok


def ransac_scan(cnt, valid, sub_ok, base, state, n, m, conf):
    """The keep / niters rule of RANSACPointSetRegistrator::run applied, in iteration order, to one round of evaluated hypotheses.
    cnt / valid [R, NM]: inlier count and validity of every model of every hypothesis; state = dict(niters, max_good, best, iters,
    stop, failed).  `best` becomes (row of this round, model slot) when a model of this round is kept."""
    R, NM = cnt.shape
    best = None
    for j in range(R):
        it = base + j
        if it >= state["niters"]:
            state["stop"] = True
            break
        if not sub_ok[j]:                       # getSubset gave up: `if (iter == 0) return false; break;`
            state["failed"] = it == 0
            state["stop"] = True
            break
        for s_ in range(NM):
            if valid[j, s_] and cnt[j, s_] > max(state["max_good"], m - 1):
                best = (j, s_)
                state["max_good"] = int(cnt[j, s_])
                state["niters"] = update_iters(conf, (n - state["max_good"]) / n, m, state["niters"])
        state["iters"] = it + 1
    return best


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
    rng = CvRNG(rng_state(seed))
    st = dict(niters=max(max_iters, 1), max_good=0, iters=0, stop=False, failed=False)
    best_H, base = None, 0
    while base < st["niters"] and not st["stop"]:
        idx, sub_ok = subsets_for_round(rng, n, 4, check_subset_homography, src, dst, ROUND)
        H, good = homography_4pt(src[idx], dst[idx])
        cnt = (reproj_err2(H, src, dst) <= t2).sum(1)
        b = ransac_scan(cnt[:, None], (good & sub_ok)[:, None], sub_ok, base, st, n, 4, confidence)
        if b is not None:
            best_H = H[b[0]]
        base += ROUND
    if best_H is None:
        return None, mask, dict(iters=st["iters"], inliers=0)
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
    return H / H[2, 2], mask, dict(iters=st["iters"], inliers=int(st["max_good"]))


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


# ================================================================================================ essential matrix
# cv2.findEssentialMat(k0, k1, eye(3), threshold, prob, RANSAC) + cv2.recoverPose as tasks/AUC.py:50-64 calls them, restated
# from the published algorithm (Nister's five-point solver, as OpenCV implements it) -- PARITY UNPINNED, see the module
# docstring.  Defaults of the call: maxIters 1000, 5-point samples, Sampson error against threshold^2, no refinement of
# the winning model; recoverPose: the (R, t) of the four decompositions with the most points in front of both cameras.
#
# Five-point solver, per sample:
#   1. the 5 x 9 epipolar constraints; their 4-dimensional null space by Gauss-Jordan elimination with row pivoting
#      (basis rows X, Y, Z, W; E = x X + y Y + z Z + W);
#   2. det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0: ten cubics in (x, y, z), expanded by polynomial arithmetic into a
#      10 x 20 matrix over the monomials MONO (Nister's order);
#   3. Gauss-Jordan on the ten leading monomials; rows e - z f, g - z h, i - z j form a 3 x 3 matrix B(z) of polynomials
#      (degrees 3, 3, 4) acting on (x, y, 1); det B(z) is the tenth-degree polynomial;
#   4. its roots by Aberth-Ehrlich iteration from a fixed start; real ones polished by Newton on the real polynomial;
#   5. (x, y) from the null vector of B(z), E from the basis.
E_MAX_ITERS = 1000
E_MODEL_POINTS = 5
ABERTH_ITERS = 40

# exponents (i, j, k) of x^i y^j z^k, Nister's order: ten leading monomials, then the ten of the quotient ring
MONO = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0), (1, 1, 1), (1, 1, 0),
        (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0), (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]
_MIDX = {m: i for i, m in enumerate(MONO)}
_LIN = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]                      # x, y, z, 1: how an entry of E is stored
_DEG2 = [(2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
_D2IDX = {m: i for i, m in enumerate(_DEG2)}
# product tables: (deg-1 index, deg-1 index) -> deg-2 index; (deg-2 index, deg-1 index) -> MONO index
MUL11 = np.array([[_D2IDX[tuple(a + b for a, b in zip(p, q))] for q in _LIN] for p in _LIN])
MUL21 = np.array([[_MIDX[tuple(a + b for a, b in zip(p, q))] for q in _LIN] for p in _DEG2])
D2_TO_MONO = np.array([_MIDX[m] for m in _DEG2])


def _pmul11(a, b):
    """[T,4] x [T,4] linear polynomials -> [T,10] quadratic."""
    out = np.zeros(a.shape[:-1] + (10,))
    for i in range(4):
        for j in range(4):
            out[..., MUL11[i, j]] += a[..., i] * b[..., j]
    return out


def _pmul21(a, b):
    """[T,10] quadratic x [T,4] linear -> [T,20] cubic over MONO."""
    out = np.zeros(a.shape[:-1] + (20,))
    for i in range(10):
        for j in range(4):
            out[..., MUL21[i, j]] += a[..., i] * b[..., j]
    return out


def _gauss_jordan(M, ncol):
    """Reduces the first `ncol` columns of each [T, r, c] matrix to the identity (row pivoting).  Returns (M, ok)."""
    M = M.copy()
    T, r, _ = M.shape
    ok = np.ones(T, bool)
    ar = np.arange(T)
    for c in range(ncol):
        piv = c + np.argmax(np.abs(M[:, c:, c]), axis=1)
        tmp = M[ar, c].copy()
        M[ar, c] = M[ar, piv]
        M[ar, piv] = tmp
        d = M[:, c, c]
        good = np.abs(d) > 1e-300
        ok &= good
        M[:, c] = M[:, c] / np.where(good, d, 1.0)[:, None]
        for rr in range(r):
            if rr != c:
                M[:, rr] = M[:, rr] - M[:, rr, c][:, None] * M[:, c]
    return M, ok


def _polymul(a, b):
    """Coefficient arrays, ascending powers: [T, na] * [T, nb] -> [T, na + nb - 1]."""
    out = np.zeros((a.shape[0], a.shape[1] + b.shape[1] - 1))
    for i in range(a.shape[1]):
        out[:, i:i + b.shape[1]] += a[:, i:i + 1] * b
    return out


def _polyval(c, z):
    """c [T, n] ascending, z [T, ...] (real or complex) -> values."""
    v = np.zeros_like(z) + c[:, -1].reshape((-1,) + (1,) * (z.ndim - 1))
    for i in range(c.shape[1] - 2, -1, -1):
        v = v * z + c[:, i].reshape((-1,) + (1,) * (z.ndim - 1))
    return v


def aberth_roots(c, iters=ABERTH_ITERS):
    """All complex roots of the degree-10 polynomials c [T, 11] (ascending powers) by Aberth-Ehrlich iteration from a
    fixed circle of starting points scaled by the Cauchy bound.  Returns [T, 10] complex."""
    T, n1 = c.shape
    n = n1 - 1
    lead = c[:, -1]
    cn = c / np.where(lead == 0, 1.0, lead)[:, None]
    # start on a circle whose radius is the geometric mean of the root moduli, |c0 / cn|^(1/n)
    rad = np.abs(cn[:, 0]) ** (1.0 / n)
    rad = np.clip(np.where(np.isfinite(rad) & (rad > 0), rad, 1.0), 1e-3, 1e3)
    k = np.arange(n)
    z = rad[:, None] * np.exp(1j * (2 * np.pi * k / n + 0.4))[None]
    dc = cn[:, 1:] * np.arange(1, n1)[None]
    for _ in range(iters):
        p = _polyval(cn.astype(complex), z)
        dp = _polyval(dc.astype(complex), z)
        ratio = p / np.where(dp == 0, 1e-300, dp)
        diff = z[:, :, None] - z[:, None, :]
        diff[:, k, k] = 1.0
        inv = 1.0 / np.where(diff == 0, 1e-300, diff)
        inv[:, k, k] = 0.0
        s = inv.sum(2)
        z = z - ratio / (1.0 - ratio * s)
    return z


def essential_5pt(x1, x2):
    """Five-point solver for T samples: x1, x2 [T, 5, 2] normalised image points (x2^T E x1 = 0).
    Returns (E [T, 10, 3, 3], valid [T, 10]): up to ten candidates per sample, Frobenius norm 1."""
    T = x1.shape[0]
    u1, v1, u2, v2 = x1[..., 0], x1[..., 1], x2[..., 0], x2[..., 1]
    Q = np.stack([u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, np.ones_like(u1)], -1)        # [T, 5, 9]
    Qr, ok = _gauss_jordan(Q, 5)                       # [I | C]: null space rows n_j = (-C[:, j], e_j), j = 0..3
    basis = np.zeros((T, 4, 9))
    basis[:, :, :5] = -np.swapaxes(Qr[:, :, 5:], 1, 2)
    basis[:, np.arange(4), 5 + np.arange(4)] = 1.0
    Ep = np.swapaxes(basis, 1, 2).reshape(T, 3, 3, 4)  # entry (r, c) as a linear polynomial over (x, y, z, 1)
    # E E^T (quadratic entries), its trace, det E, and the nine cubics of 2 E E^T E - tr(E E^T) E
    EEt = np.zeros((T, 3, 3, 10))
    for i in range(3):
        for j in range(3):
            for k in range(3):
                EEt[:, i, j] += _pmul11(Ep[:, i, k], Ep[:, j, k])
    tr = EEt[:, 0, 0] + EEt[:, 1, 1] + EEt[:, 2, 2]
    M = np.zeros((T, 10, 20))
    r = 0
    for i in range(3):
        for j in range(3):
            acc = np.zeros((T, 20))
            for k in range(3):
                acc += 2.0 * _pmul21(EEt[:, i, k], Ep[:, k, j])
            acc -= _pmul21(tr, Ep[:, i, j])
            M[:, r] = acc
            r += 1
    m01 = _pmul11(Ep[:, 1, 1], Ep[:, 2, 2]) - _pmul11(Ep[:, 1, 2], Ep[:, 2, 1])
    m02 = _pmul11(Ep[:, 1, 2], Ep[:, 2, 0]) - _pmul11(Ep[:, 1, 0], Ep[:, 2, 2])
    m03 = _pmul11(Ep[:, 1, 0], Ep[:, 2, 1]) - _pmul11(Ep[:, 1, 1], Ep[:, 2, 0])
    M[:, 9] = _pmul21(m01, Ep[:, 0, 0]) + _pmul21(m02, Ep[:, 0, 1]) + _pmul21(m03, Ep[:, 0, 2])
    Mr, ok2 = _gauss_jordan(M, 10)
    ok &= ok2
    R = Mr[:, :, 10:]                                  # trailing coefficients over [xz2, xz, x, yz2, yz, y, z3, z2, z, 1]

    def row_polys(a, b):                               # <a> - z <b>: the (x, y, 1) parts as polynomials in z, ascending
        px = np.stack([a[:, 2], a[:, 1] - b[:, 2], a[:, 0] - b[:, 1], -b[:, 0]], 1)
        py = np.stack([a[:, 5], a[:, 4] - b[:, 5], a[:, 3] - b[:, 4], -b[:, 3]], 1)
        p1 = np.stack([a[:, 9], a[:, 8] - b[:, 9], a[:, 7] - b[:, 8], a[:, 6] - b[:, 7], -b[:, 6]], 1)
        return px, py, p1

    B = [row_polys(R[:, 4], R[:, 5]), row_polys(R[:, 6], R[:, 7]), row_polys(R[:, 8], R[:, 9])]
    det = (_polymul(_polymul(B[0][0], B[1][1]) - _polymul(B[0][1], B[1][0]), B[2][2])
           + _polymul(_polymul(B[0][1], B[1][2]), B[2][0]) - _polymul(_polymul(B[0][2], B[1][1]), B[2][0])
           + _polymul(_polymul(B[0][2], B[1][0]), B[2][1]) - _polymul(_polymul(B[0][0], B[1][2]), B[2][1]))      # [T, 11]
    ok &= np.isfinite(det).all(1) & (np.abs(det[:, -1]) > 1e-300)
    det = np.where(ok[:, None], det, np.r_[np.zeros(10), 1.0][None] + np.r_[-1.0, np.zeros(10)][None])   # z^10 - 1 for void samples
    roots = aberth_roots(det)
    zr = roots.real
    cn = det / det[:, -1:]
    dc = cn[:, 1:] * np.arange(1, 11)[None]
    for _ in range(3):                                 # Newton polish on the real axis
        p, dp = _polyval(cn, zr), _polyval(dc, zr)
        zr = zr - p / np.where(dp == 0, 1e-300, dp)
    real = (np.abs(roots.imag) < 1e-6 * (1.0 + np.abs(roots.real))) & np.isfinite(zr) & ok[:, None]
    real &= np.abs(_polyval(cn, zr)) < 1e-6 * (1.0 + np.abs(_polyval(np.abs(cn), np.abs(zr))))
    # (x, y, 1) = null vector of B(z): cross product of its first two rows
    def ev(pz):
        return _polyval(pz, zr)
    b00, b01, b02 = ev(B[0][0]), ev(B[0][1]), ev(B[0][2])
    b10, b11, b12 = ev(B[1][0]), ev(B[1][1]), ev(B[1][2])
    cx, cy, cw = b01 * b12 - b02 * b11, b02 * b10 - b00 * b12, b00 * b11 - b01 * b10
    real &= np.abs(cw) > 1e-300
    cw = np.where(np.abs(cw) > 1e-300, cw, 1.0)
    x, y = cx / cw, cy / cw
    coef = np.stack([x, y, zr, np.ones_like(x)], -1)                   # [T, 10, 4]
    E = np.einsum("tsk,tkn->tsn", coef, basis).reshape(T, 10, 3, 3)
    nrm = np.sqrt((E * E).sum((2, 3)))
    real &= np.isfinite(nrm) & (nrm > 0)
    E = E / np.where(real, nrm, 1.0)[:, :, None, None]
    return E, real


def sampson_err(E, x1, x2):
    """OpenCV's EMEstimatorCallback::computeError: E [..., 3, 3], x1 / x2 [N, 2] -> [..., N]."""
    X1 = np.concatenate([x1, np.ones((len(x1), 1))], 1)
    X2 = np.concatenate([x2, np.ones((len(x2), 1))], 1)
    Ex1 = np.einsum("...ij,nj->...ni", E, X1)
    Etx2 = np.einsum("...ji,nj->...ni", E, X2)
    x2tEx1 = (Ex1 * X2).sum(-1)
    den = Ex1[..., 0] ** 2 + Ex1[..., 1] ** 2 + Etx2[..., 0] ** 2 + Etx2[..., 1] ** 2
    return x2tEx1 * x2tEx1 / np.where(den == 0, 1e-300, den)


def find_essential_ransac(x1, x2, seed=0, threshold=1.0, prob=0.99999, max_iters=E_MAX_ITERS):
    """cv2.findEssentialMat(x1, x2, eye(3), threshold, prob, RANSAC) restated.  x1, x2 [N, 2] normalised coordinates.
    Returns (E [3,3] or None, mask [N] uint8, info).  EMEstimatorCallback has no checkSubset: every 5 distinct points are a sample."""
    x1, x2 = np.asarray(x1, np.float64), np.asarray(x2, np.float64)
    n = len(x1)
    mask = np.zeros(n, np.uint8)
    if n < 5:
        return None, mask, dict(iters=0, inliers=0)
    t2 = threshold * threshold
    rng = CvRNG(rng_state(seed))
    st = dict(niters=max(max_iters, 1), max_good=0, iters=0, stop=False, failed=False)
    best_E, base = None, 0
    while base < st["niters"] and not st["stop"]:
        if n == 5:                  # `count == modelPoints`: the one sample, no loop
            idx, sub_ok = np.arange(5)[None].repeat(ROUND, 0), np.zeros(ROUND, bool)
            sub_ok[0] = True
            st["niters"] = 1
        else:
            idx, sub_ok = subsets_for_round(rng, n, 5, None, x1, x2, ROUND)
        E, valid = essential_5pt(x1[idx], x2[idx])
        cnt = (sampson_err(E, x1, x2) <= t2).sum(-1)                    # [ROUND, 10]
        b = ransac_scan(cnt, valid & sub_ok[:, None], sub_ok, base, st, n, 5, prob)
        if b is not None:
            best_E = E[b[0], b[1]]
        base += ROUND
    if best_E is None:
        return None, mask, dict(iters=st["iters"], inliers=0)
    mask[sampson_err(best_E, x1, x2) <= t2] = 1
    return best_E, mask, dict(iters=st["iters"], inliers=int(st["max_good"]))


def decompose_essential(E):
    """cv2.decomposeEssentialMat: R1 = U W Vt, R2 = U W^T Vt, t = U[:, 2] with det(U), det(Vt) forced positive."""
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    Wm = np.array([[0.0, 1, 0], [-1, 0, 0], [0, 0, 1]])
    return U @ Wm @ Vt, U @ Wm.T @ Vt, U[:, 2]


def recover_pose(E, x1, x2, mask, dist=1e9):
    """cv2.recoverPose(E, x1, x2, eye(3), dist, mask): the decomposition with the most masked points in front of both
    cameras (depths from the two-ray least-squares triangulation z1 R x1 + t = z2 x2; OpenCV triangulates by DLT -- the
    depth SIGNS, which is all the count uses, agree except for points at infinity).  Returns (n, R, t, mask_new)."""
    R1, R2, t = decompose_essential(E)
    X1 = np.concatenate([x1, np.ones((len(x1), 1))], 1)
    X2 = np.concatenate([x2, np.ones((len(x2), 1))], 1)
    best = None
    for R, tt in ((R1, t), (R2, t), (R1, -t), (R2, -t)):                # OpenCV's order; ties keep the earlier one
        a = X1 @ R.T                                                    # R x1
        # [a, -X2] [z1, z2]^T = -t  (least squares, 2 x 2 normal equations per point)
        aa, ab, bb = (a * a).sum(1), -(a * X2).sum(1), (X2 * X2).sum(1)
        ra, rb = -(a @ tt), (X2 @ tt)
        det = aa * bb - ab * ab
        det = np.where(np.abs(det) > 1e-300, det, 1e-300)
        z1 = (ra * bb - ab * rb) / det
        z2 = (aa * rb - ab * ra) / det
        good = (z1 > 0) & (z1 < dist) & (z2 > 0) & (z2 < dist) & (mask > 0)
        if best is None or good.sum() > best[0]:
            best = (int(good.sum()), R, tt, good.astype(np.uint8))
    return best


def angle_error_mat(R1, R2):
    """tasks/AUC.py:66-69."""
    cos = (np.trace(np.dot(R1.T, R2)) - 1) / 2
    cos = np.clip(cos, -1.0, 1.0)
    return np.rad2deg(np.abs(np.arccos(cos)))


def angle_error_vec(v1, v2):
    """tasks/AUC.py:72-74."""
    n = np.linalg.norm(v1) * np.linalg.norm(v2)
    return np.rad2deg(np.arccos(np.clip(np.dot(v1, v2) / n, -1.0, 1.0)))


def compute_pose_error(T_0to1, R, t):
    """tasks/AUC.py:77-84."""
    R_gt, t_gt = T_0to1[:3, :3], T_0to1[:3, 3]
    error_t = angle_error_vec(t, t_gt)
    error_t = np.minimum(error_t, 180 - error_t)
    return error_t, angle_error_mat(R, R_gt)


def estimate_pose(kpts0, kpts1, K0, K1, thresh, conf=0.99999, seed=0):
    """tasks/AUC.py:40-64 with the two cv2 calls answered by the restatements above."""
    if len(kpts0) < 5:
        return None
    f_mean = np.mean([K0[0, 0], K1[1, 1], K0[0, 0], K1[1, 1]])
    norm_thresh = thresh / f_mean
    k0 = (kpts0 - K0[[0, 1], [2, 2]][None]) / K0[[0, 1], [0, 1]][None]
    k1 = (kpts1 - K1[[0, 1], [2, 2]][None]) / K1[[0, 1], [0, 1]][None]
    E, mask, _ = find_essential_ransac(k0, k1, seed=seed, threshold=norm_thresh, prob=conf)
    assert E is not None
    n, R, t, mask_new = recover_pose(E, k0, k1, mask)
    if n > 0:
        return R, t, mask_new > 0
    return None


# --------------------------------------------------------------------------------------------- fundamental matrix (7 points)
# cv2.findFundamentalMat(pts0, pts1, cv2.FM_RANSAC) as utils/mvg.py:16 calls it: ransacReprojThreshold 3, confidence
# 0.99, maxIters 1000; 7-point samples (degenerate when the last drawn point is collinear with two earlier ones in
# either image); up to three models per sample (the real roots of det(x F1 + F2) = 0 over the two-dimensional null space
# of the 7x9 epipolar system, built from the raw pixel coordinates in double); error of a correspondence = the larger of
# its two squared point-to-epipolar-line distances; inlier when error <= threshold^2; the best model's inliers are the
# mask.  No refit follows (OpenCV returns the best minimal model).  PARITY UNPINNED, as everything in this file.
F_MAX_ITERS = 1000
F_CONFIDENCE = 0.99
F_THRESHOLD = 3.0
FLT_EPSILON = 1.1920928955078125e-07


def have_collinear(p):
    """OpenCV's haveCollinearPoints on samples p [T, m, 2]: is the LAST point on (or too close to) a line through two
    earlier ones."""
    T, m, _ = p.shape
    bad = np.zeros(T, bool)
    i = m - 1
    for j in range(i):
        dx1, dy1 = p[:, j, 0] - p[:, i, 0], p[:, j, 1] - p[:, i, 1]
        for k in range(j):
            dx2, dy2 = p[:, k, 0] - p[:, i, 0], p[:, k, 1] - p[:, i, 1]
            bad |= np.abs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (np.abs(dx1) + np.abs(dy1) + np.abs(dx2) + np.abs(dy2))
    return bad


def _det3(m):
    return (m[..., 0, 0] * (m[..., 1, 1] * m[..., 2, 2] - m[..., 1, 2] * m[..., 2, 1])
            - m[..., 0, 1] * (m[..., 1, 0] * m[..., 2, 2] - m[..., 1, 2] * m[..., 2, 0])
            + m[..., 0, 2] * (m[..., 1, 0] * m[..., 2, 1] - m[..., 1, 1] * m[..., 2, 0]))


def solve_cubic(c3, c2, c1, c0):
    """Real roots of c3 x^3 + c2 x^2 + c1 x + c0 for T polynomials (the trigonometric / Cardano form OpenCV's solveCubic
    uses).  Returns (roots [T, 3], valid [T, 3]); a vanishing leading coefficient voids the sample."""
    T = len(c3)
    roots = np.zeros((T, 3))
    valid = np.zeros((T, 3), bool)
    lead = (c3 != 0) & np.isfinite(c3) & np.isfinite(c2) & np.isfinite(c1) & np.isfinite(c0)
    inv = 1.0 / np.where(lead, c3, 1.0)
    a1, a2, a3 = c2 * inv, c1 * inv, c0 * inv
    Q = (a1 * a1 - 3.0 * a2) * (1.0 / 9.0)
    R = (2.0 * a1 * a1 * a1 - 9.0 * a1 * a2 + 27.0 * a3) * (1.0 / 54.0)
    Q3 = Q * Q * Q
    d = Q3 - R * R
    three = lead & (d > 0)
    with np.errstate(all="ignore"):
        theta = np.arccos(np.clip(R / np.sqrt(np.where(three, Q3, 1.0)), -1.0, 1.0))
        t0 = -2.0 * np.sqrt(np.where(three, Q, 0.0))
        t2 = a1 * (1.0 / 3.0)
        for k in range(3):
            roots[:, k] = np.where(three, t0 * np.cos(theta * (1.0 / 3.0) + k * (2.0 * np.pi / 3.0)) - t2, roots[:, k])
            valid[:, k] = three
        one = lead & ~three
        e = np.cbrt(np.sqrt(np.where(one, -d, 0.0)) + np.abs(R))
        e = np.where(R > 0, -e, e)
        x = np.where(e != 0, e + Q / np.where(e != 0, e, 1.0), 0.0) - t2
    roots[:, 0] = np.where(one, x, roots[:, 0])
    valid[:, 0] |= one
    valid &= np.isfinite(roots)
    return roots, valid


def fundamental_7pt(x1, x2):
    """x1, x2 [T, 7, 2] pixel coordinates -> (F [T, 3, 3, 3] up to three models each with F[2,2] = 1, valid [T, 3])."""
    T = x1.shape[0]
    u1, v1, u2, v2 = x1[..., 0], x1[..., 1], x2[..., 0], x2[..., 1]
    A = np.stack([u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, np.ones_like(u1)], -1)       # [T, 7, 9]
    G, ok = _gauss_jordan(A, 7)
    # null space: b_j = (-G[:, :, 7 + j], e_j), j = 0, 1
    B1 = np.concatenate([-G[:, :, 7], np.ones((T, 1)), np.zeros((T, 1))], 1).reshape(T, 3, 3)
    B2 = np.concatenate([-G[:, :, 8], np.zeros((T, 1)), np.ones((T, 1))], 1).reshape(T, 3, 3)
    # det(x B1 + B2) = c3 x^3 + c2 x^2 + c1 x + c0, expanded by rows
    c3, c0 = _det3(B1), _det3(B2)
    c2 = np.zeros(T)
    c1 = np.zeros(T)
    for r in range(3):
        m = B1.copy(); m[:, r] = B2[:, r]; c2 += _det3(m)
        m = B2.copy(); m[:, r] = B1[:, r]; c1 += _det3(m)
    roots, valid = solve_cubic(c3, c2, c1, c0)
    valid &= ok[:, None]
    F = roots[:, :, None, None] * B1[:, None] + B2[:, None]                 # [T, 3, 3, 3]
    s = F[..., 2, 2]
    big = np.abs(s) > 2.220446049250313e-16
    F = F / np.where(big, s, 1.0)[..., None, None]
    valid &= np.isfinite(F).all((-1, -2))
    return F, valid


def fm_error(F, x1, x2):
    """OpenCV's FMEstimatorCallback::computeError: F [..., 3, 3], x1 / x2 [N, 2] -> [..., N], float32 like OpenCV's err."""
    X1 = np.concatenate([x1, np.ones((len(x1), 1))], 1)
    X2 = np.concatenate([x2, np.ones((len(x2), 1))], 1)
    l2 = np.einsum("...ij,nj->...ni", F, X1)           # epipolar lines in image 2
    l1 = np.einsum("...ji,nj->...ni", F, X2)           # in image 1
    with np.errstate(all="ignore"):
        d2 = (l2 * X2).sum(-1); s2 = 1.0 / (l2[..., 0] ** 2 + l2[..., 1] ** 2)
        d1 = (l1 * X1).sum(-1); s1 = 1.0 / (l1[..., 0] ** 2 + l1[..., 1] ** 2)
        return np.maximum(d1 * d1 * s1, d2 * d2 * s2).astype(np.float32)


LMEDS_OUTLIER_RATIO = 0.45     # LMeDSPointSetRegistrator::run
LMEDS_MIN_POINTS = 15          # cv::findFundamentalMat: `(method & ~3) == FM_RANSAC && npoints >= 15` runs RANSAC, otherwise LMedS
LMEDS_ATTEMPTS = 1000          # its getSubset call leaves maxAttempts at the default


def find_fundamental_lmeds(x1, x2, seed=0, confidence=F_CONFIDENCE, max_iters=F_MAX_ITERS):
    """What cv2.findFundamentalMat(.., FM_RANSAC) does with 8 <= n < 15 points: OpenCV 4.9's fundam.cpp hands fewer than
    15 correspondences to the LEAST-MEDIAN registrator whatever method was asked for.  LMeDSPointSetRegistrator::run restated:
    niters = max(RANSACUpdateNumIters(confidence, 0.45, 7, maxIters), 3) (= 300 at the defaults), the same cv::RNG / getSubset /
    checkSubset stream, every model of every sample scored by the MEDIAN (element count / 2 of the sorted float32 errors), the
    strictly smallest median kept; then sigma = 2.5 * 1.4826 * (1 + 5 / (count - 7)) * sqrt(median), at least 0.001, inliers =
    error <= sigma^2, and the model stands if at least 7 points are inliers.  PARITY UNPINNED like everything in this module.
    Note for 8 <= n <= 13: element n / 2 of the sorted errors is then one of the SAMPLE's own seven residuals (zero up to rounding), so
    the "strictly smallest median" is decided by rounding noise -- in OpenCV's float32 errors too: no implementation can promise
    OpenCV's winner there, only a model that interpolates seven points and the sigma rule (tests/test_gpu_geometry.py)."""
    x1, x2 = np.asarray(x1, np.float64), np.asarray(x2, np.float64)
    n = len(x1)
    mask = np.zeros(n, np.uint8)
    niters = max(update_iters(confidence, LMEDS_OUTLIER_RATIO, 7, max_iters), 3)
    rng = CvRNG(rng_state(seed))
    best_F, min_median, base, iters, stop = None, np.inf, 0, 0, False
    while base < niters and not stop:
        idx = np.zeros((ROUND, 7), np.int64)
        ok = np.zeros(ROUND, bool)
        for j in range(ROUND):
            sub = get_subset(rng, n, 7, check_subset_fundamental, x1, x2, LMEDS_ATTEMPTS)
            if sub is None:
                break
            idx[j], ok[j] = sub, True
        F, valid = fundamental_7pt(x1[idx], x2[idx])
        with np.errstate(invalid="ignore"):
            err = fm_error(F, x1, x2)                                       # [ROUND, 3, n] float32
        med = np.sort(err, axis=-1)[..., n // 2].astype(np.float64)         # std::nth_element(.., count / 2, ..)
        for j in range(ROUND):
            it = base + j
            if it >= niters:
                stop = True
                break
            if not ok[j]:                                                   # `if (iter == 0) return false; break;`
                stop = True
                break
            for s_ in range(3):
                if valid[j, s_] and med[j, s_] < min_median:
                    min_median, best_F = med[j, s_], F[j, s_]
            iters = it + 1
        base += ROUND
    if best_F is None:
        return None, mask, dict(iters=iters, inliers=0)
    sigma = max(2.5 * 1.4826 * (1 + 5.0 / (n - 7)) * np.sqrt(min_median), 0.001)
    mask[fm_error(best_F, x1, x2) <= np.float32(sigma * sigma)] = 1
    good = int(mask.sum())
    if good < 7:                                                            # `result = count >= modelPoints`
        return None, mask, dict(iters=iters, inliers=good)
    return best_F, mask, dict(iters=iters, inliers=good)


def find_fundamental_ransac(x1, x2, seed=0, threshold=F_THRESHOLD, confidence=F_CONFIDENCE, max_iters=F_MAX_ITERS):
    """cv2.findFundamentalMat(x1, x2, cv2.FM_RANSAC) restated for n >= 8 (utils/mvg.py:13 never calls it with fewer): the RANSAC
    registrator from 15 points on, the least-median one below (find_fundamental_lmeds).
    x1, x2 [N, 2] pixel coordinates (float32 values).  Returns (F [3,3] or None, mask [N] uint8, info)."""
    x1, x2 = np.asarray(x1, np.float64), np.asarray(x2, np.float64)
    n = len(x1)
    mask = np.zeros(n, np.uint8)
    if n < 8:
        return None, mask, dict(iters=0, inliers=0)
    if n < LMEDS_MIN_POINTS:
        return find_fundamental_lmeds(x1, x2, seed, confidence, max_iters)
    t2 = np.float32(threshold * threshold)
    rng = CvRNG(rng_state(seed))
    st = dict(niters=max(max_iters, 1), max_good=0, iters=0, stop=False, failed=False)
    best_F, base = None, 0
    while base < st["niters"] and not st["stop"]:
        idx, sub_ok = subsets_for_round(rng, n, 7, check_subset_fundamental, x1, x2, ROUND)
        F, valid = fundamental_7pt(x1[idx], x2[idx])
        with np.errstate(invalid="ignore"):
            cnt = (fm_error(F, x1, x2) <= t2).sum(-1)                   # [ROUND, 3]
        b = ransac_scan(cnt, valid & sub_ok[:, None], sub_ok, base, st, n, 7, confidence)
        if b is not None:
            best_F = F[b[0], b[1]]
        base += ROUND
    if best_F is None:
        return None, mask, dict(iters=st["iters"], inliers=0)
    mask[fm_error(best_F, x1, x2) <= t2] = 1
    return best_F, mask, dict(iters=st["iters"], inliers=int(st["max_good"]))


def fundamental_estimate(pts0, pts1, seed=0):
    """utils/mvg.py:4-19 with the cv2 call answered by the restatement above."""
    if len(pts0) < 8:
        return None, pts0, pts1
    F, mask, _ = find_fundamental_ransac(pts0, pts1, seed=seed)
    if F is None:                   # cv2 hands back (None, None); the reference then fails on mask.ravel()
        raise AttributeError("'NoneType' object has no attribute 'ravel'")
    return F, pts0[mask == 1], pts1[mask == 1]
