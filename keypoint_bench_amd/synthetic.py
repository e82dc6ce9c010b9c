"""Synthetic inputs of BASELINE.md section 3 (no datasets travel to the GPU box).

Everything is built from elementwise IEEE operations on numpy ``default_rng`` streams so that the
same seed gives the same bytes here and on the GPU box (fixtures store a checksum to prove it).
"""
import hashlib

import numpy as np


def _box_blur(a, k):
    """k x k box filter, edge-replicated, accumulated in float64 in a fixed order."""
    r = k // 2
    p = np.pad(a.astype(np.float64), ((r, r), (r, r)), mode="edge")
    H, W = a.shape
    acc = np.zeros((H, W), np.float64)
    for dy in range(k):
        for dx in range(k):
            acc = acc + p[dy:dy + H, dx:dx + W]
    return acc / float(k * k)


def image_pair(i, H=480, W=640):
    """Pair i: fp32 RGB [3,H,W] in [0,1] x2.  Canvas (H+20)x(W+20) uniform noise, 5x5 box blur per
    channel, min-max normalised; view0 = canvas[10:10+H, 10:10+W]; view1 = canvas[12:, 13:] + N(0, 0.02^2)."""
    rng = np.random.default_rng(1234 + i)
    canvas = rng.random((3, H + 20, W + 20), dtype=np.float32)
    canvas = np.stack([_box_blur(c, 5) for c in canvas])
    lo, hi = canvas.min(), canvas.max()
    canvas = ((canvas - lo) / (hi - lo)).astype(np.float32)
    v0 = canvas[:, 10:10 + H, 10:10 + W].copy()
    noise = rng.normal(0.0, 0.02, size=(3, H, W))
    v1 = np.clip(canvas[:, 12:12 + H, 13:13 + W].astype(np.float64) + noise, 0.0, 1.0).astype(np.float32)
    return v0, v1


def score_uniform(seed, H, W):
    """Detector microbench map (i): uniform [0,1) float32."""
    return np.random.default_rng(seed).random((H, W), dtype=np.float32)


def score_smooth(seed, H, W):
    """Detector microbench map (ii): sigmoid(36 * twice-9x9-box-blurred N(0,1))."""
    z = np.random.default_rng(seed).normal(size=(H, W))
    z = _box_blur(_box_blur(z, 9), 9) * 36.0
    return (1.0 / (1.0 + np.exp(-z))).astype(np.float32)


def checksum(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# ---------------------------------------------------------------------------------------------- viewpoint pairs
def _mm3(a, b):
    """3x3 product in a fixed order of Python-float operations (no BLAS: the same bytes on every box)."""
    return [[a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j] for j in range(3)] for i in range(3)]


def _inv3(m):
    """Adjugate inverse, again in plain float operations."""
    (a, b, c), (d, e, f), (g, h, i) = m
    A, B, C = e * i - f * h, c * h - b * i, b * f - c * e
    D, E, F = f * g - d * i, a * i - c * g, c * d - a * f
    G, Hh, I = d * h - e * g, b * g - a * h, a * e - b * d
    det = a * A + b * D + c * G
    return [[A / det, B / det, C / det], [D / det, E / det, F / det], [G / det, Hh / det, I / det]]


def viewpoint_homography(rng, H, W, strength=1.0):
    """A seeded HPatches-like viewpoint change acting on view-0 PIXEL coordinates (column, row): rotation up to +-12 degrees,
    scale 0.88..1.12, shear, perspective and a shift, all about the image centre, scaled by `strength`."""
    import math
    u = [float(v) for v in rng.uniform(-1.0, 1.0, size=8)]
    ang = math.radians(12.0) * strength * u[0]
    sc = 1.0 + 0.12 * strength * u[1]
    sh = 0.08 * strength * u[2]
    asp = 1.0 + 0.06 * strength * u[3]
    px, py = 1.6e-4 * strength * u[4], 1.6e-4 * strength * u[5]
    tx, ty = 24.0 * strength * u[6], 18.0 * strength * u[7]
    ca, sa = math.cos(ang), math.sin(ang)
    M = [[sc * asp * ca, sc * (sh * ca - sa), tx], [sc * asp * sa, sc * (sh * sa + ca), ty], [px, py, 1.0]]
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    C, Ci = [[1.0, 0.0, cx], [0.0, 1.0, cy], [0.0, 0.0, 1.0]], [[1.0, 0.0, -cx], [0.0, 1.0, -cy], [0.0, 0.0, 1.0]]
    return _mm3(_mm3(C, M), Ci)


def _bilinear(canvas, x, y):
    """canvas [C, Hc, Wc] float64 sampled at float64 (x, y) arrays (edge-clamped), elementwise IEEE operations only."""
    Hc, Wc = canvas.shape[1:]
    x = np.clip(x, 0.0, Wc - 1.0)
    y = np.clip(y, 0.0, Hc - 1.0)
    x0 = np.minimum(np.floor(x), Wc - 2.0)
    y0 = np.minimum(np.floor(y), Hc - 2.0)
    fx, fy = x - x0, y - y0
    xi, yi = x0.astype(np.int64), y0.astype(np.int64)
    a, b = canvas[:, yi, xi], canvas[:, yi, xi + 1]
    c, d = canvas[:, yi + 1, xi], canvas[:, yi + 1, xi + 1]
    top = a + (b - a) * fx
    bot = c + (d - c) * fx
    return top + (bot - top) * fy


MARGIN = 160         # canvas border around view 0: the warped view's footprint stays inside it up to strength ~1.5 (clamped beyond)


def warped_pair(i, H=480, W=640, strength=1.0, noise=0.02):
    """Pair i of the viewpoint family: (view0, view1, h01) with view0 / view1 fp32 RGB [3,H,W] in [0,1] and h01 the 3x3 float32
    homography the TASKS see (datasets/hpatches.py:76-79's 'homography_matrix' with width = W, height = H).

    The scene is image_pair's blurred-noise canvas with a MARGIN-pixel border; view0 is its central crop; view1[r1, c1] is the
    canvas sampled bilinearly at G^-1 (c1, r1) for a seeded viewpoint homography G on pixel (column, row) coordinates, plus
    N(0, noise^2) noise, clipped.  The reference maps a keypoint at pixel centre (col, row) to (col + 0.5) / W * (W - 1)
    (extracter.py:149 with projection.py:147), so the matrix that is exact in ITS coordinates is h01 = S G S^-1 with
    S: x -> (x + 0.5) (W - 1) / W, y -> (y + 0.5) (H - 1) / H."""
    rng = np.random.default_rng(56780 + i)
    canvas = rng.random((3, H + 2 * MARGIN, W + 2 * MARGIN), dtype=np.float32)
    canvas = np.stack([_box_blur(c, 5) for c in canvas])
    lo, hi = canvas.min(), canvas.max()
    canvas = (canvas - lo) / (hi - lo)
    v0 = canvas[:, MARGIN:MARGIN + H, MARGIN:MARGIN + W].astype(np.float32)
    G = viewpoint_homography(rng, H, W, strength)
    Gi = _inv3(G)
    r1, c1 = np.mgrid[0:H, 0:W].astype(np.float64)
    den = Gi[2][0] * c1 + Gi[2][1] * r1 + Gi[2][2]
    x0 = (Gi[0][0] * c1 + Gi[0][1] * r1 + Gi[0][2]) / den
    y0 = (Gi[1][0] * c1 + Gi[1][1] * r1 + Gi[1][2]) / den
    v1 = _bilinear(canvas, x0 + MARGIN, y0 + MARGIN)
    v1 = np.clip(v1 + rng.normal(0.0, 1.0, size=(3, H, W)) * noise, 0.0, 1.0).astype(np.float32)
    sx, sy = (W - 1.0) / W, (H - 1.0) / H
    S = [[sx, 0.0, 0.5 * sx], [0.0, sy, 0.5 * sy], [0.0, 0.0, 1.0]]
    h01 = _mm3(_mm3(S, G), _inv3(S))
    h01 = np.array(h01, np.float64) / h01[2][2]
    return v0, v1, h01.astype(np.float32)


def viewpoint_case(i):
    """(strength, noise) of pair i in the graded viewpoint family the end-to-end metric tests use: four warp strengths x four noise
    levels, so that repeatability and the homography's corner error spread over easy, typical and failing pairs."""
    return (0.5, 1.0, 1.25, 1.5)[i % 4], (0.02, 0.07, 0.12, 0.16)[(i // 4) % 4]


# ---------------------------------------------------------------------------------------------- pose pairs (the AUC task's inputs)
def _rot(rx, ry, rz):
    import math
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = [[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]]
    Ry = [[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]]
    Rz = [[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]]
    return _mm3(Rz, _mm3(Ry, Rx))


def pose_case(i):
    """(strength, noise) of pose pair i: four motion strengths x four noise levels, as viewpoint_case."""
    return (0.5, 1.0, 1.5, 2.0)[i % 4], (0.02, 0.05, 0.09, 0.13)[(i // 4) % 4]


def pose_pair(i, H=480, W=640, strength=1.0, noise=0.02):
    """Pair i of the POSE family (tasks/AUC.py's inputs; r05): (view0, view1, K, pose01) -- two views of a scene made of TWO fronto-parallel
    planes (the left half of the blurred-noise canvas at depth 1, the right half at depth 1.6: points on one plane alone leave the
    essential matrix two-fold ambiguous), the second camera rotated by up to 2 / 2 / 5 degrees x strength about x / y / z and moved by
    0.06 x strength in a seeded direction.  K [3,3] float32 is the intrinsic matrix IN THE TASK'S pixel convention -- a keypoint at pixel
    centre (col, row) reaches the task as ((col + 0.5) (W - 1) / W, (row + 0.5) (H - 1) / H) (extracter.py:149, AUC.py:125-126), so
    K = S K_pixel with S that map -- and pose01 [4,4] float32 is T_0to1 = [R | t] (datasets/megadepth.py: 'pose01').
    view1[r1, c1] shows, of the two planes' pre-images under their homographies K (R + t n^T / d) K^-1, the nearer one that falls on its own
    half of the canvas (the other where only one does)."""
    import math
    rng = np.random.default_rng(91000 + i)
    canvas = rng.random((3, H + 2 * MARGIN, W + 2 * MARGIN), dtype=np.float32)
    canvas = np.stack([_box_blur(c, 5) for c in canvas])
    lo, hi = canvas.min(), canvas.max()
    canvas = (canvas - lo) / (hi - lo)
    v0 = canvas[:, MARGIN:MARGIN + H, MARGIN:MARGIN + W].astype(np.float32)
    u = [float(v) for v in rng.uniform(-1.0, 1.0, size=6)]
    R = _rot(math.radians(2.0) * strength * u[0], math.radians(2.0) * strength * u[1], math.radians(5.0) * strength * u[2])
    tv = [u[3] + (0.5 if u[3] >= 0 else -0.5), u[4], 0.6 * u[5]]                 # never along the optical axis alone
    nt = math.sqrt(tv[0] ** 2 + tv[1] ** 2 + tv[2] ** 2)
    t = [0.06 * strength * c / nt for c in tv]
    f, cx, cy = 520.0, (W - 1) / 2.0, (H - 1) / 2.0
    K = [[f, 0.0, cx], [0.0, f, cy], [0.0, 0.0, 1.0]]
    Ki = _inv3(K)
    r1, c1 = np.mgrid[0:H, 0:W].astype(np.float64)
    split = (W - 1) / 2.0                                                        # view-0 column where the scene's depth jumps
    src = []
    for d, left in ((1.0, True), (1.6, False)):
        M = [[R[a][b] + (t[a] / d if b == 2 else 0.0) for b in range(3)] for a in range(3)]       # R + t n^T / d, n = (0, 0, 1)
        Gi = _inv3(_mm3(_mm3(K, M), Ki))
        den = Gi[2][0] * c1 + Gi[2][1] * r1 + Gi[2][2]
        x0 = (Gi[0][0] * c1 + Gi[0][1] * r1 + Gi[0][2]) / den
        y0 = (Gi[1][0] * c1 + Gi[1][1] * r1 + Gi[1][2]) / den
        src.append((x0, y0, (x0 <= split) if left else (x0 > split)))
    (xa, ya, oka), (xb, yb, okb) = src
    use_a = oka | ~okb                                                           # the near plane wherever its pre-image lies on its half, else the far one
    x0, y0 = np.where(use_a, xa, xb), np.where(use_a, ya, yb)
    v1 = _bilinear(canvas, x0 + MARGIN, y0 + MARGIN)
    v1 = np.clip(v1 + rng.normal(0.0, 1.0, size=(3, H, W)) * noise, 0.0, 1.0).astype(np.float32)
    sx, sy = (W - 1.0) / W, (H - 1.0) / H
    S = [[sx, 0.0, 0.5 * sx], [0.0, sy, 0.5 * sy], [0.0, 0.0, 1.0]]
    Kt = np.array(_mm3(S, K), np.float64).astype(np.float32)
    pose = np.eye(4, dtype=np.float64)
    pose[:3, :3] = np.array(R)
    pose[:3, 3] = np.array(t)
    return v0, v1, Kt, pose.astype(np.float32)
