/*
 * kpb_oracle.c -- CPU restatement of the keypoint_bench hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP path in keypoint_bench_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product never does.
 *
 * Every function restates one piece of the reference (linyicheng1/keypoint_bench) and cites it:
 *   kpbo_fast_nms      utils/extracter.py:6-100    (fast_nms, literal round-by-round semantics)
 *   kpbo_detection     utils/extracter.py:193-221  (detection = A1 nms -> A2 border -> A3 compaction -> A4 top-k)
 *   kpbo_sample        utils/matcher.py:221-226    (grid_sample bilinear, align_corners=True, zero padding)
 *   kpbo_match         utils/matcher.py:227-230    (skimage.feature.match_descriptors; third-party, NOT vendored
 *                                                   in the reference -> restated from its documented behaviour:
 *                                                   float64 cdist euclidean, argmin axis 1, cross-check with
 *                                                   argmin axis 0, strict < max_distance.  PARITY UNPINNED
 *                                                   for this one function)
 * Pinning: tests/test_oracle_golden.py checks these against the .npz fixtures in tests/golden, which were produced
 * by importing the reference's own utils/extracter.py and models/ALike.py (tests/golden/make_golden.py).
 *
 * Plain C99, no dependencies.  Build: gcc -O2 -fPIC -shared -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * A1  fast_nms (utils/extracter.py:6-100).
 * One round (lines 54-96):
 *   - unfold with zero padding nms_dist, window ks*ks in raster order (54-66)
 *   - argmax over the window; torch.argmax returns the FIRST index of the maximum, so the centre
 *     (index ks*ks/2) wins iff it is strictly greater than every earlier cell and >= every later
 *     cell, where cells outside the image hold 0.0 (69-70)
 *   - stop when the number of maxima did not change (73-78)
 *   - otherwise every pixel that has a maximum somewhere in its window, the pixel itself excluded,
 *     is set to 0.0 (81-96)
 * Returns the number of rounds executed (number of unfold/argmax evaluations).
 * ---------------------------------------------------------------------------------------------- */
int kpbo_fast_nms(const float* in, float* out, int H, int W, int nms_dist)
{
    const size_t P = (size_t)H * W;
    memcpy(out, in, P * sizeof(float));
    if (nms_dist == 0) return 0; /* extracter.py:40-41 */
    const int r = nms_dist;
    unsigned char* ismax = (unsigned char*)malloc(P);
    long count = -1;
    int rounds = 0;
    for (;;) {
        long new_count = 0;
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                const float c = out[(size_t)y * W + x];
                int is = 1;
                for (int dy = -r; dy <= r && is; ++dy) {
                    const int yy = y + dy;
                    for (int dx = -r; dx <= r; ++dx) {
                        if (dy == 0 && dx == 0) continue;
                        const int xx = x + dx;
                        const float v = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.0f : out[(size_t)yy * W + xx];
                        const int earlier = (dy < 0) || (dy == 0 && dx < 0);
                        if (earlier ? !(c > v) : !(c >= v)) { is = 0; break; }
                    }
                }
                ismax[(size_t)y * W + x] = (unsigned char)is;
                new_count += is;
            }
        }
        ++rounds;
        if (new_count == count) break;
        count = new_count;
        /* fold of the expanded mask with the centre channel zeroed, then masked_fill(fold>0, 0) */
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                if (!ismax[(size_t)y * W + x]) continue;
                const int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r;
                const int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
                for (int yy = y0; yy <= y1; ++yy)
                    for (int xx = x0; xx <= x1; ++xx)
                        if (yy != y || xx != x) out[(size_t)yy * W + xx] = 0.0f;
            }
        }
    }
    free(ismax);
    return rounds;
}

/* ------------------------------------------------------------------------------------------------
 * A2  remove_border_points (utils/extracter.py:164-190): zero a border_dist frame, AFTER nms.
 * Python slice semantics: [:b] and [-b:] clamp to the extent.
 * ---------------------------------------------------------------------------------------------- */
void kpbo_remove_border(float* map, int H, int W, int border)
{
    if (border <= 0) return;
    const int bx = border > W ? W : border, by = border > H ? H : border;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            if (x < bx || x >= W - bx || y < by || y >= H - by) map[(size_t)y * W + x] = 0.0f;
}

typedef struct { float s; int idx; } kpbo_cand;

static int cand_cmp_desc(const void* a, const void* b)
{
    const kpbo_cand* p = (const kpbo_cand*)a;
    const kpbo_cand* q = (const kpbo_cand*)b;
    if (p->s > q->s) return -1;
    if (p->s < q->s) return 1;
    return (p->idx > q->idx) - (p->idx < q->idx); /* ties: ascending raster index (tie rule) */
}

/* ------------------------------------------------------------------------------------------------
 * A5  detection (utils/extracter.py:193-221) = A1 -> A2 -> A3 -> A4.
 * A3 prob_map_to_positions_with_prob (129-161): raster-order compaction of map > threshold;
 *    x = (col + 0.5) / W, y = (row + 0.5) / H in float32 (nonzero().float() + 0.5 divided by an
 *    int64 [H, W] tensor promotes to float32); columns returned as (x, y, score) (161).
 * A4 (217-220): if N > top_k keep the top_k highest scores, in descending-score order
 *    (torch.argsort(descending=True); ties are unordered in the reference, here ascending raster
 *    index); then, if min_score > 0, keep score > min_score.
 * Returns n; fills out_kps[n*3] and out_idx[n] (flat raster index row*W+col).  cap = capacity.
 * ---------------------------------------------------------------------------------------------- */
int kpbo_detection(const float* score, int H, int W, int nms_dist, float threshold, int border,
                   int top_k, float min_score, float* out_kps, int* out_idx, int cap)
{
    const size_t P = (size_t)H * W;
    float* m = (float*)malloc(P * sizeof(float));
    kpbo_fast_nms(score, m, H, W, nms_dist);
    kpbo_remove_border(m, H, W, border);
    kpbo_cand* c = (kpbo_cand*)malloc(P * sizeof(kpbo_cand));
    int n = 0;
    for (size_t i = 0; i < P; ++i)
        if (m[i] > threshold) { c[n].s = m[i]; c[n].idx = (int)i; ++n; }
    if (n > top_k) {
        qsort(c, (size_t)n, sizeof(kpbo_cand), cand_cmp_desc);
        n = top_k;
    }
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if (min_score > 0.0f && !(c[i].s > min_score)) continue;
        if (k >= cap) break;
        const int row = c[i].idx / W, col = c[i].idx % W;
        out_kps[3 * k + 0] = ((float)col + 0.5f) / (float)W;
        out_kps[3 * k + 1] = ((float)row + 0.5f) / (float)H;
        out_kps[3 * k + 2] = c[i].s;
        out_idx[k] = c[i].idx;
        ++k;
    }
    free(c);
    free(m);
    return k;
}

/* ------------------------------------------------------------------------------------------------
 * M1  descriptor sampling (utils/matcher.py:221-226).
 *   grid = (xy - 0.5) * 2 ; F.grid_sample(desc_map, grid, align_corners=True)  (bilinear, zeros)
 * ATen's CPU kernel un-normalises with (g + 1) * ((size - 1) / 2), takes w = x - floor(x),
 * e = 1 - w (same for y) and sums nw*s*e + ne*s*w + sw*n*e + se*n*w with out-of-range taps = 0.
 * desc is addressed with explicit element strides (sc, sh, sw) so NCHW and channels-last both work.
 * pts has row stride pstride floats, columns 0,1 = (x, y) normalised.
 * ---------------------------------------------------------------------------------------------- */
void kpbo_sample(const float* desc, int C, int Hd, int Wd, long sc, long sh, long sw_,
                 const float* pts, int n, int pstride, float* out)
{
    const float fx = (float)(Wd - 1) / 2.0f, fy = (float)(Hd - 1) / 2.0f;
    for (int i = 0; i < n; ++i) {
        const float gx = (pts[(size_t)i * pstride + 0] - 0.5f) * 2.0f;
        const float gy = (pts[(size_t)i * pstride + 1] - 0.5f) * 2.0f;
        const float x = (gx + 1.0f) * fx, y = (gy + 1.0f) * fy;
        const float xw = floorf(x), yn = floorf(y);
        const float w = x - xw, e = 1.0f - w, nn = y - yn, s = 1.0f - nn;
        const float c_nw = s * e, c_ne = s * w, c_sw = nn * e, c_se = nn * w;
        const long x0 = (long)xw, y0 = (long)yn, x1 = x0 + 1, y1 = y0 + 1;
        const int vx0 = x0 >= 0 && x0 < Wd, vx1 = x1 >= 0 && x1 < Wd;
        const int vy0 = y0 >= 0 && y0 < Hd, vy1 = y1 >= 0 && y1 < Hd;
        for (int ch = 0; ch < C; ++ch) {
            const float* p = desc + (size_t)ch * sc;
            const float nw = (vx0 && vy0) ? p[y0 * sh + x0 * sw_] : 0.0f;
            const float ne = (vx1 && vy0) ? p[y0 * sh + x1 * sw_] : 0.0f;
            const float sw = (vx0 && vy1) ? p[y1 * sh + x0 * sw_] : 0.0f;
            const float se = (vx1 && vy1) ? p[y1 * sh + x1 * sw_] : 0.0f;
            out[(size_t)i * C + ch] = nw * c_nw + ne * c_ne + sw * c_sw + se * c_se;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * M2  skimage.feature.match_descriptors(d0, d1, metric='euclidean', max_distance, cross_check)
 * as called at utils/matcher.py:227-230 (restated; see file header).
 *   distances = scipy cdist(float64): sqrt(sum_k (a_k - b_k)^2), k ascending, no FMA
 *   indices2 = argmin(distances, axis=1) (first index on ties)
 *   cross_check: keep i where argmin(distances, axis=0)[indices2[i]] == i
 *   keep distances[i, indices2[i]] < max_distance (strict)
 * out_pairs[k*2] = (i, j) ascending i; out_dist[k] = distance.  Returns K.
 * ---------------------------------------------------------------------------------------------- */
int kpbo_match(const float* d0, int n, const float* d1, int m, int C, double max_distance,
               int cross_check, int* out_pairs, double* out_dist)
{
    if (n == 0 || m == 0) return 0;
    double* rowmin = (double*)malloc((size_t)n * sizeof(double));
    int* rowarg = (int*)malloc((size_t)n * sizeof(int));
    double* colmin = (double*)malloc((size_t)m * sizeof(double));
    int* colarg = (int*)malloc((size_t)m * sizeof(int));
    for (int j = 0; j < m; ++j) { colmin[j] = INFINITY; colarg[j] = 0; }
    for (int i = 0; i < n; ++i) {
        double best = INFINITY; int arg = 0;
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int k = 0; k < C; ++k) {
                const double d = (double)d0[(size_t)i * C + k] - (double)d1[(size_t)j * C + k];
                s += d * d;
            }
            const double dist = sqrt(s);
            /* numpy.argmin: the first NaN wins, otherwise the first minimum */
            if ((dist < best || (dist != dist && best == best)) ) { best = dist; arg = j; }
            if ((dist < colmin[j] || (dist != dist && colmin[j] == colmin[j]))) { colmin[j] = dist; colarg[j] = i; }
        }
        rowmin[i] = best; rowarg[i] = arg;
    }
    int K = 0;
    for (int i = 0; i < n; ++i) {
        const int j = rowarg[i];
        if (cross_check && colarg[j] != i) continue;
        /* skimage applies the distance filter only `if max_distance < np.inf` (so inf / nan distances survive max_distance = inf) */
        if (max_distance < INFINITY && !(rowmin[i] < max_distance)) continue;
        out_pairs[2 * K] = i; out_pairs[2 * K + 1] = j; out_dist[K] = rowmin[i];
        ++K;
    }
    free(rowmin); free(rowarg); free(colmin); free(colarg);
    return K;
}

/* ------------------------------------------------------------------------------------------------
 * covisibility warp + ground-truth mutual nearest neighbours.
 *
 * warp_homography, utils/projection.py:137-167.  kps rows are (x, y[, ...]) normalised; hm is the 3x3
 * homography, row-major fp32.  Arithmetic as torch 2.10 CPU evaluates the reference's lines here
 * (pinned by tests/golden/covis.npz): x*(w-1), y*(h-1) in fp32 (144); the einsum row (146) is
 * h0*x, then fused-multiply-add of h1*y, then + h2; divide by the third row (147); keep points with
 * 0 <= u <= w-1 and 0 <= v <= h-1 (153); both outputs divided by (w-1, h-1) (163-164).
 * ids receives the kept row numbers ascending (156) followed by the rejected ones ascending (157).
 * Returns the number kept. */
int kpbo_warp_homography(const float* kps, int n, int stride, const float* hm, int width, int height, int fused,
                         float* kps0_valid, float* kps01_valid, int* ids)
{
    if (fused < 0) fused = 9 * n > 400;
    const float sx = (float)(width - 1), sy = (float)(height - 1);
    int k = 0, nout = 0;
    int* rejected = (int*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    for (int i = 0; i < n; ++i) {
        const float x = kps[(size_t)i * stride] * sx, y = kps[(size_t)i * stride + 1] * sy;
        float r[3];
        for (int q = 0; q < 3; ++q)
            r[q] = (fused ? fmaf(hm[3 * q + 1], y, hm[3 * q] * x) : hm[3 * q] * x + hm[3 * q + 1] * y) + hm[3 * q + 2];
        const float u = r[0] / r[2], v = r[1] / r[2];
        if (u >= 0.f && u <= sx && v >= 0.f && v <= sy) {
            kps0_valid[2 * k] = x / sx;  kps0_valid[2 * k + 1] = y / sy;
            kps01_valid[2 * k] = u / sx; kps01_valid[2 * k + 1] = v / sy;
            ids[k++] = i;
        } else {
            rejected[nout++] = i;
        }
    }
    for (int i = 0; i < nout; ++i) ids[k + i] = rejected[i];
    free(rejected);
    return k;
}

/* compute_keypoints_distance, tasks/repeatability.py:39-51: torch.norm(p=2) over the pair (dx, dy) evaluates
 * sqrt(fma(dy, dy, dx*dx)) on this CPU build (pinned by the fixture). */
static float kp_dist(const float* a, const float* b)
{
    const float dx = a[0] - b[0], dy = a[1] - b[1];
    return sqrtf(fmaf(dy, dy, dx * dx));
}

/* val_key_points after the two warps, tasks/repeatability.py:69-92 (mutual_argmax 9-32).
 * k0[M,2], k01[M,2] = covisible keypoints of image 0 and their warps; k1[N,2], k10[N,2] likewise for image 1.
 *   dm[i][j] = (|k0_i - k10_j| + |k1_j - k01_i|) / 2   (69-71);  dm[i][i] = 99999 for i < min(M,N)   (72-73)
 *   value = -dm;  value -= value.min()  (18, 36);  mutual where value equals its row max and its column max,
 *   ALL ties kept, listed row-major (20-32)
 *   dist = dm[pairs] * scale01   (75-81);  errors[i] = min_j dm[i][j] * scale10   (77, 81, 85)
 * pairs/dist get at most cap entries; returns the number of mutual cells (may exceed cap: caller checks). */
long kpbo_val_keypoints(const float* k0, const float* k01, int M, const float* k1, const float* k10, int N,
                        float scale01, float scale10, int* pairs, float* dist, long cap, float* errors)
{
    float* dm = (float*)malloc((size_t)M * N * sizeof(float));
    float* rmin = (float*)malloc((size_t)M * sizeof(float));
    float* cmin = (float*)malloc((size_t)N * sizeof(float));
    float dmax = -INFINITY;
    const int nd = M < N ? M : N;
    for (int j = 0; j < N; ++j) cmin[j] = INFINITY;
    for (int i = 0; i < M; ++i) {
        float best = INFINITY;
        for (int j = 0; j < N; ++j) {
            float d = (kp_dist(k0 + 2 * i, k10 + 2 * j) + kp_dist(k1 + 2 * j, k01 + 2 * i)) / 2.f;
            if (i == j && i < nd) d = 99999.f;
            dm[(size_t)i * N + j] = d;
            if (d < best) best = d;
            if (d < cmin[j]) cmin[j] = d;
            if (d > dmax) dmax = d;
        }
        rmin[i] = best;
        errors[i] = best * scale10;
    }
    /* value = (-dm) - min(-dm) = (-dm) - (-dmax); rounding is monotone, so the row/column maxima of value are the
     * images of the row/column minima of dm */
    long K = 0;
    for (int i = 0; i < M; ++i) {
        const float vr = (-rmin[i]) - (-dmax);
        for (int j = 0; j < N; ++j) {
            const float v = (-dm[(size_t)i * N + j]) - (-dmax);
            const float vc = (-cmin[j]) - (-dmax);
            if (v == vr && v == vc) {
                if (K < cap) { pairs[2 * K] = i; pairs[2 * K + 1] = j; dist[K] = dm[(size_t)i * N + j] * scale01; }
                ++K;
            }
        }
    }
    free(dm); free(rmin); free(cmin);
    return K;
}

/* ------------------------------------------------------------------------------------------------
 * the tensor Lucas-Kanade tracker, utils/matcher.py:7-142 (OpticalFlow).
 *
 * The reference unfolds six win*win*C-channel patch maps per level (93-107) and grid_samples them at the
 * keypoints (110, 116-119).  Sampling the unfolded map at (px, py) equals, for window cell (ky, kx),
 *     sum over the four bilinear taps (xt, yt) of (px, py) that lie inside the image (grid_sample's zero padding)
 *         w_t * img[c][yt + ky - r][xt + kx - r]        (0 outside the image: unfold's zero padding, 93)
 * which is what sample_cell() computes, so nothing is materialised.  All arithmetic fp32 as in the reference.
 * Sums over the window run in (c, ky, kx) order; torch's einsum order differs, so agreement with the reference is
 * to rounding (see tests/test_oracle_lk.py for the tolerance), not bit for bit. */
static float lk_pix(const float* img, int H, int W, int y, int x)
{
    return (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.0f;
}

/* Sobel pair of matcher.py:23-24 as conv2d (cross-correlation, zero padding 1) evaluates it (87-90) */
static void lk_sobel(const float* img, int H, int W, float* dx, float* dy)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float sx = 0.f, sy = 0.f;
            static const float kx[3][3] = {{1, 0, -1}, {2, 0, -2}, {1, 0, -1}};
            static const float ky[3][3] = {{1, 2, 1}, {0, 0, 0}, {-1, -2, -1}};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    const float v = lk_pix(img, H, W, y + i - 1, x + j - 1);
                    sx += kx[i][j] * v; sy += ky[i][j] * v;
                }
            dx[(size_t)y * W + x] = sx; dy[(size_t)y * W + x] = sy;
        }
}

typedef struct { int x0, y0; float w[4]; int ok[4]; } lk_taps;

/* grid_sample(align_corners=True, bilinear, zeros) tap set of pixel position (px, py), reached through the
 * reference's normalise (109, 115) / ATen's unnormalise round trip */
static lk_taps lk_make_taps(float px, float py, int H, int W)
{
    lk_taps t;
    const float gx = px / (float)(W - 1) * 2.0f - 1.0f, gy = py / (float)(H - 1) * 2.0f - 1.0f;
    const float ix = (gx + 1.0f) * ((float)(W - 1) / 2.0f), iy = (gy + 1.0f) * ((float)(H - 1) / 2.0f);
    const float fx = floorf(ix), fy = floorf(iy);
    const float wx = ix - fx, wy = iy - fy, ex = 1.0f - wx, sy = 1.0f - wy;
    t.x0 = (int)fx; t.y0 = (int)fy;
    t.w[0] = sy * ex; t.w[1] = sy * wx; t.w[2] = wy * ex; t.w[3] = wy * wx;       /* nw, ne, sw, se */
    for (int k = 0; k < 4; ++k) {
        const int xt = t.x0 + (k & 1), yt = t.y0 + (k >> 1);
        t.ok[k] = xt >= 0 && xt < W && yt >= 0 && yt < H;
    }
    return t;
}

static float lk_sample_cell(const float* img, int H, int W, const lk_taps* t, int oy, int ox)
{
    float s = 0.0f;
    for (int k = 0; k < 4; ++k)
        if (t->ok[k]) s += lk_pix(img, H, W, t->y0 + (k >> 1) + oy, t->x0 + (k & 1) + ox) * t->w[k];
    return s;
}

/* optical_flow_level, matcher.py:77-133, for one image pair.  img1/img2 [C][H][W]; pts1/pts2 [n][2] pixel positions
 * at this level; pts2 is updated in place (pts_pre). */
static void lk_level(const float* img1, const float* img2, int C, int H, int W, const float* pts1, float* pts2, int n,
                     int win, int iters)
{
    const int r = win / 2, E = C * win * win;
    const size_t P = (size_t)H * W;
    float* dx2 = (float*)malloc(P * C * sizeof(float));
    float* dy2 = (float*)malloc(P * C * sizeof(float));
    float* p1 = (float*)malloc((size_t)E * sizeof(float));
    for (int c = 0; c < C; ++c) lk_sobel(img2 + c * P, H, W, dx2 + c * P, dy2 + c * P);
    for (int i = 0; i < n; ++i) {
        const lk_taps t1 = lk_make_taps(pts1[2 * i], pts1[2 * i + 1], H, W);
        int e = 0;
        for (int c = 0; c < C; ++c)
            for (int ky = 0; ky < win; ++ky)
                for (int kx = 0; kx < win; ++kx) p1[e++] = lk_sample_cell(img1 + c * P, H, W, &t1, ky - r, kx - r);
        float px = pts2[2 * i], py = pts2[2 * i + 1];
        for (int it = 0; it < iters; ++it) {
            const lk_taps t = lk_make_taps(px, py, H, W);
            float g00 = 0.f, g01 = 0.f, g11 = 0.f, b0 = 0.f, b1 = 0.f;
            e = 0;
            for (int c = 0; c < C; ++c)
                for (int ky = 0; ky < win; ++ky)
                    for (int kx = 0; kx < win; ++kx, ++e) {
                        const float v = lk_sample_cell(img2 + c * P, H, W, &t, ky - r, kx - r);
                        const float jx = lk_sample_cell(dx2 + c * P, H, W, &t, ky - r, kx - r);
                        const float jy = lk_sample_cell(dy2 + c * P, H, W, &t, ky - r, kx - r);
                        const float dI = p1[e] - v;                                   /* 117 */
                        g00 += jx * jx; g01 += jx * jy; g11 += jy * jy;               /* 121 */
                        b0 += dI * jx; b1 += dI * jy;                                 /* 122 */
                    }
            const float det = g00 * g11 - g01 * g01;                                  /* 123 */
            if (det > 1e-6f) {
                /* inverse (124); the update einsum 'bik,bk->bk' (125) sums the inverse over its SECOND index and
                 * multiplies row-wise: dx = (inv00 + inv01) * b0, dy = (inv10 + inv11) * b1 */
                const float i00 = g11 / det, i01 = -g01 / det, i11 = g00 / det;
                px = px - (i00 + i01) * b0;
                py = py - (i01 + i11) * b1;
            }
        }
        pts2[2 * i] = px; pts2[2 * i + 1] = py;
    }
    free(dx2); free(dy2); free(p1);
}

static void lk_avgpool(const float* img, int C, int H, int W, int k, float* out)
{
    const int Ho = H / k, Wo = W / k;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float s = 0.f;
                for (int i = 0; i < k; ++i)
                    for (int j = 0; j < k; ++j) s += img[(size_t)c * H * W + (size_t)(y * k + i) * W + x * k + j];
                out[(size_t)c * Ho * Wo + (size_t)y * Wo + x] = s / (float)(k * k);
            }
}

/* OpticalFlow.__call__, matcher.py:48-75.  pts1, pts2 [n][2] normalised; unit [n][2] = (cos, sin) of the random angles
 * (55-56; drawn by the caller so that runs can be compared).  out_pts [n][2] in PIXELS of the full image (as the
 * reference returns them), out_err [n] (73). */
void kpbo_lk_track(const float* img1, const float* img2, int C, int H, int W, const float* pts1, const float* pts2,
                   const float* unit, int n, float distance, int win, int levels, int iters, float* out_pts, float* out_err)
{
    float* p1 = (float*)malloc((size_t)n * 2 * sizeof(float));
    float* p2 = (float*)malloc((size_t)n * 2 * sizeof(float));
    float* cur = (float*)malloc((size_t)n * 2 * sizeof(float));
    float* l1 = (float*)malloc((size_t)n * 2 * sizeof(float));
    for (int i = 0; i < n; ++i) {
        p1[2 * i] = pts1[2 * i] * (float)(W - 1); p1[2 * i + 1] = pts1[2 * i + 1] * (float)(H - 1);      /* 51 */
        p2[2 * i] = pts2[2 * i] * (float)(W - 1); p2[2 * i + 1] = pts2[2 * i + 1] * (float)(H - 1);      /* 52 */
        float x = p2[2 * i] + unit[2 * i] * distance, y = p2[2 * i + 1] + unit[2 * i + 1] * distance;    /* 59 */
        x = fminf(fmaxf(x, 10.f), (float)(W - 10)); y = fminf(fmaxf(y, 10.f), (float)(H - 10));          /* 60-61 */
        cur[2 * i] = x; cur[2 * i + 1] = y;
    }
    for (int lv = 0; lv < levels; ++lv) {                                                                /* 76-87 */
        const int idx = levels - lv - 1;
        const float scale = (float)(1 << idx);
        const int k = idx == 0 ? 1 : 2 * idx;                      /* build_pyramid (45): level i>0 is avg_pool2d(img, 2i, 2i) */
        const int Hl = H / k, Wl = W / k;
        float *a = (float*)img1, *b = (float*)img2;
        if (k > 1) {
            a = (float*)malloc((size_t)C * Hl * Wl * sizeof(float)); b = (float*)malloc((size_t)C * Hl * Wl * sizeof(float));
            lk_avgpool(img1, C, H, W, k, a); lk_avgpool(img2, C, H, W, k, b);
        }
        for (int i = 0; i < 2 * n; ++i) { l1[i] = p1[i] / scale; cur[i] = cur[i] / scale; }               /* 79-80 */
        lk_level(a, b, C, Hl, Wl, l1, cur, n, win, iters);
        for (int i = 0; i < 2 * n; ++i) cur[i] = cur[i] * scale;                                          /* 85 */
        if (k > 1) { free(a); free(b); }
    }
    for (int i = 0; i < n; ++i) {
        const float dx = cur[2 * i] - p2[2 * i], dy = cur[2 * i + 1] - p2[2 * i + 1];
        out_pts[2 * i] = cur[2 * i]; out_pts[2 * i + 1] = cur[2 * i + 1];
        out_err[i] = fminf(sqrtf(dx * dx + dy * dy), 8.0f);                                               /* 73 */
    }
    free(p1); free(p2); free(cur); free(l1);
}

/* ------------------------------------------------------------------------------------------------
 * warp_se3, utils/projection.py:195-268 (with interpolate_depth 271-373, unproject 30-53, project 56-75): the
 * depth-based covisibility warp of the 'se3' datasets.  All fp32, formulas as torch's CPU kernels evaluate them
 * (fixture tests/golden/se3.npz).  The three einsums are [n,k] x [k,k] matmuls over the n0 points that survive
 * interpolate_depth in image 0: this torch build hands them to its BLAS (fused multiply-add chain a0*b0, fma, fma ...)
 * when m*n*k >= 400 and to its own unfused loop below that; `fused` < 0 follows that rule, 0/1 force a form. */
static int se3_interp(const float* depth, int h, int w, float x, float y, float* z)
{
    const int border = 10;                                                       /* 272 */
    const float i = y, j = x;                                                    /* 273: (w,h) -> (i,j) */
    const long it = (long)floorf(i), jt = (long)floorf(j), ib = (long)ceilf(i), jr = (long)ceilf(j);
    if (!(it >= border && jt >= border && jr < w - border && ib < h - border)) return 0;     /* 285-301: corners */
    const float dtl = depth[it * w + jt], dtr = depth[it * w + jr], dbl = depth[ib * w + jt], dbr = depth[ib * w + jr];
    if (!(dtl > 0.f && dtr > 0.f && dbl > 0.f && dbr > 0.f)) return 1;           /* 322-325: valid corners, no depth */
    const float di = i - (float)it, dj = j - (float)jt;                          /* 348-349 */
    const float wtl = (1.f - di) * (1.f - dj), wtr = (1.f - di) * dj, wbl = di * (1.f - dj), wbr = di * dj;
    *z = ((wtl * dtl + wtr * dtr) + wbl * dbl) + wbr * dbr;                      /* 355-358 */
    return 2;
}

static float se3_dot3(const float* a, float b0, float b1, float b2, int fused)
{
    return fused ? fmaf(a[2], b2, fmaf(a[1], b1, a[0] * b0)) : (a[0] * b0 + a[1] * b1) + a[2] * b2;
}

/* kps [n][stride] normalised; depth0 [H0][W0], depth1 [H1][W1]; kinv0 = inverse(intrinsics0) (computed by the caller with
 * torch.inverse, as unproject does), k1 = intrinsics1, pose = pose01 [4][4], bbox0 / bbox1 = (row, col).
 * out_k0 / out_k01 [n][2]; out_valid [n] ids; out_out [n] = ids_outside then ids_occlude; counts[0..1] = their lengths. */
void kpbo_warp_se3(const float* kps, int n, int stride, const float* depth0, int H0, int W0, const float* depth1, int H1, int W1,
                   const float* kinv0, const float* k1, const float* pose, const float* bbox0, const float* bbox1, int fused,
                   float* out_k0, float* out_k01, int* out_valid, int* out_out, int* counts)
{
    int n0 = 0;
    for (int i = 0; i < n; ++i) {
        float z;
        n0 += se3_interp(depth0, H0, W0, kps[(size_t)i * stride] * (float)W0, kps[(size_t)i * stride + 1] * (float)H0, &z) == 2;
    }
    const int f3 = fused < 0 ? 9 * n0 >= 400 : fused, f4 = fused < 0 ? 16 * n0 >= 400 : fused;
    int nv = 0, nout = 0, nocc = 0;
    int* occ = (int*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    for (int i = 0; i < n; ++i) {
        const float x = kps[(size_t)i * stride] * (float)W0, y = kps[(size_t)i * stride + 1] * (float)H0;      /* 204 */
        float z0;
        if (se3_interp(depth0, H0, W0, x, y, &z0) != 2) continue;                                   /* 211: not in ids0 */
        const float bu = (x + bbox0[1]) + 0.5f, bv = (y + bbox0[0]) + 0.5f;                         /* 214 */
        const float d0 = bu * z0, d1 = bv * z0;                                                     /* 42 */
        float p[3], q[3], zuv[3];
        for (int r = 0; r < 3; ++r) p[r] = se3_dot3(kinv0 + 3 * r, d0, d1, z0, f3);                 /* 46 */
        for (int r = 0; r < 3; ++r) {                                                               /* 221 */
            const float* t = pose + 4 * r;
            q[r] = f4 ? fmaf(t[3], 1.0f, fmaf(t[2], p[2], fmaf(t[1], p[1], t[0] * p[0]))) : ((t[0] * p[0] + t[1] * p[1]) + t[2] * p[2]) + t[3] * 1.0f;
        }
        for (int r = 0; r < 3; ++r) zuv[r] = se3_dot3(k1 + 3 * r, q[0], q[1], q[2], f3);            /* 68 */
        const float u = zuv[0] / zuv[2], v = zuv[1] / zuv[2], z01 = zuv[2];                         /* 73-75 */
        const float u01 = (u - bbox1[1]) - 0.5f, v01 = (v - bbox1[0]) - 0.5f;                       /* 227 */
        float z1;
        const int c = se3_interp(depth1, H1, W1, u01, v01, &z1);                                    /* 234 */
        if (c == 0) { out_out[nout++] = i; continue; }                                              /* 236-239: ids_outside */
        if (c == 1) continue;                                                                       /* corners but no depth: neither list */
        if (fabsf(z01 - z1) < 0.05f) {                                                              /* 246 */
            out_k0[2 * nv] = x / (float)W0; out_k0[2 * nv + 1] = y / (float)H0;                     /* 266 */
            out_k01[2 * nv] = u01 / (float)W1; out_k01[2 * nv + 1] = v01 / (float)H1;               /* 267 */
            out_valid[nv++] = i;
        } else {
            occ[nocc++] = i;                                                                        /* 249 */
        }
    }
    for (int i = 0; i < nocc; ++i) out_out[nout + i] = occ[i];                                      /* 262 */
    counts[0] = nv; counts[1] = nout + nocc;
    free(occ);
}
