/*
 * kpb.h -- C ABI of libkpb.so: the MI355X (gfx950) keypoint extract -> NMS/top-K -> brute-force
 * match hot path of linyicheng1/keypoint_bench.
 *
 * The reference has no FFI layer of its own: its seams are three Python call signatures
 * (SURVEY.md section 8b).  Each entry point below names the reference code it replaces; the ctypes
 * binding a maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer named *_dev is a DEVICE pointer (e.g. torch.Tensor.data_ptr()); the caller owns
 *     all input and output buffers; nothing is allocated per call once a shape has been seen.
 *   - all work is enqueued on the context's HIP stream and is asynchronous unless stated.
 *   - return value: 0 = OK, < 0 = error (KPB_E_*); kpb_last_error() gives the text.
 *   - one context per process/GPU, single caller thread (the reference runs batch_size 1 from one
 *     Python thread, config/config_MHA.yaml:10).  ctypes releases the GIL during calls.
 *   - "batch" lets the runner push many independent images/pairs through one launch wave; the
 *     reference semantics are those of batch element 0 repeated (utils/extracter.py:161).
 */
#ifndef KPB_H
#define KPB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KPB_VERSION 1

#define KPB_OK 0
#define KPB_E_INVALID (-1)      /* bad argument: a caller's mistake, never a reason to route the call elsewhere */
#define KPB_E_HIP (-2)          /* a HIP runtime call failed */
#define KPB_E_NOMEM (-3)        /* workspace allocation failed */
#define KPB_E_NEGATIVE (-4)     /* score map holds negative values: outside this path's contract */
#define KPB_E_NOT_CONVERGED (-5)/* NMS sweeps exhausted (only from the *_nosync pipeline paths) */
#define KPB_E_WEIGHTS (-6)      /* malformed weight blob */
#define KPB_E_UNSUPPORTED (-7)  /* a well-formed request beyond a documented limit of the kernels (nms_dist, top_k, matches per pair,
                                   tracker window ...): the one code, with KPB_E_NEGATIVE, on which install() hands the call
                                   to the reference's own function */

#define KPB_MAX_NMS_DIST 16
#define KPB_MAX_TOPK 8192       /* bound of the on-chip sort, applies only when top_k < H*W */

typedef struct kpb_ctx kpb_ctx;
typedef struct kpb_net kpb_net;

/* extractor_params of config/config_MHA.yaml:68-73, read at utils/extracter.py:207-211 */
typedef struct kpb_detect_params {
    int32_t nms_dist;
    float threshold;
    int32_t border_dist;
    int32_t top_k;
    float min_score;
} kpb_detect_params;

/* brute_force_params of config/config_MHA.yaml:82-85 (metric is always euclidean here) */
typedef struct kpb_match_params {
    double max_distance;
    int32_t cross_check;
} kpb_match_params;

int kpb_version(void);
/* ctx may be NULL: returns the message of the last failed call that had no context. */
const char* kpb_last_error(const kpb_ctx* ctx);

/* device = HIP device ordinal; stream = hipStream_t to enqueue on (NULL = the default stream).
 * Pass torch.cuda.current_stream().cuda_stream to order against torch work. */
int kpb_ctx_create(int device, void* stream, kpb_ctx** out);
/* Re-targets the context (after synchronising the old stream), e.g. when torch's current stream changed. */
int kpb_ctx_set_stream(kpb_ctx* ctx, void* stream);
void kpb_ctx_destroy(kpb_ctx* ctx);
int kpb_sync(kpb_ctx* ctx);

/* Limits of a context a caller may move (the reference has no counterpart: its tensors live wherever torch puts them).
 *   KPB_OPT_COVIS_STORE_BYTES  kpb_val_keypoints keeps its M x N distance cells (4 bytes each) in the context's workspace when they fit
 *                              this many bytes (default 4 GiB) and re-evaluates them in every pass when they do not -- or when the
 *                              workspace cannot grow; both forms give the same bits (tests/test_gpu_covis.py runs both).  0 = never keep.
 * Returns KPB_E_INVALID for an unknown option or a negative value. */
#define KPB_OPT_COVIS_STORE_BYTES 1
int kpb_ctx_set_option(kpb_ctx* ctx, int option, int64_t value);

/* Per-kernel timing for bench.py's roofline leg: when enabled every kernel launch is bracketed by two
 * HIP events on the context's stream.  kpb_prof_report synchronises, writes one text line per kernel
 * ("<name> <launches> <total_ms>\n") into buf and clears the records. */
int kpb_prof_enable(kpb_ctx* ctx, int on);
int kpb_prof_report(kpb_ctx* ctx, char* buf, size_t cap);

/* ---- A1: utils/extracter.py:6-100 fast_nms --------------------------------------------------
 * score_dev [batch][H][W] fp32, non-negative; out_map_dev same shape (may not alias score_dev).
 * Synchronous: iterates sweeps until every image reached the fixed point. */
int kpb_fast_nms(kpb_ctx* ctx, const float* score_dev, int batch, int H, int W, int nms_dist,
                 float* out_map_dev);

/* ---- A5: utils/extracter.py:193-221 detection (A1 nms, A2 border, A3 compaction, A4 top-k) ---
 * score_dev [batch][H][W] fp32 (the reference's [1,1,H,W]); not modified.
 * cap = min(top_k, H*W) rows are reserved per image:
 * out_kps_dev [batch][cap][3] = (x, y, score), x = (col+0.5)/W, y = (row+0.5)/H;
 * out_idx_dev [batch][cap] flat raster index row*W+col (may be NULL);
 * out_n_dev   [batch] number of valid rows (<= cap).
 * Row order: raster when N <= top_k, descending score (ties: ascending raster index) otherwise.
 * sync != 0: blocks until done, re-running sweeps for images that had not converged, and
 * returns KPB_E_NEGATIVE if any map held a negative score.  sync == 0: enqueue only; call
 * kpb_detect_check() later -- score_dev and the outputs must stay alive until then, and a second kpb_detect before
 * that check is refused (KPB_E_INVALID): the context holds ONE pending detection. */
int kpb_detect(kpb_ctx* ctx, const float* score_dev, int batch, int H, int W,
               const kpb_detect_params* params, float* out_kps_dev, int32_t* out_idx_dev,
               int32_t* out_n_dev, int sync);
/* Completes the last kpb_detect(sync=0): synchronises and, if some image had not reached the NMS fixed
 * point within the enqueued sweeps, runs more sweeps and rewrites the outputs.  Returns 0 (outputs were
 * final), 1 (outputs were rewritten: redo whatever consumed them), or KPB_E_NEGATIVE. */
int kpb_detect_check(kpb_ctx* ctx);
/* Keypoint counts of the last COMPLETED kpb_detect (sync = 1, or sync = 0 + kpb_detect_check), as host integers: the kernels leave
 * them in pinned host memory, so the `N` of the reference's `detection(...) -> Tensor[N, 3]` (utils/extracter.py:217-221) costs no
 * second device read-back.  out_n_host [batch] is host memory; batch must be the batch of that call. */
int kpb_detect_counts(kpb_ctx* ctx, int32_t* out_n_host, int batch);

/* ---- M1: utils/matcher.py:221-226 descriptor sampling (grid_sample, align_corners=True) -----
 * desc_dev: [batch] maps of C channels, Hd x Wd, element strides (sb, sc, sh, sw) so both NCHW and
 * channels-last tensors work.  pts_dev [batch][max_n][pts_cols] (cols 0,1 = x,y normalised).
 * n_dev [batch] valid rows per item, or NULL = max_n everywhere.  out_dev [batch][max_n][C]. */
int kpb_sample(kpb_ctx* ctx, const float* desc_dev, int batch, int C, int Hd, int Wd, int64_t sb,
               int64_t sc, int64_t sh, int64_t sw, const float* pts_dev, int pts_cols, int max_n,
               const int32_t* n_dev, float* out_dev);

/* ---- M2: skimage.feature.match_descriptors as called at utils/matcher.py:227-230 -------------
 * d0_dev [batch][max_n][C], d1_dev [batch][max_m][C] fp32; n_dev/m_dev [batch] or NULL.
 * float64 euclidean distances, argmin with first-index ties, optional cross-check, strict
 * < max_distance.  out_pairs_dev [batch][max_n][2] int32 (i, j) ascending i;
 * out_dist_dev [batch][max_n] float64 (may be NULL); out_k_dev [batch]. */
int kpb_match(kpb_ctx* ctx, const float* d0_dev, const float* d1_dev, int batch, int C, int max_n,
              int max_m, const int32_t* n_dev, const int32_t* m_dev, const kpb_match_params* params,
              int32_t* out_pairs_dev, double* out_dist_dev, int32_t* out_k_dev);

/* Match counts of the last kpb_match as host integers: waits for the context's stream, then hands over what match_finalize left in
 * pinned host memory -- the `K` of skimage's `matches` array [K, 2] (utils/matcher.py:227-230) without a device read-back of its own.
 * out_k_host [batch] is host memory; batch must be the batch of that call. */
int kpb_match_counts(kpb_ctx* ctx, int32_t* out_k_host, int batch);

/* ---- M3: utils/matcher.py:231-233 row gather ---------------------------------------------------
 * out[b][i][:] = src[b][idx[b][i*idx_stride + idx_col]][:] for i < k[b].  cols floats per row. */
int kpb_gather_rows(kpb_ctx* ctx, const float* src_dev, int batch, int src_rows, int cols,
                    const int32_t* idx_dev, int idx_rows, int idx_stride, int idx_col,
                    const int32_t* k_dev, float* out_dev);

/* ---- 8(f)1: covisibility warp, utils/projection.py:137-167 warp_homography -----------------------
 * kps_dev [batch][max_n][stride] fp32 rows (x, y, ...) normalised; n_dev [batch] or NULL (= max_n);
 * hmat_dev [batch][9] row-major fp32 homography; wh_dev [batch][2] int32 (width, height) of the target
 * image.  Keeps the points whose warp lands in [0, w-1] x [0, h-1]:
 * out_kps0_dev / out_kps01_dev [batch][max_n][2] = the kept points and their warps (both normalised),
 * out_ids_dev [batch][max_n] = kept row numbers ascending, then the rejected ones ascending,
 * out_n_dev [batch] = number kept. */
int kpb_warp_homography(kpb_ctx* ctx, const float* kps_dev, int batch, int max_n, int stride,
                        const int32_t* n_dev, const float* hmat_dev, const int32_t* wh_dev,
                        float* out_kps0_dev, float* out_kps01_dev, int32_t* out_ids_dev,
                        int32_t* out_n_dev);

/* ---- depth-based covisibility warp, utils/projection.py:195-268 warp_se3 (with interpolate_depth 271-373) ------
 * One image pair per call.  kps_dev [n][stride] normalised; depth0_dev [H0][W0], depth1_dev [H1][W1] fp32 (0 = no
 * depth); cam_dev [38] = inverse(intrinsics0)[9], intrinsics1[9], pose01[16], bbox0 (row, col), bbox1 (row, col).
 * out_kps0_dev / out_kps01_dev [n][2]: covisible points and their warps (normalised by the depth-map sizes);
 * out_ids_dev [n]: their row numbers; out_ids_out_dev [n]: points that leave image 1, then occluded points (each
 * ascending); out_counts_dev [2] = (covisible, left-or-occluded). */
int kpb_warp_se3(kpb_ctx* ctx, const float* kps_dev, int n, int stride, const float* depth0_dev, int H0, int W0,
                 const float* depth1_dev, int H1, int W1, const float* cam_dev, float* out_kps0_dev,
                 float* out_kps01_dev, int32_t* out_ids_dev, int32_t* out_ids_out_dev, int32_t* out_counts_dev);

/* ---- 8(f)1: ground-truth mutual nearest neighbours, tasks/repeatability.py:69-85 (val_key_points;
 * mutual_argmax 9-32, compute_keypoints_distance 39-51) -------------------------------------------
 * k0_dev, k01_dev [batch][max_m][2]: covisible keypoints of image 0 and their warps; k1_dev, k10_dev
 * [batch][max_n][2] likewise for image 1; m_dev / n_dev [batch] or NULL.  scale_dev [batch][2] =
 * (warp01 resize-or-width, warp10 resize-or-width).  dm = (|k0-k10| + |k1-k01|)/2 with the leading
 * diagonal set to 99999; a cell is mutual when `-dm - min(-dm)` equals its row and its column maximum
 * (every tie kept).  out_pairs_dev [batch][cap][2] (i, j) row-major, out_dist_dev [batch][cap] =
 * dm*scale01, out_errors_dev [batch][max_m] = row minimum*scale10, out_counts_dev [batch][2] =
 * (mutual cells -- may exceed cap, in which case only the first cap were written --, cells with
 * dist <= th). */
int kpb_val_keypoints(kpb_ctx* ctx, const float* k0_dev, const float* k01_dev, const float* k1_dev,
                      const float* k10_dev, int batch, int max_m, int max_n, const int32_t* m_dev,
                      const int32_t* n_dev, const float* scale_dev, float th, int32_t* out_pairs_dev,
                      float* out_dist_dev, int cap, float* out_errors_dev, int32_t* out_counts_dev);

/* ---- 8(f)4: the tensor Lucas-Kanade tracker, utils/matcher.py:7-142 OpticalFlow ------------------
 * img1_dev, img2_dev [C][H][W] planar fp32 (the reference's [1,C,H,W]); pts1_dev / pts2_dev [n][pts_stride]
 * normalised (x, y, ...): where the points are in image 1 and where they are expected in image 2;
 * unit_dev [n][2] = (cos, sin) of the random start angles the reference draws (matcher.py:55-56; the
 * caller draws them so that runs are reproducible).  Pyramid of `levels` (avg-pool 2i), `iterations`
 * Gauss-Newton steps per level on a win_size x win_size x C window.  out_pts_dev [n][2] in PIXELS of the
 * full image, as the reference returns them; out_err_dev [n] = min(|out - pts2|, 8). */
typedef struct kpb_lk_params {
    float distance;
    int32_t win_size, levels, iterations;
} kpb_lk_params;
int kpb_lk_track(kpb_ctx* ctx, const float* img1_dev, const float* img2_dev, int C, int H, int W,
                 const float* pts1_dev, const float* pts2_dev, int pts_stride, const float* unit_dev,
                 int n, const kpb_lk_params* params, float* out_pts_dev, float* out_err_dev);

/* ---- configs[3]: epipolar residual of the matches, tasks/FundamentalMatrix.py:137-161 ------------------------------
 * kps0_dev [batch][max_k][cols0] matched rows of image 0, normalised (x, y, ...): scaled to pixels (x (W-1), y (H-1), 1);
 * kps1_dev [batch][max_k][cols1] matched rows of image 1 as the reference's matcher branch leaves them --
 * mode1 0 (brute force, 120-122): used as a 3-vector as they are (normalised x, y, score); 1 (LightGlue, 132-135):
 * scaled to pixels with a 1 appended; 2 (optical flow, 117-119): pixel (x, y) with a 1 appended.
 * k_dev [batch] valid rows or NULL (= max_k); fmat_dev [batch][9] row-major fundamental matrices (batch['fundamental']).
 * out_err_dev [batch][max_k] = |x1^T F x0| / max(|(F x0)_xy|, 1e-6); out_stats_dev [batch][3] = (mean error,
 * share of errors < th, their count). */
int kpb_epipolar_error(kpb_ctx* ctx, const float* kps0_dev, int cols0, const float* kps1_dev, int cols1, int batch,
                       int max_k, const int32_t* k_dev, const float* fmat_dev, int W, int H, int mode1, float th,
                       float* out_err_dev, float* out_stats_dev);

/* ---- 8(f)3: robust homography, cv2.findHomography(pts0, pts1, cv2.RANSAC) as called at tasks/MHA.py:45-47 ----------
 * PARITY UNPINNED: OpenCV is a third-party dependency absent from the reference tree and this image, and its RANSAC
 * draws from its own RNG.  The kernel restates OpenCV's published algorithm with its default parameters (see
 * csrc/geometry.hip, oracle/geometry_ref.py) and is validated against analytic ground truth.
 * m0_dev [batch][max_k][cols0], m1_dev [batch][max_k][cols1]: the matched rows of image 0 / image 1, normalised (x, y, ..);
 * k_dev [batch] valid rows or NULL; scale_dev [batch][4] = (sx0, sy0, sx1, sy1): pixels = normalised * scale, in fp32 as
 * MHA.py:41-42 (both sides (w-1, h-1) of warp01 there).  seed_dev [batch] per-pair sampler seeds, or NULL (= seed).
 * out_h_dev [batch][9] float64 row-major, H[2][2] = 1 (zeros when nothing was found); out_mask_dev [batch][max_k] uint8
 * inliers of the best sample model (as cv2 returns them); out_info_dev [batch][4] = (found 0/1, inliers, hypotheses
 * evaluated, 0).  Fewer than 4 matches: found = 0 (cv2 raises there; tasks/MHA.py has no guard). */
typedef struct kpb_ransac_params {
    double threshold;       /* ransacReprojThreshold, pixels: 3.0 */
    double confidence;      /* 0.995 */
    int32_t max_iters;      /* 2000 */
    int32_t refine;         /* 1: least-squares refit + Levenberg-Marquardt on the inliers, as cv2 does */
} kpb_ransac_params;
int kpb_find_homography(kpb_ctx* ctx, const float* m0_dev, int cols0, const float* m1_dev, int cols1, int batch, int max_k,
                        const int32_t* k_dev, const float* scale_dev, const uint32_t* seed_dev, uint32_t seed,
                        const kpb_ransac_params* params, double* out_h_dev, uint8_t* out_mask_dev, int32_t* out_info_dev);

/* ---- 8(f)3: relative pose, cv2.findEssentialMat(k0, k1, eye(3), threshold, prob, RANSAC) + cv2.recoverPose as called at
 * tasks/AUC.py:50-64 -- PARITY UNPINNED, like kpb_find_homography (OpenCV absent; restated: Nister's five-point solver inside
 * RANSAC with maxIters 1000 and Sampson error, the winning model unrefined; recoverPose = the decomposition with the most
 * masked points in front of both cameras).
 * m0_dev / m1_dev / k_dev / scale_dev / seed_dev / seed: as kpb_find_homography (AUC.py:125-126 scales side 0 with image 0's
 * (w-1, h-1) and side 1 with image 1's).  cam_dev [batch][8] float64 = (cx0, cy0, fx0, fy0, cx1, cy1, fx1, fy1): pixels are
 * normalised as (p - c) / f (AUC.py:47-48) -- in float32 when cam_f32 != 0 (the datasets hand float32 intrinsics over and
 * numpy then stays in float32), in float64 otherwise; thr_dev [batch] float64 = thresh / f_mean (AUC.py:44-45).
 * max_k <= 4096 (the matches stay in LDS: 32 bytes each; config/config_vo.yaml asks top_k 2000).  out_e_dev [batch][9] float64 (Frobenius norm 1; zeros when nothing was found), out_mask_dev [batch][max_k]
 * inliers of that model, out_info_dev [batch][4] = (found, inliers, hypotheses evaluated, 0), out_pts_dev [batch][max_k][4]
 * float64 = the normalised coordinates (u0, v0, u1, v1), input of kpb_recover_pose. */
int kpb_find_essential(kpb_ctx* ctx, const float* m0_dev, int cols0, const float* m1_dev, int cols1, int batch, int max_k,
                       const int32_t* k_dev, const float* scale_dev, const double* cam_dev, int cam_f32, const double* thr_dev,
                       const uint32_t* seed_dev, uint32_t seed, double prob, int max_iters, double* out_e_dev,
                       uint8_t* out_mask_dev, int32_t* out_info_dev, double* out_pts_dev);
/* e_dev / pts_dev / mask_dev / info_dev: outputs of kpb_find_essential; dist: recoverPose's distanceThresh (1e9 at AUC.py:60).
 * out_rt_dev [batch][12] float64 = R row-major then t; out_mask_dev [batch][max_k]: masked points in front of both cameras
 * for the chosen (R, t) (what cv2 leaves in `mask`); out_good_dev [batch]: their number (0: no pose). */
int kpb_recover_pose(kpb_ctx* ctx, const double* e_dev, const double* pts_dev, const uint8_t* mask_dev, int batch, int max_k,
                     const int32_t* k_dev, const int32_t* info_dev, double dist, double* out_rt_dev, uint8_t* out_mask_dev,
                     int32_t* out_good_dev);

/* ---- 8(f)3: fundamental matrix, cv2.findFundamentalMat(pts0, pts1, cv2.FM_RANSAC) as called at utils/mvg.py:16 (the
 * FundamentalMatrixRansac task, tasks/FundamentalMatrix.py:12-86) -- PARITY UNPINNED, like kpb_find_homography (OpenCV absent;
 * restated: 7-point samples, up to three models each, error = the larger squared point-to-epipolar-line distance, the
 * winning minimal model returned unrefined).  Arguments as kpb_find_homography; params: threshold 3.0, confidence 0.99,
 * max_iters 1000 are cv2's defaults, `refine` is ignored.  scale_dev: (w-1, h-1, w-1, h-1) at FundamentalMatrix.py:76-77.
 * out_f_dev [batch][9] float64 row-major with F[2][2] = 1 (zeros when nothing was found), out_mask_dev [batch][max_k] its
 * inliers, out_info_dev [batch][4] = (found, inliers, hypotheses evaluated, 0).  Fewer than 8 matches: found = 0
 * (utils/mvg.py:13-15 does not call cv2 then and keeps every match). */
int kpb_find_fundamental(kpb_ctx* ctx, const float* m0_dev, int cols0, const float* m1_dev, int cols1, int batch, int max_k,
                         const int32_t* k_dev, const float* scale_dev, const uint32_t* seed_dev, uint32_t seed,
                         const kpb_ransac_params* params, double* out_f_dev, uint8_t* out_mask_dev, int32_t* out_info_dev);

/* ---- 8(f)2: per-image input transform after decoding, datasets/hpatches.py:47-69 --------------------
 * src_dev [batch][Hs][Ws][3] uint8 as decoded (BGR from cv2.imread with swap_rb = 1, RGB with 0);
 * out_dev [batch][3][Hd][Wd] fp32 = cv2.resize(src / 255, (Wd, Hd)) (INTER_LINEAR), channels first.
 * Hd == Hs and Wd == Ws is transforms.ToTensor (datasets/megadepth.py:312-313). */
int kpb_preprocess(kpb_ctx* ctx, const uint8_t* src_dev, int batch, int Hs, int Ws, int swap_rb,
                   int Hd, int Wd, float* out_dev);

/* ---- N1..: extractor networks (models/ALike.py:136-164 ALNet.forward, ...) ---------------------
 * arch: KPB_ARCH_*.  blob: a .kpbw container (keypoint_bench_amd/weights.py) holding the folded
 * tensors; copied, the caller may free it. */
#define KPB_ARCH_ALIKE 1        /* models/ALike.py      ALNet (ALIKE-t channel plan), BN folded */
#define KPB_ARCH_SUPERPOINT 2   /* models/SuperPoint.py SuperPointNet, tensors named as its state_dict */
#define KPB_ARCH_XFEAT 3        /* models/XFeat.py      XFeatModel, BN folded */
#define KPB_ARCH_DISK 4         /* models/disk.py       DISK (thin U-Net, 5x5), H, W multiples of 16 */
int kpb_net_create(kpb_ctx* ctx, int arch, const void* blob, size_t len, kpb_net** out);
void kpb_net_destroy(kpb_net* net);
int kpb_net_desc_dim(const kpb_net* net);   /* descriptor channels C */
int kpb_net_desc_div(const kpb_net* net);   /* descriptor map is (H/div) x (W/div): 1 for ALIKE, 8 for SuperPoint/XFeat */
/* img_dev [batch][3][H][W] fp32 RGB in [0,1], H and W multiples of 32 (model_interface.py:192-204).
 * score_out_dev [batch][H][W]; desc_out_dev [batch][H/div][W/div][C] (channels-last storage of the
 * reference's [B,C,H/div,W/div] tensor).  ALIKE only: NULL skips the dense descriptor map (the features
 * needed by kpb_net_desc_at stay resident in the net until the next forward).
 * SuperPoint: H, W multiples of 8; RGB is summed to one channel (SuperPoint.py:42); descriptors L2-normalised. */
int kpb_net_forward(kpb_net* net, const float* img_dev, int batch, int H, int W,
                    float* score_out_dev, float* desc_out_dev);
/* Descriptors of the last forward at keypoints, equal to sampling the dense map with kpb_sample
 * (the head and bilinear sampling are both linear): pts/n/out as in kpb_sample. */
int kpb_net_desc_at(kpb_net* net, const float* pts_dev, int pts_cols, int max_n,
                    const int32_t* n_dev, float* out_dev);

/* ---- N5: models/lightglue.py LightGlue.match (447-477) / _forward (506-652) ----------------------------
 * blob: .kpbw container (arch KPB_ARCH_LIGHTGLUE) with the tensors of the reference state_dict under their
 * own names (transformers.{i}.self_attn.Wqkv.weight ... log_assignment.{i}.final_proj.weight,
 * token_confidence.{i}.token.0.weight, posenc.Wr.weight, optional input_proj.*).
 * desc_scale: the reference's self.desc_scale (8 for SuperPoint maps, 1 for DISK). */
#define KPB_ARCH_LIGHTGLUE 5
typedef struct kpb_lg kpb_lg;
typedef struct kpb_lg_params {      /* LightGlue.default_conf, lightglue.py:335-348 */
    float depth_confidence;         /* 0.95; <= 0 disables early stopping */
    float width_confidence;         /* 0.99; <= 0 disables point pruning */
    float filter_threshold;         /* 0.1 */
    int32_t prune_min_kpts;         /* pruning_keypoint_thresholds: -1 on the reference's CPU path, 1024 / 1536 on CUDA */
} kpb_lg_params;
int kpb_lg_create(kpb_ctx* ctx, const void* blob, size_t len, float desc_scale, kpb_lg** out);
void kpb_lg_destroy(kpb_lg* lg);
int kpb_lg_input_dim(const kpb_lg* lg);
/* Attention arithmetic.  0 (default): every product as split-f16 MFMA triples with fp32 accumulation = the fp32 result of the
 * reference's CPU branch (models/lightglue.py:135-137), which the parity fixtures pin.  1: what the reference runs on a GPU
 * (lightglue.py:129-134: q.half(), k.half(), v.half() through scaled_dot_product_attention, the half result cast back) -- Q, K, V and
 * the probabilities as plain f16, one MFMA per product, fp32 softmax, output rounded to f16.  Opt-in, reported separately by bench.py. */
int kpb_lg_set_attention(kpb_lg* lg, int mode);
/* pts*_dev [batch][max_k][3] normalised (x, y, score) as detection returns them; n*_dev [batch] or NULL;
 * desc*_dev: descriptor maps of the two sides, [batch] x (C, Hd, Wd) with element strides (sb, sc, sh, sw);
 * img_w / img_h: params['w'], params['h'] of the reference call.
 * out_pairs_dev [batch][max_k][2] (index into pts0, index into pts1) ascending in the first index,
 * out_scores_dev [batch][max_k], out_k_dev [batch], out_stop_dev [batch] (layers run; may be NULL).  Asynchronous. */
int kpb_lg_match(kpb_lg* lg, const float* pts0_dev, const float* pts1_dev, const int32_t* n0_dev, const int32_t* n1_dev,
                 int batch, int max_k, const float* desc0_dev, const float* desc1_dev, int C, int Hd, int Wd,
                 int64_t sb, int64_t sc, int64_t sh, int64_t sw, int img_w, int img_h, const kpb_lg_params* params,
                 int32_t* out_pairs_dev, float* out_scores_dev, int32_t* out_k_dev, int32_t* out_stop_dev);

#ifdef __cplusplus
}
#endif
#endif /* KPB_H */
