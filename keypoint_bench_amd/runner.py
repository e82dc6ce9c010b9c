"""The harness side of the hot path, mirroring models/model_interface.py (MInterface) for pair datasets:
crop to multiples of 32 (192-204), two forwards (205-212), one task call per pair, per-task result
lists (103-117) and the end-of-run reductions of on_test_end (119-144).

What the reference does not have and this adds: pairs are sharded one process per
GPU (pair index i goes to rank i mod W); every rank emits one fixed-width fp32 row per pair; ONE
all-gather (RCCL on GPUs, gloo in the CPU tests) moves the rows; rank 0 reduces them exactly as
on_test_end would.  There is no other communication: weights are replicated and pairs are independent.

Tasks (`task_type`): ``repeatability``, ``MHA``, ``AUC``, ``FundamentalMatrix``, ``FundamentalMatrixRansac`` and
``visual_odometer`` run on the device end to end, the robust-geometry stages included (csrc/geometry.hip restates the
OpenCV calls of the reference's tasks; parity unpinned, see utils/mvg.py); ``match_stats`` is a dependency-free task used
by bench.py and the tests: it returns [n_kps0, n_kps1, n_matches].  Any other metric function can be passed in through
``task_fn`` (it then runs pair by pair), and ``install()`` swaps this package's kernels under an importable reference
checkout so that main.py runs unchanged.  Without a ``task_fn`` pairs are batched through PairPipeline / SequencePipeline
and the task is evaluated for a whole batch on the device; host-resident dataset items travel through a pinned staging
ring (HostStager) so that their PCIe copy runs under the previous batch's kernels.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

ROW_WIDTH = 16  # float64 values per pair row: [valid, v0 .. v14] (the widest row: visual_odometer's R, t, step length = 13)


# ------------------------------------------------------------------------------------------ config
def load_config(path):
    """Reads a reference-style YAML (config/config_MHA.yaml): returns the ``test.model.params`` dict plus
    ``test.data.params`` under key 'data_params'."""
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    test = cfg["test"] if "test" in cfg else cfg
    params = dict(test["model"]["params"]) if "params" in test["model"] else dict(test["model"])
    params["data_params"] = (test.get("data") or {}).get("params", {})
    return params


def build_model(params, dense_descriptors=True):
    """model_interface.py:43-86 for the model types this package carries kernels for."""
    mt = params["model_type"]
    if mt == "Alike":
        from .models.ALike import ALNet
        net = ALNet(params["Alike_params"], dense_descriptors=dense_descriptors)
        w = params["Alike_params"].get("weight")
        if w and os.path.exists(w):
            net.load_state_dict(torch.load(w, map_location="cpu"))
        else:   # the checkpoint shipped with the package (folded from the reference's weights/alike-t.pth)
            here = os.path.dirname(os.path.abspath(__file__))
            with open(os.path.join(here, "weights", "alike-t.kpbw"), "rb") as f:
                net.load_packed(f.read())
        return net.eval()
    builders = {"SuperPoint": ("SuperPoint", "SuperPointNet", "SuperPoint_params", None),
                "XFeat": ("XFeat", "XFeatModel", "XFeat_params", None),
                "DISK": ("disk", "DISK", "DISK_params", "extractor")}
    if mt in builders:   # model_interface.py:59-63, 67-69, 76-81
        import importlib
        mod, cls, pkey, sub = builders[mt]
        net = getattr(importlib.import_module("keypoint_bench_amd.models." + mod), cls)()
        w = (params.get(pkey) or {}).get("weight")
        if not (w and os.path.exists(w)):
            raise FileNotFoundError("%s checkpoint %r not found (it is not shipped with the reference tree either)" % (mt, w))
        sd = torch.load(w, map_location="cpu")
        net.load_state_dict(sd[sub] if sub else sd)
        return net.eval()
    raise NotImplementedError("model_type %r: no MI355X kernels in this build (Alike, SuperPoint, XFeat, DISK)" % (mt,))


# ------------------------------------------------------------------------------------------ sharding
def shard_indices(n_items, rank, world):
    """Pair datasets (independent units): rank r takes the pair indices i = r (mod world)."""
    return list(range(rank, n_items, world))


def shard_chunk(n_items, rank, world):
    """Sequence datasets (Exceptions): rank r takes the contiguous frames [r n / W, (r+1) n / W); it also
    READS frame start-1 (one-frame overlap: the `last_batch` of its first step, model_interface.py:217-228) but emits
    no row for it."""
    return list(range(rank * n_items // world, (rank + 1) * n_items // world))


def rows_per_rank(n_items, world):
    return (n_items + world - 1) // world


def pack_rows(values, n_items, rank, world):
    """values: list of per-pair value lists (<= ROW_WIDTH-1 floats each), in shard order.
    Returns a [rows_per_rank, ROW_WIDTH] float64 array, padded rows have valid = 0.  Rows stay float64 end to end: the
    reference keeps its per-pair results (pose errors, the visual-odometry R / t it chains over thousands of frames) in
    float64, and 128 bytes per pair is nothing to the one all-gather."""
    out = np.zeros((rows_per_rank(n_items, world), ROW_WIDTH), np.float64)
    for j, v in enumerate(values):
        v = np.asarray(v, np.float64).ravel()
        assert v.size <= ROW_WIDTH - 1
        out[j, 0] = 1.0
        out[j, 1:1 + v.size] = v
    return out


def gather_rows(local_rows, n_items, device=None, shard=shard_indices):
    """One all-gather of the fixed-width rows; returns [n_items, ROW_WIDTH-1] in item order on every rank.
    `shard` is the rule the ranks used to pick their items (shard_indices or shard_chunk)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    t = torch.from_numpy(np.ascontiguousarray(local_rows, np.float64))
    if world == 1:
        allr = t[None]
    else:
        if device is not None:
            t = t.to(device)
        buf = torch.empty((world * t.shape[0], t.shape[1]), dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(buf, t)     # concatenation along dim 0: rank r owns rows [r*R, (r+1)*R)
        allr = buf.cpu().reshape(world, t.shape[0], t.shape[1])
    allr = allr.numpy()
    out = np.zeros((n_items, ROW_WIDTH - 1), np.float64)
    for r in range(world):
        idx = shard(n_items, r, world)
        rows = allr[r][: len(idx)]
        assert (rows[:, 0] == 1.0).all(), "missing rows from rank %d" % r
        out[idx] = rows[:, 1:]
    return out


# ------------------------------------------------------------------------------------------ reductions
def pose_auc(errors, thresholds):
    """tasks/AUC.py:86-98."""
    errors = np.asarray(errors, np.float64)
    sort_idx = np.argsort(errors)
    errors = errors[sort_idx]
    recall = (np.arange(len(errors)) + 1) / len(errors)
    errors = np.r_[0.0, errors]
    recall = np.r_[0.0, recall]
    aucs = []
    for t in thresholds:
        last = np.searchsorted(errors, t)
        r = np.r_[recall[:last], recall[last - 1]]
        e = np.r_[errors[:last], t]
        trap = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
        aucs.append(trap(r, x=e) / t)
    return aucs


def aggregate(task_type, rows, params=None):
    """on_test_end (model_interface.py:119-164) on the gathered rows."""
    rows = np.asarray(rows, np.float64)
    if task_type == "repeatability":      # rows: [num_feat, repeatability, mean_error]
        err = rows[:, 2]
        return {"num_feat": float(rows[:, 0].mean()), "repeatability": float(rows[:, 1].mean()),
                "rep_mean_err": float(err[~np.isnan(err)].mean()) if (~np.isnan(err)).any() else float("nan")}
    if task_type == "MHA":                # rows: one hit flag per threshold
        n = len(params["MHA_params"]["th"]) if params else 3
        return {"MHA": [float(rows[:, i].mean()) for i in range(n)]}
    if task_type == "AUC":                # rows: [max(err_t, err_R), inliers]
        th = params["AUC_params"]["th"] if params else [5, 10, 20]
        return {"AUC": [float(a) for a in pose_auc(rows[:, 0], th)], "inliers": float(rows[:, 1].mean())}
    if task_type in ("FundamentalMatrix", "FundamentalMatrixRansac"):  # rows: [error, ratio, num]
        return {"fundamental_error": float(rows[:, 0].mean()), "fundamental_radio": float(rows[:, 1].mean()),
                "fundamental_num": float(rows[:, 2].mean())}
    if task_type == "visual_odometer":    # rows: [R (9), t (3), ground-truth step length] per frame; the chain is composed here
        from .tasks.visual_odometer import compose
        r_est, t_est = compose(rows[:, :13])
        return {"r_est": r_est, "t_est": t_est}
    if task_type == "match_stats":        # rows: [n0, n1, matches]
        return {"mean_kps": float(rows[:, :2].mean()), "mean_matches": float(rows[:, 2].mean())}
    raise NotImplementedError(task_type)


# ------------------------------------------------------------------------------------------ per-pair step
SEQUENCE_DATASETS = ("Kitti", "Euroc", "TartanAir")     # model_interface.py:217


def crop32(img):
    """model_interface.py:192-204 (the reference names H 'w' and W 'h'; the effect is a crop of both to x32)."""
    H, W = img.shape[-2:]
    return img[..., : H - H % 32, : W - W % 32]


def is_decoded_u8(v):
    """A decoded image as cv2.imread / PIL hand it over: uint8 [H, W, 3] (the dataset has not run its own transform)."""
    return getattr(v, "dtype", None) in (np.uint8, torch.uint8) and getattr(v, "ndim", 0) == 3 and v.shape[-1] == 3


def as_image(v, device):
    """Dataset items hold numpy arrays [C,H,W] (datasets/hpatches.py:74-83: no DataLoader collation here) or tensors;
    a uint8 [H,W,3] array is a decoded image and gets the datasets' transform (RGB order kept, / 255, HWC -> CHW:
    megadepth.py:312-313 ToTensor) on the device.  A datasets.RawImage (a located, unread .ppm) is read here."""
    from .datasets import RawImage
    if isinstance(v, RawImage):
        v = np.asarray(v)
    if is_decoded_u8(v):
        from .utils.preprocess import to_tensor_resized
        return to_tensor_resized(v, device=device)[0]
    t = torch.as_tensor(v)
    if t.dtype != torch.float32:
        t = t.float()
    return t.to(device, non_blocking=True)


def _host_array(v):
    """numpy view of a host-resident image (numpy array or CPU tensor; a datasets.RawImage stands for the uint8 array it will be read
    as), None for anything already on a device."""
    from .datasets import RawImage
    if isinstance(v, (np.ndarray, RawImage)):
        return v
    if torch.is_tensor(v) and not v.is_cuda:
        return v.detach().numpy()
    return None


def crop32_host(a):
    """crop32 for host arrays: float [.., H, W] or decoded uint8 [H, W, 3]."""
    if is_decoded_u8(a):
        H, W = a.shape[:2]
        if hasattr(a, "read_into"):         # datasets.RawImage: the crop is remembered and applied by the read
            return a.crop(H - H % 32, W - W % 32)
        return a[: H - H % 32, : W - W % 32]
    return crop32(a)


class HostStager:
    """SURVEY 8(f)2, host half: a ring of pinned staging buffers filled by a pool of copy threads.  While the device works
    on batch k, batch k + 1 is gathered from the dataset items into the other pinned buffer (numpy copies release the GIL);
    each batch then crosses PCIe as ONE asynchronous copy.  Decoded uint8 images travel as they are (a quarter of the
    bytes) and become fp32 CHW on the device (csrc/preprocess.hip)."""

    def __init__(self, slots=3, workers=None):
        import concurrent.futures
        import threading
        self.slots = slots
        self.workers = workers or max(2, min(16, len(os.sched_getaffinity(0))))
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.workers)
        import queue
        self.free = queue.Queue()
        for i in range(slots):
            self.free.put(i)
        self.order = queue.Queue()          # slots in the order they were handed out = the order they come back
        self.bufs = {}

    def acquire(self, shape, dtype):
        """Blocks until a slot is free; returns (slot id, pinned tensor of `shape`).  Pinning is slow (about a second per
        GB), so a slot keeps its buffer and serves smaller requests of the same dtype from it."""
        slot = self.free.get()
        self.order.put(slot)
        n = int(np.prod(shape))
        buf = self.bufs.get((slot, dtype))
        if buf is None or buf.numel() < n:
            buf = torch.empty((n,), dtype=dtype).pin_memory()
            self.bufs[(slot, dtype)] = buf
        return slot, buf[:n].view(tuple(shape))

    def release(self):
        """Gives back the oldest slot still out (slots are consumed in the order they were filled).  Never blocks: a release
        with no slot out is a caller's accounting error -- it is reported (a warning, and `unmatched_releases` counts it for the
        tests) rather than turned into a hang or silently dropped."""
        import queue
        try:
            self.free.put(self.order.get_nowait())
        except queue.Empty:
            import warnings
            self.unmatched_releases = getattr(self, "unmatched_releases", 0) + 1
            warnings.warn("HostStager.release() with no slot out: slot accounting error in the caller", RuntimeWarning, stacklevel=2)

    def fill(self, pinned, arrays):
        """pinned[j] <- arrays[j] for every j, in parallel.  A datasets.RawImage (HPatches' .ppm, located but unread) is READ into its
        row: the file's bytes land in pinned memory with no array in between."""
        dst = pinned.numpy()
        n = len(arrays)

        def chunk(k):       # one task per worker, a strided share of the rows each: 512 executor round trips per batch cost more than the copies (r06)
            for j in range(k, n, self.workers):
                a = arrays[j]
                if hasattr(a, "read_into"):
                    a.read_into(dst[j])
                else:
                    np.copyto(dst[j], a.reshape(dst.shape[1:]))
        list(self.pool.map(chunk, range(min(self.workers, n))))

    def close(self):
        self.pool.shutdown(wait=True)


def is_sequence_item(item):
    ds = item.get("dataset")
    if isinstance(ds, (list, tuple)):
        ds = ds[0]
    return ds in SEQUENCE_DATASETS or "image1" not in item


def repeatability_row(idx, img0, score0, desc0, img1, score1, desc1, warp01, warp10, params):
    """The 'repeatability' task of model_interface.py:242-248 on the device: one row [num_feat, repeatability, mean_error]
    per pair (tasks/repeatability.py:87-92), the shape aggregate('repeatability', ...) reduces."""
    from .tasks.repeatability import repeatability
    res = repeatability(idx, img0, score0, img1, score1, warp01, warp10, params)
    return [float(res["num_feat"]), float(res["repeatability"]), float(res["mean_error"])]


def mha_row(idx, img0, score0, desc0, img1, score1, desc1, warp01, warp10, params):
    """model_interface.py:249-253: one hit flag per threshold (tasks/MHA.py:68-72)."""
    from .tasks.MHA import mha
    return [float(v) for v in mha(idx, img0, score0, desc0, img1, score1, desc1, warp01, warp10, params)]


def auc_row(idx, img0, score0, desc0, img1, score1, desc1, warp01, warp10, params):
    """model_interface.py:254-259: [max(err_t, err_R), inliers] (tasks/AUC.py:151-154)."""
    from .tasks.AUC import auc
    r = auc(idx, img0, score0, desc0, img1, score1, desc1, warp01, warp10, params)
    return [float(r["AUC"]), float(r["inliers"])]


def match_stats(idx, img0, score0, desc0, img1, score1, desc1, warp01, warp10, params):
    from .utils.extracter import detection
    from .utils.matcher import brute_force_matcher
    k0 = detection(score0, params["extractor_params"])
    k1 = detection(score1, params["extractor_params"])
    m0, _ = brute_force_matcher(k0, k1, desc0, desc1, params["matcher_params"]["brute_force_params"])
    return [k0.shape[0], k1.shape[0], m0.shape[0]]


def fund_ransac_row(idx, img0, score0, desc0, img1, score1, desc1, warp01, warp10, params, matcher=None):
    """model_interface.py:277-282: [0, kept / detected keypoints, kept keypoints] (tasks/FundamentalMatrix.py:86)."""
    from .tasks.FundamentalMatrix import fundamental_matrix_ransac
    r = fundamental_matrix_ransac(idx, img0, img1, score0, score1, desc0, desc1, matcher, params)
    return [float(r["fundamental_error"]), float(r["fundamental_radio"]), float(r["fundamental_num"])]


TASKS = {"repeatability": repeatability_row, "MHA": mha_row, "AUC": auc_row, "match_stats": match_stats,
         "FundamentalMatrixRansac": fund_ransac_row}


# ---- the same tasks over a whole PairPipeline batch (rows equal to the single-pair functions above)
def _batched_match_stats(pipe, items, params, indices=None):
    B = pipe.B
    n, k = pipe.n.tolist(), pipe.k.tolist()
    return [[n[b], n[B + b], k[b]] for b in range(len(items))]


def _batched_repeatability(pipe, items, params, indices=None):
    from .tasks.repeatability import repeatability_batch
    B, f = pipe.B, len(items)
    pad = lambda ws: ws + [ws[-1]] * (B - f)
    rows = repeatability_batch(pipe.kps, pipe.n, pad([it["warp01_params"] for it in items]), pad([it["warp10_params"] for it in items]),
                               params["repeatability_params"]["th"])
    return rows[:f]


def _batched_mha(pipe, items, params, indices=None):
    from .tasks.MHA import mha_batch
    return mha_batch(pipe, items, params, indices)


def _batched_auc(pipe, items, params, indices=None):
    from .tasks.AUC import auc_batch
    return auc_batch(pipe, items, params, indices)


def _batched_fund_ransac(pipe, items, params, indices=None):
    from .tasks.FundamentalMatrix import fundamental_ransac_batch
    return fundamental_ransac_batch(pipe, items, params, indices)


def _covis_tables(items, B, dev):
    """(hmat [2B,9], wh [2B,2]) of the MHA flow: rows 0..B-1 warp01 of each pair, B..2B-1 warp10 (padded by repetition)."""
    from .tasks.repeatability import homography_tables
    pad = lambda ws: ws + [ws[-1]] * (B - len(ws))
    hm01, wh01, _ = homography_tables(pad([it["warp01_params"] for it in items]), dev)
    hm10, wh10, _ = homography_tables(pad([it["warp10_params"] for it in items]), dev)
    return torch.cat([hm01, hm10]).contiguous(), torch.cat([wh01, wh10]).contiguous()


# task -> (rows function, pipeline needs the match stage, keypoints are covisibility-filtered before matching)
BATCHED_TASKS = {"match_stats": (_batched_match_stats, True, False), "repeatability": (_batched_repeatability, False, False),
                 "MHA": (_batched_mha, True, True), "AUC": (_batched_auc, True, False),
                 "FundamentalMatrixRansac": (_batched_fund_ransac, True, False)}


def _homo_only(item):
    return all(item.get(k, {}).get("mode", "homo") == "homo" for k in ("warp01_params", "warp10_params"))


class PairRunner:
    """Per-rank evaluation loop (MInterface.test_step / on_test_end, model_interface.py:119-299) over

    * a pair dataset: any indexable of dicts with 'image0', 'image1' and optionally 'warp01_params' / 'warp10_params', as
      datasets/hpatches.py:74-83 returns them (numpy [C,H,W] images; tensors work too).  Pairs of equal cropped shape are
      collected `batch` at a time and pushed through PairPipeline (net -> detection -> sampling -> match in one launch
      wave), the task is evaluated on the whole batch on the device; ragged shapes, se3 warps and user-supplied
      `task_fn`s take the single-pair path (`test_step`), whose rows the batched path reproduces bit for bit;
    * a sequence dataset ('dataset' in Kitti / Euroc / TartanAir, items with 'image0' and 'fundamental'): the
      `last_batch` flow of model_interface.py:217-228 with the FundamentalMatrix task (261-276), frames sharded in
      contiguous chunks with a one-frame overlap."""

    def __init__(self, params, task_fn=None, model=None, device="cuda:0", batch=64, matcher=None, dense_descriptors=False):
        self.params = params
        self.device = torch.device(device)
        self.model = model if model is not None else build_model(params, dense_descriptors=dense_descriptors)
        self.matcher = matcher
        self.user_task = task_fn is not None
        self.task_fn = task_fn if task_fn is not None else TASKS.get(params.get("task_type"), match_stats)
        self.batch = int(batch)
        self.results = []
        self.last_batch = None
        self._pipes = {}
        self.batched_pairs = 0
        self.staged_batches = 0
        self._stager = None
        self.decode_workers = None          # threads that run dataset[i] ahead of the staging thread: None = the dataset decides (`thread_safe = True`: one per host core, at most 16; otherwise ONE, in order); 0 = inline; n > 1 = the caller's promise

    # ---- single pair (model_interface.py:189-212 + the task call)
    def test_step(self, batch, idx):
        raw0, raw1 = as_image(batch["image0"], self.device), as_image(batch["image1"], self.device)
        if raw0.dim() == 3:
            raw0, raw1 = raw0[None], raw1[None]
        img0, img1 = crop32(raw0), crop32(raw1)
        with torch.no_grad():
            s0, d0 = self.model(img0)     # model_interface.py:205-207
            s1, d1 = self.model(img1)
        kw = {"matcher": self.matcher} if self.task_fn is fund_ransac_row else {}
        # the tasks get the UNCROPPED images, as model_interface.py:242-259 hands batch['image0'] / batch['image1'] over
        # (MHA.py:59-60 takes its resize factors from their shape)
        r = self.task_fn(idx, raw0, s0, d0, raw1, s1, d1, batch.get("warp01_params", {}), batch.get("warp10_params", {}),
                         self.params, **kw)
        self.results.append(r)
        return r

    # ---- one frame of a sequence (model_interface.py:217-228, 261-276)
    def sequence_step(self, batch, idx, task_type="FundamentalMatrix"):
        from .tasks.FundamentalMatrix import fundamental_matrix
        cur = dict(batch)
        img = as_image(batch["image0"], self.device)
        cur["image0"] = img[None] if img.dim() == 3 else img
        if "fundamental" in batch:
            f = torch.as_tensor(batch["fundamental"], dtype=torch.float32).to(self.device)
            cur["fundamental"] = f[None] if f.dim() == 2 else f                       # what DataLoader collation adds
        if self.last_batch is None:
            self.last_batch = cur
        with torch.no_grad():
            s0, d0 = self.model(self.last_batch["image0"])
            s1, d1 = self.model(cur["image0"])
        last_img = self.last_batch["image0"]
        self.last_batch = cur
        mp = self.params["matcher_params"]
        if task_type == "visual_odometer":    # 283-298: the per-frame motion; the chain is composed from the gathered rows
            from .tasks.visual_odometer import relative_motion, step_length
            R, t = relative_motion(idx, last_img, cur, s0, s1, d0, d1, self.matcher, self.params)
            r = list(R.reshape(9)) + list(t.reshape(3)) + [step_length(cur)]
            self.results.append(r)
            return r
        if mp["type"] == "optical_flow":      # 262-267: the tracker works on the two images
            res = fundamental_matrix(idx, last_img, cur, s0, s1, last_img, cur["image0"], self.matcher, self.params)
        else:
            res = fundamental_matrix(idx, last_img, cur, s0, s1, d0, d1, self.matcher, self.params)
        r = [float(res["fundamental_error"]), float(res["fundamental_radio"]), float(res["fundamental_num"])]
        self.results.append(r)
        return r

    # ---- batched pairs
    def _pipe(self, B, H, W, match, lightglue=None):
        from .pipeline import PairPipeline
        key = (B, H, W, match)
        if key not in self._pipes:
            self._pipes = {k: v for k, v in self._pipes.items() if k[1:3] == (H, W)}      # one resolution resident at a time
            mp = self.params.get("matcher_params", {})
            bf = mp.get("brute_force_params", dict(metric="euclidean", max_distance=float("inf"), cross_check=True))
            self._pipes[key] = (PairPipeline(self.model, self.params["extractor_params"], bf, B, H, W, device=self.device, match=match),
                                torch.empty((2 * B, 3, H, W), dtype=torch.float32, device=self.device))
        return self._pipes[key]

    def _begin(self, group, task_type, staged=None):
        """Enqueues one batch (image gather, net, detection, [covisibility], descriptors, match) and returns without waiting.
        group: list of (index, item, img0, img1) with equal cropped shapes.  staged: device tensor [2f, ...] holding the group's
        images -- views 0 first, then views 1 -- when the HostStager path has brought them over."""
        fn, match, covis = BATCHED_TASKS[task_type]
        f = len(group)
        u8 = is_decoded_u8(group[0][2])
        H, W = group[0][2].shape[:2] if u8 else group[0][2].shape[-2:]
        B = self.batch if f > self.batch // 2 else f        # a short tail gets a pipeline of its own size
        pipe, images = self._pipe(B, H, W, match)
        if staged is not None:      # the group's images, already on the device (copied on the staging stream)
            if u8:                  # decoded bytes: the datasets' transform on the device
                from .utils.preprocess import to_tensor_resized
                staged = to_tensor_resized(staged, device=self.device)
            images[:f].copy_(staged[:f].reshape(f, 3, H, W))
            images[B:B + f].copy_(staged[f:].reshape(f, 3, H, W))
            self.staged_batches += 1
        else:               # device-resident items: ONE gather kernel per view instead of 2 f copy launches
            torch.stack([g[2].reshape(3, H, W) for g in group], out=images[:f])
            torch.stack([g[3].reshape(3, H, W) for g in group], out=images[B:B + f])
        for j in range(f, B):                               # pad with the last pair; its rows are dropped
            images[j].copy_(images[f - 1])
            images[B + j].copy_(images[B + f - 1])
        items = [g[1] for g in group]
        pipe.enqueue(images, _covis_tables(items, B, self.device) if covis else None)
        return fn, pipe, items, [g[0] for g in group]

    def _end(self, token):
        """Waits for the batch `_begin` enqueued (NMS convergence check) and evaluates the task on it.  Returns the rows, or a
        zero-argument callable that computes them from host copies of the device results (tasks with a host half: MHA corner
        errors, AUC pose angles)."""
        fn, pipe, items, idxs = token
        pipe.finish()
        rows = fn(pipe, items, self.params, idxs)
        self.batched_pairs += len(items)
        return rows

    def _flush(self, group, task_type, staged=None, between=None):
        """`_begin`, then `between()` (the previous batch's host-side row arithmetic, under this batch's kernels), then `_end`."""
        token = self._begin(group, task_type, staged)
        if between is not None:
            between()
        return self._end(token)

    def _run_pairs(self, dataset, indices, task_type):
        batched = (not self.user_task) and task_type in BATCHED_TASKS and self.batch > 1 and hasattr(self.model, "_handle")
        if task_type == "FundamentalMatrixRansac":          # the batched rows are the brute-force branch without cv2 drawing
            mp = self.params["matcher_params"]
            batched = batched and mp["type"] == "brute_force" and not mp.get("save_result") and not self.params["extractor_params"].get("save_result")
        out = {}
        host = batched and len(indices) > 0 and _host_array(dataset[indices[0]]["image0"]) is not None
        if host:
            return self._run_pairs_staged(dataset, indices, task_type)
        group, shape = [], None
        late = []               # [(group, callable)]: rows whose host half is still to run (under the next batch's kernels)
        inflight = None         # (group, token): a batch whose kernels are running while the next group is being collected

        def settle():
            while late:
                g, fn_rows = late.pop(0)
                for (i, _, _, _), r in zip(g, fn_rows()):
                    out[i] = r

        def land():
            nonlocal inflight
            if inflight is not None:
                g, token = inflight
                inflight = None
                res = self._end(token)
                if callable(res):
                    late.append((g, res))
                else:
                    for (i, _, _, _), r in zip(g, res):
                        out[i] = r

        def flush():
            nonlocal group, inflight
            if group:
                land()                                              # the pipeline's buffers are free again
                inflight = (group, self._begin(group, task_type))
                settle()                                            # host halves of the batch before, under this batch's kernels
            group = []

        def single(item, i):        # the drop-in path shares the context (one detection in flight at a time): land first
            flush()
            land()
            settle()
            out[i] = self.test_step(item, i)

        for i in indices:
            item = dataset[i]
            if not batched or (task_type not in ("AUC", "FundamentalMatrixRansac") and not _homo_only(item)):
                single(item, i)
                continue
            a, b = crop32(as_image(item["image0"], self.device)), crop32(as_image(item["image1"], self.device))
            if a.shape[-2:] != b.shape[-2:]:                # the two views differ in size: single-pair path
                single(item, i)
                continue
            if shape is not None and a.shape[-2:] != shape:
                flush()
            shape = a.shape[-2:]
            group.append((i, item, a, b))
            if len(group) == self.batch:
                flush()
        flush()
        land()
        settle()
        return [out[i] for i in indices]

    def _run_pairs_staged(self, dataset, indices, task_type):
        """The batched pair path for HOST-resident dataset items (numpy arrays / CPU tensors, fp32 CHW or decoded uint8 HWC):
        a producer thread walks the dataset, groups pairs of equal cropped shape and gathers each group into a pinned
        buffer of the HostStager while the main thread runs the previous group on the device.  Rows are those of the
        device-resident path; pairs the batched path does not take (ragged views, se3 warps for the homography tasks) go
        through `test_step` on the main thread, in order."""
        import queue
        import threading
        if self._stager is None:
            self._stager = HostStager()
        st = self._stager
        q = queue.Queue(maxsize=1)

        def shape_of(a):
            return tuple(a.shape[:2]) if is_decoded_u8(a) else tuple(a.shape[-2:])

        from .datasets import Prefetcher
        fetched = None              # created inside the try below, so that close() runs whatever fails in between
        stop = threading.Event()    # set when the consumer fails: the producer stops at its next item instead of staging the rest

        def produce():
            try:
                group, key = [], None

                def emit():
                    nonlocal group
                    if group:
                        u8 = is_decoded_u8(group[0][2])
                        f = len(group)
                        shp = (2 * f,) + (tuple(group[0][2].shape) if u8 else (3,) + shape_of(group[0][2]))
                        _, pinned = st.acquire(shp, torch.uint8 if u8 else torch.float32)
                        st.fill(pinned, [g[2] for g in group] + [g[3] for g in group])
                        q.put(("batch", group, pinned))
                    group = []

                for i, item in fetched:       # dataset[i], decoded / read a few items ahead on a thread pool (datasets.Prefetcher)
                    if stop.is_set():
                        return
                    a, b = _host_array(item["image0"]), _host_array(item["image1"])
                    single = a is None or b is None or (task_type not in ("AUC", "FundamentalMatrixRansac") and not _homo_only(item))
                    if not single:
                        a, b = crop32_host(a), crop32_host(b)
                        if a.dtype != b.dtype or shape_of(a) != shape_of(b) or (a.dtype != np.uint8 and a.dtype != np.float32):
                            single = True
                    if single:
                        emit()
                        q.put(("single", i, item))
                        continue
                    k = (a.dtype, shape_of(a))
                    if key is not None and k != key:
                        emit()
                    key = k
                    group.append((i, item, a, b))
                    if len(group) == self.batch:
                        emit()
                emit()
                q.put(("end", None, None))
            except BaseException as e:          # surfaces on the main thread
                q.put(("error", e, None))

        out = {}
        copy_stream = None
        pending = None          # (group, device tensor, copy-done event): its PCIe copy runs under the previous group's kernels
        held = []               # copies whose pinned slot has not been given back yet (slot ownership, explicit)

        def start_copy(group, pinned):
            main = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(copy_stream):
                dev = pinned.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            dev.record_stream(main)
            p = (group, dev, ev)
            held.append(p)
            return p

        late = []

        def settle():
            while late:
                g, fn_rows = late.pop(0)
                for (i, _, _, _), r in zip(g, fn_rows()):
                    out[i] = r

        def finish(p):
            group, dev, ev = p
            ev.synchronize()        # the copy has left the pinned slot: the producer may refill it
            st.release()
            held.remove(p)          # from here on a failing task must not give the slot back a second time
            res = self._flush(group, task_type, staged=dev, between=settle)
            if callable(res):
                late.append((group, res))
            else:
                for (i, _, _, _), r in zip(group, res):
                    out[i] = r

        th = None
        try:
            copy_stream = torch.cuda.Stream(self.device)
            fetched = Prefetcher(dataset, indices, workers=self.decode_workers)
            th = threading.Thread(target=produce, daemon=True)
            th.start()
            while True:
                kind, x, y = q.get()
                if kind == "error":
                    raise x
                nxt = start_copy(x, y) if kind == "batch" else None
                if pending is not None:
                    finish(pending)
                pending = nxt
                if kind == "end":
                    break
                if kind == "single":
                    out[x] = self.test_step(y, x)
            settle()
        finally:
            for p in held:                      # only on an error path: slots whose copy was started but never consumed
                try:
                    p[2].synchronize()
                except Exception:
                    pass
                st.release()
            held.clear()
            stop.set()

            def drain():
                try:
                    while True:
                        kind, x, y = q.get_nowait() if th is None or not th.is_alive() else q.get(timeout=0.1)
                        if kind == "batch":     # staged but never copied: its pinned slot goes back to the ring
                            st.release()
                except queue.Empty:
                    pass

            while th is not None and th.is_alive():         # drain so that the producer can finish if the consumer failed
                drain()
            if th is not None:
                th.join()
            drain()                             # the producer's LAST item may still sit in q (maxsize 1) after it has exited
            if fetched is not None:
                fetched.close()
        return [out[i] for i in indices]

    # ---- batched sequence (BASELINE configs[3]: brute-force branch)
    def _run_sequence(self, dataset, indices, task_type="FundamentalMatrix"):
        from .pipeline import SequencePipeline
        from .tasks.FundamentalMatrix import epipolar_error
        mp = self.params["matcher_params"]
        batched = (not self.user_task) and self.batch > 1 and hasattr(self.model, "_handle") and \
            (mp["type"] == "brute_force" or (mp["type"] == "light_glue" and self.matcher is None))
        rows = []
        if not indices:
            return rows
        if not batched:
            self.last_batch = None
            if indices[0] > 0:              # one-frame overlap: the frame before the chunk becomes last_batch, no row
                prev = dict(dataset[indices[0] - 1])
                img = as_image(prev["image0"], self.device)
                prev["image0"] = img[None] if img.dim() == 3 else img
                self.last_batch = prev
            for i in indices:
                rows.append(self.sequence_step(dataset[i], i, task_type))
            return rows
        first = as_image(dataset[indices[0]]["image0"], self.device)
        H, W = first.shape[-2:]
        F = min(self.batch, len(indices))
        pipe = SequencePipeline(self.model, self.params["extractor_params"], mp["brute_force_params"], F, H, W, device=self.device)
        images = torch.empty((F, 3, H, W), dtype=torch.float32, device=self.device)
        if indices[0] > 0:
            pipe.prime(as_image(dataset[indices[0] - 1]["image0"], self.device))
        vo = task_type == "visual_odometer"
        th = None if vo else self.params["FundamentalMatrix_params"]["th"]
        for c0 in range(0, len(indices), F):
            chunk = indices[c0:c0 + F]
            f = len(chunk)
            fm, items = [], []
            for i in chunk:
                item = dataset[i]
                items.append(item)
                if not vo:
                    fm.append(torch.as_tensor(item["fundamental"], dtype=torch.float32).reshape(9))
            frames = [it["image0"] for it in items]
            host = [_host_array(v) for v in frames]
            if all(h is not None and h.dtype == np.float32 for h in host):      # host frames: one pinned gather, one PCIe copy
                if self._stager is None:
                    self._stager = HostStager()
                _, pinned = self._stager.acquire((f, 3, H, W), torch.float32)
                self._stager.fill(pinned, host)
                images[:f].copy_(pinned, non_blocking=True)
                torch.cuda.current_stream(self.device).synchronize()
                self._stager.release()
            elif all(torch.is_tensor(v) and v.is_cuda and v.dtype == torch.float32 for v in frames):    # device frames: one gather kernel
                torch.stack([v.reshape(3, H, W) for v in frames], out=images[:f])
            else:
                for j, v in enumerate(frames):
                    images[j].copy_(as_image(v, self.device).reshape(3, H, W), non_blocking=True)
            pipe.run(images[:f], first=(chunk[0] == 0))
            if vo:
                from .tasks.visual_odometer import relative_motion_batch
                rows.extend(relative_motion_batch(pipe, items, chunk))
                self.batched_pairs += f
                continue
            fmat = torch.stack(fm).to(self.device)
            _, stats = epipolar_error(pipe.m0[:f], pipe.m1[:f], fmat, W, H, 0, th, k_dev=pipe.k)      # FundamentalMatrix.py:120-122: mode 0
            st, kk = stats.cpu().numpy(), pipe.k[:f].tolist()
            for j in range(f):
                if kk[j] == 0:
                    raise ZeroDivisionError("division by zero")        # FundamentalMatrix.py:159 on an empty match set
                rows.append([float(st[j, 0]), float(st[j, 2]) / kk[j], float(st[j, 2])])
            self.batched_pairs += f
        return rows

    def run(self, dataset, task_type=None):
        """Shards the dataset over the ranks, evaluates the local items, gathers the rows (ONE all-gather), reduces on
        every rank like on_test_end.  Returns (aggregate dict, rows [n_items, ROW_WIDTH-1])."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        n = len(dataset)
        task_type = task_type or self.params.get("task_type", "match_stats")
        self.results = []
        sequence = n > 0 and is_sequence_item(dataset[0])
        shard = shard_chunk if sequence else shard_indices
        indices = shard(n, rank, world)
        if sequence:
            self.results = self._run_sequence(dataset, indices, "visual_odometer" if task_type == "visual_odometer" else "FundamentalMatrix")
        else:
            if not self.user_task:
                self.task_fn = TASKS.get(task_type, self.task_fn)
            self.results = self._run_pairs(dataset, indices, task_type)
        rows = gather_rows(pack_rows(self.results, n, rank, world), n, device=self.device if world > 1 else None, shard=shard)
        return aggregate(task_type, rows, self.params), rows


# ------------------------------------------------------------------------------------------ install shim
def install():
    """Swap this package's kernels under an importable reference checkout: see keypoint_bench_amd/shim.py."""
    from . import shim
    return shim.install()
