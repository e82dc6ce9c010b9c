"""The harness side of the hot path, mirroring models/model_interface.py (MInterface) for pair datasets:
crop to multiples of 32 (192-204), two forwards (205-212), one task call per pair, per-task result
lists (103-117) and the end-of-run reductions of on_test_end (119-144).

What the reference does not have and this adds: pairs are sharded one process per
GPU (pair index i goes to rank i mod W); every rank emits one fixed-width fp32 row per pair; ONE
all-gather (RCCL on GPUs, gloo in the CPU tests) moves the rows; rank 0 reduces them exactly as
on_test_end would.  There is no other communication: weights are replicated and pairs are independent.

Tasks: ``repeatability`` runs entirely on the device (detection, covisibility warp, val_key_points).  The other
metric functions of tasks/*.py need cv2 (RANSAC) and stay the reference's own code; pass them in through
``task_fn`` (or let ``install()`` swap this package's kernels under an importable reference checkout so that
main.py runs unchanged).  ``match_stats`` is a dependency-free task used by bench.py and the tests: it returns
[n_kps0, n_kps1, n_matches].
"""
import os

import numpy as np
import torch
import torch.distributed as dist

ROW_WIDTH = 8   # floats per pair row: [valid, v0 .. v6]


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
    """Pair indices of this rank: i = rank (mod world)."""
    return list(range(rank, n_items, world))


def rows_per_rank(n_items, world):
    return (n_items + world - 1) // world


def pack_rows(values, n_items, rank, world):
    """values: list of per-pair value lists (<= ROW_WIDTH-1 floats each), in shard order.
    Returns a [rows_per_rank, ROW_WIDTH] float32 array, padded rows have valid = 0."""
    out = np.zeros((rows_per_rank(n_items, world), ROW_WIDTH), np.float32)
    for j, v in enumerate(values):
        v = np.asarray(v, np.float32).ravel()
        assert v.size <= ROW_WIDTH - 1
        out[j, 0] = 1.0
        out[j, 1:1 + v.size] = v
    return out


def gather_rows(local_rows, n_items, device=None):
    """One all-gather of the fixed-width rows; returns [n_items, ROW_WIDTH-1] in pair-index order on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    t = torch.from_numpy(np.ascontiguousarray(local_rows, np.float32))
    if world == 1:
        allr = t[None]
    else:
        if device is not None:
            t = t.to(device)
        buf = torch.empty((world * t.shape[0], t.shape[1]), dtype=torch.float32, device=t.device)
        dist.all_gather_into_tensor(buf, t)     # concatenation along dim 0: rank r owns rows [r*R, (r+1)*R)
        allr = buf.cpu().reshape(world, t.shape[0], t.shape[1])
    allr = allr.numpy()
    out = np.zeros((n_items, ROW_WIDTH - 1), np.float32)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        rows = allr[r][: len(idx)]
        assert (rows[:, 0] == 1.0).all(), "missing pair rows from rank %d" % r
        out[idx] = rows[:, 1:]
    del rank
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
    if task_type == "FundamentalMatrix":  # rows: [error, ratio, num]
        return {"fundamental_error": float(rows[:, 0].mean()), "fundamental_radio": float(rows[:, 1].mean()),
                "fundamental_num": float(rows[:, 2].mean())}
    if task_type == "match_stats":        # rows: [n0, n1, matches]
        return {"mean_kps": float(rows[:, :2].mean()), "mean_matches": float(rows[:, 2].mean())}
    raise NotImplementedError(task_type)


# ------------------------------------------------------------------------------------------ per-pair step
def crop32(img):
    """model_interface.py:192-204 (the reference names H 'w' and W 'h'; the effect is a crop of both to x32)."""
    H, W = img.shape[-2:]
    return img[..., : H - H % 32, : W - W % 32]


def repeatability_row(idx, img0, score0, desc0, img1, score1, desc1, warp01, warp10, params):
    """The 'repeatability' task of model_interface.py:205-212 on the device: one row [num_feat, repeatability, mean_error]
    per pair (tasks/repeatability.py:87-92), the shape aggregate('repeatability', ...) reduces."""
    from .tasks.repeatability import repeatability
    res = repeatability(idx, img0, score0, img1, score1, warp01, warp10, params)
    return [float(res["num_feat"]), float(res["repeatability"]), float(res["mean_error"])]


TASKS = {"repeatability": repeatability_row}


def match_stats(idx, img0, score0, desc0, img1, score1, desc1, warp01, warp10, params):
    from .utils.extracter import detection
    from .utils.matcher import brute_force_matcher
    k0 = detection(score0, params["extractor_params"])
    k1 = detection(score1, params["extractor_params"])
    m0, _ = brute_force_matcher(k0, k1, desc0, desc1, params["matcher_params"]["brute_force_params"])
    return [k0.shape[0], k1.shape[0], m0.shape[0]]


class PairRunner:
    """Per-rank evaluation loop over a pair dataset (any indexable giving dicts with 'image0', 'image1'
    and optionally 'warp01_params' / 'warp10_params', as datasets/hpatches.py:74-83 does)."""

    def __init__(self, params, task_fn=None, model=None, device="cuda:0"):
        self.params = params
        self.device = torch.device(device)
        self.model = model if model is not None else build_model(params)
        self.task_fn = task_fn if task_fn is not None else TASKS.get(params.get("task_type"), match_stats)
        self.results = []

    def test_step(self, batch, idx):
        img0 = crop32(batch["image0"]).to(self.device)
        img1 = crop32(batch["image1"]).to(self.device)
        if img0.dim() == 3:
            img0, img1 = img0[None], img1[None]
        s0, d0 = self.model(img0)     # model_interface.py:205-207
        s1, d1 = self.model(img1)
        r = self.task_fn(idx, img0, s0, d0, img1, s1, d1, batch.get("warp01_params", {}), batch.get("warp10_params", {}),
                         self.params)
        self.results.append(r)
        return r

    def run(self, dataset, task_type=None):
        """Shards the dataset over the ranks, runs test_step on the local pairs, gathers, reduces on every rank."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        n = len(dataset)
        self.results = []
        for i in shard_indices(n, rank, world):
            self.test_step(dataset[i], i)
        rows = gather_rows(pack_rows(self.results, n, rank, world), n, device=self.device if world > 1 else None)
        return aggregate(task_type or self.params.get("task_type", "match_stats"), rows, self.params), rows


# ------------------------------------------------------------------------------------------ install shim
def install():
    """Swap this package's kernels under an importable reference checkout (its root on sys.path) so that
    ``python3 main.py -c config/config_MHA.yaml test`` and every task run unchanged: replaces
    utils.extracter.{detection,fast_nms}, utils.matcher.brute_force_matcher and models.ALike.ALNet, and
    re-binds the names in task / harness modules that imported them earlier."""
    import importlib
    import sys
    from .models.ALike import ALNet
    from .models.SuperPoint import SuperPointNet
    from .models.XFeat import XFeatModel
    from .models.disk import DISK
    from .utils import extracter as ex, matcher as ma
    swapped = []
    for modname, names in (("utils.extracter", {"detection": ex.detection, "fast_nms": ex.fast_nms}),
                           ("utils.matcher", {"brute_force_matcher": ma.brute_force_matcher, "OpticalFlow": ma.OpticalFlow,
                                              "optical_flow_tensor": ma.optical_flow_tensor}),
                           ("models.ALike", {"ALNet": ALNet}), ("models.SuperPoint", {"SuperPointNet": SuperPointNet}),
                           ("models.XFeat", {"XFeatModel": XFeatModel}), ("models.disk", {"DISK": DISK})):
        try:
            mod = importlib.import_module(modname)
        except Exception:
            continue
        for k, v in names.items():
            setattr(mod, k, v)
            swapped.append(modname + "." + k)
    # the two covisibility warps and the repeatability core (`warp` keeps dispatching on params['mode'])
    from .tasks import repeatability as rp
    from .utils import projection as pj
    try:
        mod = importlib.import_module("utils.projection")
        mod.warp_homography = pj.warp_homography
        mod.warp_se3 = pj.warp_se3
        swapped += ["utils.projection.warp_homography", "utils.projection.warp_se3"]
        mod = importlib.import_module("tasks.repeatability")

        mod.val_key_points = rp.val_key_points
        swapped.append("tasks.repeatability.val_key_points")
    except Exception:
        pass
    for name, mod in list(sys.modules.items()):
        if mod is None or not (name.startswith("tasks.") or name == "models.model_interface"):
            continue
        for k, v in (("detection", ex.detection), ("brute_force_matcher", ma.brute_force_matcher), ("optical_flow_tensor", ma.optical_flow_tensor), ("ALNet", ALNet),
                     ("SuperPointNet", SuperPointNet), ("XFeatModel", XFeatModel), ("DISK", DISK)):
            if hasattr(mod, k):
                setattr(mod, k, v)
                swapped.append(name + "." + k)
    return swapped
